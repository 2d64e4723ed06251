#!/usr/bin/env python
"""bench.py — env steps/sec of the Track2D hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE config 3 — Track2D-BlockPartialPZR-v0 (AD-VAT dueling), 4096 envs per GPU,
tat-maze-lstm tracker + target, train-mode -1, 20-step rollouts. A "step" is ONE batched env step of the full
A3C path: policy forward for both players (PyTorch-ROCm) -> HIP step/observe kernel for all envs (in-launch
auto-reset) -> every 20th step the n-step/GAE loss, backward, ONE all-reduce of the flat gradient bucket (RCCL)
and the SharedAdam update. value = (K x envs over all ranks) / max-over-ranks wall time of the MEDIAN of >= 5
repeats of the K-step region. Two schedules of the same iteration are measured (--schedule both, the default): the
PIPELINED one (main.py's default: the rollout of iteration i + 1 runs on a second HIP stream while the learner, all-reduce
and update of iteration i run, every gradient exactly one update late — the bounded form of the reference's worker
asynchrony; a timed region of K steps = K / 20 phases of that pipeline, each one rollout next to the previous rollout's
update) and the SYNCHRONOUS one (no overlap, no delay). `value` is the faster one's (`schedule` names it), the other's
numbers ride along in the `synchronous` / `pipelined` object. Weak scaling (`value`): 4096 envs per GPU, independent shards keyed by global env id;
the strong form of the same metric (4096 envs GLOBAL, 4096/N per GPU — SURVEY §8d's headline) is measured in the same
run and reported in the `strong` object of the same line.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (re-exec under
torch.distributed.run on 127.0.0.1, the role main.py:102-116 plays in the reference); it refuses to report unless
the process group really has N ranks (checked with an all-reduce of ones and an all-gather of the device ids).

Extra objects on the same JSON line:
  roofline     SURVEY 8(d)'s figure for the env kernel OF THE TIMED REGION: algorithmic bytes per env-step (709 B with byte
               observations, 1723 B with float ones) x envs per launch / that kernel's duration IN SITU. Since round 3 the rollout
               ends every env step inside k_act_step (both LSTM cells + heads + draws + env step + observation), so its duration
               is measured where it runs: a whole 20-step rollout of the player is captured into a hipGraph twice — as it is,
               and with the k_act_step launches left out — and replayed alternately with HIP events on the launch stream; the
               difference / 20 is the kernel's in-situ cost (`avg_launch_us`; rocprofv3's average for the kernel in a replayed
               iteration, profiles/r06_iteration_kernel_stats.txt, rides along as `rocprof_in_iteration_us` and must agree).
               `frac` = that figure / 8 TB/s. The same duration priced with EVERYTHING the fused kernel moves (gate
               pre-activations, cell / hidden state, activated gates parked for the learner) is reported under its own name,
               `policy_state_included`, never as `frac`. `traffic` = HBM bytes per launch from the committed rocprofv3 --pmc
               passes (profiles/r06_pmc_traffic.json). `other_variants`: the stand-alone step kernel (k_step2; what t2d_step /
               t2d_step_u8 launch for callers that bring their own actions), measured alone in a 9-launch graph.
  env_only     the stand-alone step kernel driven with on-device random actions (no policy): launches/s -> env steps/s, per
               launch and in the persistent mode (t2d_rollout_random: up to 10 steps per launch).
  policy_stem  informational f32-MFMA roofline of the conv-stem kernels (the largest single kernels of the iteration).
  cpu_baseline reference-shaped 16-worker (+1 evaluator) CPU A3C on the oracle (oracle/cpu_a3c.py --suite), rank 0, N=1
               only: the headline workload; `cpu_baselines` adds BASELINE config 1 and the env-only 1/16-process rates.
"""
import argparse
import json
import os
import sys
import time

# The pipelined schedule's CU partition (train.PipelinedIteration.tune_streams) gives each chain a CU-masked stream with a
# hardware queue of its own; iterations slow down 2-3x once a process owns more than four hardware queues (measured), so a
# single-GPU run lets the ordinary streams share two (read by the HIP runtime at initialisation: set before torch touches the
# GPU). Multi-rank runs keep the runtime's default — RCCL's streams want queues too, and there is no N > 1 box to measure on.
_SET_HW_QUEUES = int(os.environ.get("WORLD_SIZE", "1")) == 1 and "GPU_MAX_HW_QUEUES" not in os.environ
if _SET_HW_QUEUES:
    os.environ["GPU_MAX_HW_QUEUES"] = "2"

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_STEP = 1723          # algorithmic bytes per env-step
B_STEP_U8 = 709        # ... of the u8-observation variant (obs 338 B instead of 1352 B)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n, argv):
    """Start `n` ranks of this script under torch.distributed.run (one per GPU, rendezvous on 127.0.0.1) and return
    the exit code. Rank 0 prints the JSON line; stdout/stderr are inherited."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    if _SET_HW_QUEUES:                    # (this launcher process set it for a single-GPU run: the ranks keep the runtime's default)
        env.pop("GPU_MAX_HW_QUEUES", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def init_ranks(a):
    """Join the process group (RCCL unless BENCH_DIST_BACKEND says otherwise) and PROVE it has a.gpus ranks.
    Returns (world, rank, local_rank, info) where info goes into the JSON line."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d — refusing to report a %d-GPU number from %d rank(s)"
                         % (a.gpus, world, a.gpus, world))
    # test hooks (not used by the driver): run N ranks on ONE device over gloo to exercise the N>1 code path
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    use_cuda = torch.cuda.is_available()
    info = {"backend": backend if world > 1 else None, "ranks": 1, "devices": [local_rank]}
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if use_cuda:
            torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        dev = torch.device("cuda", local_rank) if use_cuda else torch.device("cpu")
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                      # brings the communicator up; the sum counts its ranks
        ids = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(ids, torch.tensor([local_rank], dtype=torch.int64, device=dev))
        n_comm = int(round(float(ones.item())))
        if dist.get_world_size() != a.gpus or n_comm != a.gpus:
            raise SystemExit("bench.py: communicator has %d ranks (world_size %d), expected %d"
                             % (n_comm, dist.get_world_size(), a.gpus))
        info.update(ranks=n_comm, devices=[int(t.item()) for t in ids])
    return world, rank, local_rank, info


def _lt_status():
    """Where the kernel choices of the direct hipBLASLt calls (csrc/lt_gemm.cpp) come from in THIS process: the validated record
    (lt_tuning_gfx950.json made with the loaded library build) or timing at first use."""
    from active_tracking_rl_amd import fused
    return fused.lt_tuning_status()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--env", default="Track2D-BlockPartialPZR-v0")
    ap.add_argument("--network", default="tat-maze-lstm")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="run the iteration eagerly instead of as hipGraphs")
    ap.add_argument("--per-step-autograd", action="store_true",
                    help="reference-shaped learner (autograd graph built during the rollout) instead of the "
                         "actor/learner split with time-batched re-evaluation")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--actor-step", choices=("mfma", "gemm"), default=None,
                    help="A/B switch of the actor's LSTM step: one MFMA kernel (csrc/actor_step_hip.hip) or library GEMMs + "
                         "fused cell kernel (default: the model's setting)")
    ap.add_argument("--f32-obs", action="store_true", help="float32 observations between env and policy (default: bytes, "
                                                          "decoded in the stem's conv1)")
    ap.add_argument("--repeats", type=int, default=5, help="minimum number of timed repeats of the K-step region (median "
                                                          "reported; more are run until they hold >= 1 s of GPU work)")
    ap.add_argument("--schedule", choices=("both", "pipelined", "synchronous"), default="both",
                    help="iteration schedule behind `value`: pipelined (rollout i+1 under learner i, one update of gradient "
                         "delay; train.PipelinedIteration) or synchronous (train.GraphedIteration); both = value from the "
                         "pipelined one, the synchronous numbers in the `synchronous` object")
    ap.add_argument("--pipelined-timeout", type=float, default=240.0,
                    help="seconds the pipelined measurements may take before the line is printed with the synchronous "
                         "numbers only (a watchdog: an N>1 run must not hang the driver)")
    ap.add_argument("--no-shards", action="store_true", help="skip the per-shard-size sweep of the N=1 line")
    ap.add_argument("--global-envs", type=int, default=4096, help="env count of the strong-scaling form")
    ap.add_argument("--launch-check", action="store_true",
                    help="only start the ranks, verify the communicator and print a stub line (runs without a GPU "
                         "with BENCH_DIST_BACKEND=gloo; covers the launcher in the CPU test-suite)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus, sys.argv[1:]))
    world, rank, local_rank, comm = init_ranks(a)
    if a.launch_check:
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, **comm}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from active_tracking_rl_amd import build
    if rank == 0:
        build.build()
    if world > 1:
        dist.barrier()
    from active_tracking_rl_amd import gemm_tuning
    tuned = gemm_tuning.enable() and os.environ.get("ATR_DISABLE_GEMM_TUNING") != "1"
    from active_tracking_rl_amd.train import (GraphedIteration, PipelinedIteration, capture_allreduce_default, default_args,
                                              make_player, rollout)

    T = 20
    steps = max(T, (a.steps + T - 1) // T * T)          # an A3C iteration is T env steps + one update
    globals_steps = steps
    warm = max(T, (a.warmup + T - 1) // T * T)
    repeats = max(1, a.repeats)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    def measure(envs_per_gpu, min_gpu_seconds=1.0, schedule="synchronous", region_steps=None):
        """Build the player for `envs_per_gpu` envs on this rank, warm up, then time `repeats` repeats of `steps` env
        steps (region_steps when given), each bracketed by barrier + synchronize; per repeat the MAX over ranks; returns the
        median repeat."""
        steps = globals_steps if region_steps is None else int(region_steps)
        args = default_args(env=a.env, network=a.network, num_envs=envs_per_gpu, num_steps=T, gpu_ids=[local_rank],
                            aux="reward" if "tat" in a.network else "none", train_mode=-1, obs_u8=not a.f32_obs)
        player, optimizer = make_player(args, device, rank, world)
        if a.actor_step is not None:       # "mfma": round 2's atr_actor_step path from 3072 rows up; "gemm": never
            player.model.fused_actor_step = a.actor_step == "mfma"
            player.model.mfma_step_min_rows = 3072 if a.actor_step == "mfma" else (1 << 30)

        def eager_iteration():
            rollout(player, T, fast=not a.per_step_autograd)
            player.optimize(None, optimizer, player.model, args.train_mode, device)

        iteration, graphed, drain, trials, flush = eager_iteration, False, (lambda: None), None, (lambda: None)
        if schedule == "pipelined":
            sched = PipelinedIteration(player, optimizer, args)
            trials = [{"ms_per_iteration": round(ms, 4), "chosen": bool(c), "streams": lb} for ms, c, lb in sched.tune_streams()]
            # timed regions end on a PHASE boundary of the pipeline (sched.sync): each holds K env steps of rollouts and K / T
            # learner updates — the updates of the rollouts one call back
            iteration, graphed, drain, flush = sched.run, True, sched.sync, sched.finish
        elif not a.no_graph:
            try:
                iteration = GraphedIteration(player, optimizer, args, fast=not a.per_step_autograd).run
                graphed = True
            except Exception as ex:  # fall back to eager, and say so in the JSON line
                print("hipGraph capture failed, running eagerly: %r" % (ex,), file=sys.stderr)
                torch.cuda.synchronize(device)
        for _ in range(warm // T):
            iteration()
        drain()

        def timed_repeat():
            fence()
            t0 = time.perf_counter()
            for _ in range(steps // T):
                iteration()
            drain()                      # pipelined: the caller's stream joins both streams; then barrier + synchronize
            fence()
            return time.perf_counter() - t0
        # whatever --steps is, the timed repeats together hold >= min_gpu_seconds of GPU work (a 20-step region is 6 ms):
        # a pilot repeat sizes the count, every rank uses the same count (max over ranks of the pilot time)
        pilot = torch.tensor([timed_repeat()], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(pilot, op=dist.ReduceOp.MAX)
        reps = max(repeats, int(min_gpu_seconds / max(float(pilot.item()), 1e-6)) + 1)
        times = [timed_repeat() for _ in range(reps)]
        tt = torch.tensor(times, dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ts = sorted(tt.tolist())
        med = ts[len(ts) // 2]
        flush()
        torch.cuda.synchronize(device)
        ar_us = None
        if world > 1:   # the one collective of the path, timed alone: flat fp32 gradient bucket, eager, back to back
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(5):
                player.allreduce_grads(optimizer)
            fence()
            e0.record()
            for _ in range(50):
                player.allreduce_grads(optimizer)
            e1.record()
            torch.cuda.synchronize(device)
            tu = torch.tensor([e0.elapsed_time(e1) * 1e3 / 50], dtype=torch.float64, device=device)
            dist.all_reduce(tu, op=dist.ReduceOp.MAX)
            ar_us = float(tu.item())
        res = {"value": steps * envs_per_gpu * world / med, "ms_per_step": med * 1e3 / steps,
               "ms_per_iteration": med * 1e3 / (steps // T), "timed_gpu_seconds": float(sum(ts)),
               "envs_per_gpu": envs_per_gpu, "global_envs": envs_per_gpu * world, "repeats": reps, "region_steps": steps,
               "spread": {"min_ms_per_step": ts[0] * 1e3 / steps, "max_ms_per_step": ts[-1] * 1e3 / steps},
               "hipgraph": graphed, "schedule": schedule, "stream_trials_ms_per_iteration": trials, "allreduce_us": ar_us,
               "allreduce_elems": int(optimizer.bucket.grad.numel()) if hasattr(optimizer, "bucket") else None}
        return res, player, optimizer, args

    def measure_set(schedule, keep_player):
        """All the timed regions of one schedule: the strong form (N > 1: 4096 envs GLOBAL, sharded), at N = 1 the per-GPU
        shard sizes of the strong form at every N (4096/8, /4, /2 envs on this one GPU), and the weak form (headline)."""
        strong = None
        if world > 1:
            per = a.global_envs // world
            strong, p_, o_, _ = measure(per, schedule=schedule)
            strong["scaling"] = "strong"
            p_.env.close()
            del p_, o_
            torch.cuda.empty_cache()
        shards = None
        if world == 1 and not a.no_shards:
            shards = {"note": "headline workload at the per-GPU shard sizes of the strong form (4096 envs over 8/4/2/1 "
                              "GPUs), measured on ONE GPU; the N-GPU strong value is bounded by N x shard value, minus the "
                              "gradient all-reduce (allreduce_us of an N>1 line: %d fp32 elements)" % 0, "sizes": {}}
            # (the shard entries are this file's own auxiliary measurement: regions of >= 200 steps whatever --steps is — a
            # fence after every 20-step region costs a 512-env shard ~6 % — each entry says so in `region_steps`)
            for per in (a.global_envs // 8, a.global_envs // 4, a.global_envs // 2):
                r_, p_, o_, _ = measure(per, min_gpu_seconds=0.5, schedule=schedule, region_steps=max(steps, 10 * T))
                shards["sizes"][str(per)] = {k: r_[k] for k in ("value", "ms_per_iteration", "ms_per_step", "repeats",
                                                                "spread", "stream_trials_ms_per_iteration", "region_steps")}
                p_.env.close()
                del p_, o_
                torch.cuda.empty_cache()
        weak, player, optimizer, args = measure(a.envs_per_gpu, schedule=schedule)
        if shards is not None:
            shards["sizes"][str(a.envs_per_gpu)] = {k: weak[k] for k in ("value", "ms_per_iteration", "ms_per_step", "repeats",
                                                                         "spread", "stream_trials_ms_per_iteration",
                                                                         "region_steps")}
            shards["note"] = shards["note"].replace("(allreduce_us of an N>1 line: 0 fp32 elements)",
                                                    "(allreduce_us of an N>1 line: %s fp32 elements)" % weak["allreduce_elems"])
        if strong is None:
            strong = dict(weak, scaling="strong", note="N=1: the strong and weak forms coincide")
        if not keep_player:
            player.env.close()
            player = optimizer = None
            torch.cuda.empty_cache()
        return {"strong": strong, "shards": shards, "weak": weak}, player, optimizer, args

    # the synchronous schedule first (its player also serves the kernel sections below); the pipelined one is measured after
    # the line is assembled, under a watchdog
    first = "pipelined" if (a.schedule == "pipelined" and not a.no_graph) else "synchronous"
    want_pipelined = a.schedule == "both" and not a.no_graph and not a.per_step_autograd
    res0, player, optimizer, args = measure_set(first, keep_player=True)
    strong, shards, weak = res0["strong"], res0["shards"], res0["weak"]
    graphed = weak["hipgraph"]
    value, n_total = weak["value"], a.envs_per_gpu * world

    # ---- roofline of the step/observe kernel: HIP events on the launch stream --------------------------
    # M policy-shaped launches (int64 action tensors, fresh per launch) are captured once into a hipGraph and
    # replayed, so the events bracket back-to-back GPU work rather than the Python/ctypes launch rate. The time
    # includes the generator launch (k_gen) every 10th step — the amortised cost of in-launch auto-reset.
    core = player.env.core
    n = a.envs_per_gpu
    M, REPS = 100, 5
    acts = torch.randint(0, 4, (M, 2, n), device=device)
    out = (torch.empty((n, 2, 13, 13), device=device), torch.empty((n, 2), device=device),
           torch.empty((n,), dtype=torch.uint8, device=device))
    core.flush()
    torch.cuda.synchronize(device)
    side = torch.cuda.Stream(device=device)
    with torch.cuda.stream(side):
        for i in range(20):
            core.step(acts[i, 0], acts[i, 1], out)
        core.flush()
    torch.cuda.synchronize(device)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for i in range(M):
            core.step(acts[i, 0], acts[i, 1], out)
        core.flush()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay()
    torch.cuda.synchronize(device)
    e0.record()
    for _ in range(REPS):
        g.replay()
    e1.record()
    torch.cuda.synchronize(device)
    step_us = e0.elapsed_time(e1) * 1e3 / (M * REPS)          # per step incl. the generator pass every 10th step
    # the step kernel alone: a graph of gen_every - 1 = 9 launches captured right after a generator pass holds no
    # k_gen. (Timing only: replays do not advance the library's host-side window counter, so no pass runs between
    # them and an env that finishes twice re-enters the same pre-generated episode; the env is reset afterwards.)
    core.flush()
    torch.cuda.synchronize(device)
    g9 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g9, capture_error_mode="thread_local"):
        for i in range(9):
            core.step(acts[i, 0], acts[i, 1], out)
    core.flush()
    torch.cuda.synchronize(device)
    tot = 0.0
    for _ in range(40):
        e0.record()
        g9.replay()
        e1.record()
        torch.cuda.synchronize(device)
        tot += e0.elapsed_time(e1)
    k_us = tot * 1e3 / (40 * 9)
    # the u8-observation variant of the same kernel (t2d_step_u8; SURVEY 8(d): B_step = 709 B, reported under its own name)
    k8_us = None
    if getattr(core, "supports_u8", False):
        out8 = (torch.empty((n, 2, 13, 13), dtype=torch.uint8, device=device), out[1], out[2])
        core.flush()
        torch.cuda.synchronize(device)
        g8 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g8, capture_error_mode="thread_local"):
            for i in range(9):
                core.step_u8(acts[i, 0], acts[i, 1], out8)
        core.flush()
        torch.cuda.synchronize(device)
        tot = 0.0
        for _ in range(40):
            e0.record()
            g8.replay()
            e1.record()
            torch.cuda.synchronize(device)
            tot += e0.elapsed_time(e1)
        k8_us = tot * 1e3 / (40 * 9)
    # the fused end-of-step kernel (k_act_step: both players' cells + heads + draws + env step), which the timed region uses at
    # every batch size: (1) alone, a 9-launch graph with gate tensors of the shape the rollout feeds it; (2) IN SITU
    fused_in_region = bool(getattr(player.model, "env_step_fused_seen", False))
    ka_us, ka_bytes, ka_situ_us, situ_detail = None, None, None, None
    one_gate = masked_h = False
    if getattr(core, "supports_u8", False) and n % 2 == 0:
        from active_tracking_rl_amd import fused as fz
        mdl = player.model
        R = args.rnn_out
        one_gate = n <= getattr(mdl, "pair_gemm_max_rows", 0) or (getattr(mdl, "cat_gate_gemm", False)
                                                                  and n >= getattr(mdl, "cat_gemm_min_rows", 1 << 30))
        masked_h = getattr(mdl, "cat_gate_gemm", False) and n >= getattr(mdl, "cat_gemm_min_rows", 1 << 30)
        # (round 5) on that path the rollout keeps the gate GEMM's output for the learner instead of the activated gates
        keep_pre = bool(masked_h and getattr(mdl, "store_preacts", False) and not getattr(mdl, "coop_step", False))
        gts = torch.randn(2, n, 4 * R, device=device)
        hgt = None if one_gate else torch.randn(2, n, 4 * R, device=device)
        cprev, hout, cout = (torch.zeros(2, n, R, device=device) for _ in range(3))
        actst = torch.empty(2, n, 4 * R, device=device)
        actn = torch.empty(2, n, dtype=torch.int64, device=device)
        rows_next = torch.empty(2, n, 256 + R, device=device) if masked_h else None
        smp = fz.ActionSampler(device, seed=5)
        heads = (mdl.player0.actor.actor_linear, mdl.player1.actor.actor_linear)
        bsum = [torch.zeros(4 * R, device=device) for _ in range(2)]
        embt = torch.randn(4, 4 * R, device=device) if getattr(mdl, "tat", False) else None
        out8 = (torch.empty((n, 2, 13, 13), dtype=torch.uint8, device=device), out[1], out[2])

        def act_launch():
            fz.act_env_step(core, [gts[0], gts[1]], [hgt[0], hgt[1]] if hgt is not None else None, bsum,
                            [cprev[0], cprev[1]], out[2], [hout[0], hout[1]], [cout[0], cout[1]],
                            None if keep_pre else [actst[0], actst[1]], smp,
                            heads, actn, emb=embt, env_out=out8,
                            hm_out=[rows_next[0][:, 256:], rows_next[1][:, 256:]] if masked_h else None)
        core.flush()
        smp.begin_block()
        with torch.cuda.stream(side):
            act_launch()
            core.flush()
        torch.cuda.synchronize(device)
        ga = torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga, capture_error_mode="thread_local"):
            for i in range(9):
                act_launch()
        smp.end_block()
        core.flush()
        torch.cuda.synchronize(device)
        tot = 0.0
        for _ in range(40):
            e0.record()
            ga.replay()
            e1.record()
            torch.cuda.synchronize(device)
            tot += e0.elapsed_time(e1)
        ka_us = tot * 1e3 / (40 * 9)
        # everything the fused kernel moves per env-step: the env's 709 B (u8 observations) + per player: gate pre-activations
        # read (4R floats; twice when ig / hg arrive separately), c_prev read, h / c written, activated gates written (the
        # learner's cache; not when the rollout keeps the GEMM's output instead: keep_pre), the masked hidden row for the next step's GEMM (one-GEMM path), + actions 16 B + previous done 1 B
        per_player = 4 * R * 4 * (1 if one_gate else 2) + 3 * R * 4 + (0 if keep_pre else 4 * R * 4) + (R * 4 if masked_h else 0)
        ka_bytes = B_STEP_U8 + 2 * per_player + 17
        # (2) in situ: a whole T-step rollout of the player as a hipGraph, with and without the k_act_step launches; replayed
        # alternately, HIP events on the launch stream around each replay; difference / T = what the kernel costs where it runs
        if fused_in_region:
            def capture_rollout(skip_act):
                real = fz.act_env_step

                def no_act(env_core, ig, hg, biases, c_prev, done, h_out, c_out, acts, sampler, actors, actions_out, **kw):
                    sampler._ordinal += 2          # (the launch is left out; the draw bookkeeping of the block goes on)
                    return actions_out
                if skip_act:
                    fz.act_env_step = no_act
                try:
                    core.flush()
                    torch.cuda.synchronize(device)
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                        rollout(player, T)
                        player.clear_actions()
                        if hasattr(player.model, "cache_dense"):
                            player.model.cache_dense(False)
                finally:
                    fz.act_env_step = real
                return gr
            g_full, g_skip = capture_rollout(False), capture_rollout(True)
            for gg in (g_full, g_skip):
                gg.replay()
            torch.cuda.synchronize(device)
            t_full = t_skip = 0.0
            REP = 30
            for _ in range(REP):
                for which, gg in (("full", g_full), ("skip", g_skip)):
                    e0.record()
                    gg.replay()
                    e1.record()
                    torch.cuda.synchronize(device)
                    if which == "full":
                        t_full += e0.elapsed_time(e1)
                    else:
                        t_skip += e0.elapsed_time(e1)
            ka_situ_us = (t_full - t_skip) * 1e3 / (REP * T)
            situ_detail = {"rollout_graph_us": t_full * 1e3 / REP, "rollout_graph_without_k_act_step_us": t_skip * 1e3 / REP,
                           "launches": T, "replays_each": REP}
            del g_full, g_skip
    core.reset()
    achieved = B_STEP * n / (k_us * 1e-6) / 1e9
    # env-only loop with on-device random actions
    core.step_random(50, 7, out)
    torch.cuda.synchronize(device)
    e0.record()
    core.step_random(1000, 7, out)
    e1.record()
    torch.cuda.synchronize(device)
    eo_us = e0.elapsed_time(e1) * 1e3 / 1000
    # persistent mode: up to 10 steps per launch, every step's outputs written (t2d_rollout_random)
    fo = core.rollout_random(T, 7)
    torch.cuda.synchronize(device)
    e0.record()
    for _ in range(50):
        core.rollout_random(T, 7, fo)
    e1.record()
    torch.cuda.synchronize(device)
    eof_us = e0.elapsed_time(e1) * 1e3 / (50 * T)
    del fo

    # ---- the policy-side hot kernels (the largest single kernels of the iteration): f32-MFMA roofline ------------
    # learner shape of the target's encoder: 2 frames x 4096 envs x 20 steps. ALGORITHMIC FLOPs = taps that read real
    # pixels (conv1 16 x 361, conv2 32 x 16 x 100 MACs per frame; backward = 2x conv2 + conv1 recompute + dW1). The
    # wave-per-frame kernels (launches below 16384 frames: the rollout's) issue the dense zero-padded products (16 x 49 x 9 and
    # 32 x 16 x 144 MACs: 1.42x, `issued_over_algorithmic`); the 16-frame kernels measured here issue conv2's real products only.
    stem_roof = None
    try:
        from active_tracking_rl_amd import fused
        enc = player.model.player1.encoder
        Ms = 2 * n * T
        xs_ = torch.randint(0, 5, (Ms, 169), device=device).float()
        ys_ = torch.empty((Ms, 512), device=device)
        dys_ = torch.randn((Ms, 512), device=device)
        prm = [enc.conv1.weight.detach(), enc.conv1.bias.detach(), enc.conv2.weight.detach(), enc.conv2.bias.detach()]

        def t_us(fn, launches=5, reps=4):
            """microseconds per launch, the launches replayed from a hipGraph (how the learner issues them) with HIP events
            around the replays; best of 3"""
            fn()
            torch.cuda.synchronize(device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(launches):
                    fn()
            g.replay()
            torch.cuda.synchronize(device)
            best = None
            for _ in range(3):
                e0.record()
                for _ in range(reps):
                    g.replay()
                e1.record()
                torch.cuda.synchronize(device)
                us = e0.elapsed_time(e1) * 1e3 / (reps * launches)
                best = us if best is None else min(best, us)
            del g
            return best
        w1c, w2c = prm[0].contiguous(), prm[2].contiguous()
        f_us = t_us(lambda: fused.stem_into(xs_, enc.conv1, enc.conv2, ys_))
        b_us = t_us(lambda: fused._stem_backward(xs_, ys_, dys_, w1c, prm[1], w2c, (prm[0].shape, prm[2].shape)))
        c1, c2, c1d, c2d = 16 * 361, 32 * 16 * 100, 16 * 49 * 9, 32 * 16 * 144
        f_tf = 2.0 * (c1 + c2) * Ms / (f_us * 1e-6) / 1e12
        b_tf = 2.0 * (2 * c2 + 2 * c1) * Ms / (b_us * 1e-6) / 1e12
        dense = float(c1d + c2d) / (c1 + c2)
        stem_roof = {"bound": "mfma", "kernel": "atr::k_stem_fwd16 / atr::k_stem_bwd16 + k_stem_reduce (v_mfma_f32_16x16x4_f32; 16 frames "
                                                  "per workgroup pass: the launch sizes from 16384 frames up since round 5)",
                     "frames": Ms, "fwd_us": f_us, "bwd_us": b_us,
                     "achieved": f_tf, "achieved_bwd": b_tf, "peak": 157.3, "unit": "TFLOP/s",
                     "frac": f_tf / 157.3, "frac_bwd": b_tf / 157.3, "issued_over_algorithmic": dense,
                     "note": "dense f32 MFMA peak (MI355X_MICROARCH.md); informational — `roofline` above stays the "
                             "env step kernel of SURVEY 8(d)"}
        del xs_, ys_, dys_
    except Exception as ex:
        stem_roof = {"error": repr(ex)}

    traffic, traffic_src, traffic_u8, traffic_u8_src, traffic_act, traffic_act_src = None, None, None, None, None, None
    for fname in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json"):
        try:   # HBM traffic of the same kernels from the committed rocprofv3 --pmc passes (cannot be taken inside this run)
            pm = json.load(open(os.path.join(ROOT, "profiles", fname)))
            if pm.get("n_envs") == n and traffic is None:
                traffic = pm["traffic_bytes_per_launch"]
                traffic_src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)" % fname
            if pm.get("n_envs") == n and traffic_act is None and pm.get("act_step") is not None:
                # (only the passes of the kernel in the form the timed region runs it: one gate tensor since round 4)
                if bool(pm["act_step"].get("one_gate_tensor", False)) == bool(one_gate):
                    traffic_act = pm["act_step"]["traffic_bytes_per_launch"]
                    traffic_act_src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, %s)" % (fname, pm["act_step"]["kernel"])
            if pm.get("n_envs") == n and traffic_u8 is None and pm.get("traffic_bytes_per_launch_u8") is not None:
                traffic_u8 = pm["traffic_bytes_per_launch_u8"]
                traffic_u8_src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, k_step2<OBS_U8>)" % fname
        except Exception:
            pass
    kernel_sum, rocprof_act_us = None, None
    for fname in ("r06_iteration_kernel_stats.txt", "r05_iteration_kernel_stats.txt", "r04_iteration_kernel_stats.txt", "r03_iteration_kernel_stats.txt"):
        try:   # one replayed iteration under rocprofv3 --kernel-trace --stats (tools/iter_profile.py), committed: the sum of its
            # kernel durations, and the in-iteration average of the env kernel the in-situ figure below must agree with
            tot_ms = its = None
            for ln in open(os.path.join(ROOT, "profiles", fname)):
                if ln.startswith("# total kernel time"):
                    tot_ms = float(ln.split()[4])
                if ln.startswith("# iterations"):
                    its = float(ln.split()[2])
                if "k_act_step<2, false, true" in ln and rocprof_act_us is None:
                    rocprof_act_us = {"us": float(ln.split()[-2]), "source": "profiles/%s" % fname}
            if tot_ms is not None and its:
                kernel_sum = {"ms": tot_ms / its, "source": "profiles/%s (rocprofv3 --kernel-trace --stats of "
                              "tools/iter_profile.py: total kernel time / iterations)" % fname}
                break
        except Exception:
            pass
    note_k = ("step_f32 / step_u8 (other_variants): the stand-alone step kernel alone — a hipGraph of 9 policy-shaped launches "
              "captured right after a generator pass (no k_gen inside) replayed 40x, HIP events on the launch stream around each "
              "replay; the generator pass is reported separately (avg_step_us_incl_generator)")
    f32_variant = {"kernel": "t2d::k_step2<..., OBS_F32_VEC4> (t2d_step: float32 observations, SURVEY 8(d) B_step = 1723 B)",
                   "in_timed_region": not getattr(player.env, "obs_u8", False) and not fused_in_region,
                   "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                   "traffic": traffic, "traffic_source": traffic_src, "bytes_per_launch": B_STEP * n, "avg_launch_us": k_us,
                   "avg_step_us_incl_generator": step_us, "achieved_incl_generator": B_STEP * n / (step_us * 1e-6) / 1e9}
    u8_variant = None if k8_us is None else {
        "kernel": "t2d::k_step2<..., OBS_U8> (t2d_step_u8: observations left as bytes, decoded by the policy stem's conv1 "
                  "load; SURVEY 8(d) B_step = 709 B)",
        "in_timed_region": bool(getattr(player.env, "obs_u8", False)) and not fused_in_region,
        "achieved": B_STEP_U8 * n / (k8_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": B_STEP_U8 * n / (k8_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": traffic_u8, "traffic_source": traffic_u8_src,
        "bytes_per_launch": B_STEP_U8 * n, "avg_launch_us": k8_us}
    act_variant = None
    if ka_us is not None:
        # interface-mandated env traffic only (709 B per env-step with byte observations) over the kernel's duration
        # where it runs; the figure that also counts the policy state the fused kernel moves goes under its own name
        dur = ka_situ_us if ka_situ_us else ka_us
        b8d = B_STEP_U8 if getattr(player.env, "obs_u8", False) else B_STEP
        act_variant = {
            "kernel": "t2d::k_act_step<%s> (atr_act_env_step: both players' LSTM cells + heads + draws + env step + "
                      "observation in one launch)" % ("OBS_U8" if getattr(player.env, "obs_u8", False) else "OBS_F32"),
            "in_timed_region": fused_in_region,
            "achieved": b8d * n / (dur * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": b8d * n / (dur * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "frac_8d": b8d * n / (dur * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "bytes_per_launch": b8d * n, "bytes_per_env_step": b8d,
            "avg_launch_us": dur,
            "avg_launch_us_source": ("in situ: (20-step rollout graph) - (the same graph without the k_act_step launches), "
                                     "/ 20, HIP events on the launch stream" if ka_situ_us else
                                     "alone: 9-launch graph (the timed region did not run this kernel)"),
            "in_situ": situ_detail, "alone_us": ka_us, "rocprof_in_iteration_us": rocprof_act_us,
            "traffic": traffic_act, "traffic_source": traffic_act_src,
            "policy_state_included": {
                "bytes_per_env_step": ka_bytes, "bytes_per_launch": ka_bytes * n,
                "achieved": ka_bytes * n / (dur * 1e-6) / 1e9, "frac_of_peak": ka_bytes * n / (dur * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "alone": {"avg_launch_us": ka_us, "frac_of_peak": ka_bytes * n / (ka_us * 1e-6) / 1e9 / HBM_PEAK_GBS},
                "note": "NOT SURVEY 8(d)'s figure: the env's 709 B + per player gate pre-activations read (one tensor: the "
                        "LSTMCell's two GEMMs are one product since round 4), c_prev read, h / c written, " +
                        ("(no activated gates: since round 5 the rollout keeps the gate GEMM's output for the learner instead), "
                         if keep_pre else "activated gates written for the learner, ") +
                        "the masked hidden row for the next step's GEMM, + actions + previous done"}}
    # `roofline` = the env kernel of the TIMED REGION at this batch size; the other variants ride along, each with its flag
    variants = {"step_f32": f32_variant, "step_u8": u8_variant, "act_step": act_variant}
    pick = "act_step" if (fused_in_region and act_variant) else ("step_u8" if (u8_variant and u8_variant["in_timed_region"])
                                                                 else "step_f32")
    roofline = dict(variants[pick])
    roofline.update(bound="hbm", variant=pick, note=note_k,
                    other_variants={k: v for k, v in variants.items() if k != pick and v is not None})
    line = {
        "metric": "env steps/sec, Track2D-BlockPartialPZR-v0 @4096 envs, 1/2/4/8 GPU",
        "value": value, "unit": "env steps/s", "n_gpus": world, "steps": steps, "warmup": a.warmup,
        "warmup_effective": warm, "steps_requested": a.steps, "repeats": repeats, "spread": weak["spread"],
        "ms_per_step": weak["ms_per_step"], "ms_per_iteration": weak["ms_per_iteration"],
        "timed_gpu_seconds": weak["timed_gpu_seconds"], "repeats_run": weak["repeats"], "kernel_sum_per_iteration": kernel_sum,
        "shards": shards, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "ranks": comm["ranks"], "rccl_ranks": comm["ranks"] if (comm["backend"] == "nccl" or world == 1) else None,
        "devices": comm["devices"], "dist_backend": comm["backend"],
        "allreduce": {"mode": "captured in the update graph" if capture_allreduce_default() else
                              "eager, between the learner's graph and the update graph",
                      "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default"),
                      "basis": "profiles/r04_multirank_1gpu.txt (1-rank RCCL group, collective forced)"},
        "allreduce_us": weak["allreduce_us"], "allreduce_elems": weak["allreduce_elems"],
        "weak": {k: weak[k] for k in ("value", "ms_per_step", "envs_per_gpu", "global_envs", "allreduce_us")},
        "strong": {k: strong[k] for k in ("value", "ms_per_step", "envs_per_gpu", "global_envs", "allreduce_us",
                                          "spread") if k in strong},
        "dtype": "u8 map bits / f64 reward -> f32 obs+reward; policy fp32", "data": "synthetic",
        "config": {"workload": "%s, %d envs/GPU x %d GPU, %s tracker+target, train-mode -1, 20-step A3C rollouts "
                               "(policy fwd + HIP env step + loss/backward + grad all-reduce + SharedAdam)"
                               % (a.env, n, world, a.network),
                   "global_envs": n_total, "rollout": T, "hipgraph": graphed,
                   "episode_source": "device Philox generators (the reference's sequential algorithms on a counter-based bit source: "
                                     "bit-exact against the oracle's PHILOX mode, distribution-exact against the reference); the "
                                     "reference's own numpy-stream episodes also run on the device, draw for draw, for every target "
                                     "mode (t2d_np_attach: a parity mode, not the timed path)",
                   "obs": "u8 (t2d_step_u8 -> atr_stem_*_u8)" if getattr(player.env, "obs_u8", False) else "f32",
                   "gemm_algos": {"torch": "TunableOp picks from tunableop_gfx950.csv" if tuned else "library default",
                                  "hipblaslt_direct": _lt_status()}, "parallelism": "dp%d (env shards, 1 grad all-reduce/update)" % world},
        "roofline": roofline,
        "env_only": {"value": n * world / (eo_us * 1e-6), "unit": "env steps/s", "us_per_launch": eo_us,
                     "note": "same kernel, on-device random actions, one launch per batched step, per-rank x ranks",
                     "fused_value": n * world / (eof_us * 1e-6), "fused_us_per_step": eof_us,
                     "fused_gbs": B_STEP * n / (eof_us * 1e-6) / 1e9,
                     "fused_note": "t2d_rollout_random: up to 10 env steps per launch (state in registers, map rows re-read "
                                   "through L1/L2), all %d steps' observations/rewards/done written; bit-identical to the "
                                   "per-step launches" % T},
        "policy_stem": stem_roof,
    }
    sched_note = {"synchronous": "rollout, learner, all-reduce, update in sequence (train.GraphedIteration): every gradient "
                                 "applied to the weights it was computed on",
                  "pipelined": "rollout i+1 on a second HIP stream while learner + all-reduce + update i run "
                               "(train.PipelinedIteration; main.py's default): the same kernels and the same work per "
                               "iteration, every gradient applied exactly one update late — the bounded form of the "
                               "reference's Hogwild worker asynchrony (main.py:86-116, train.py:71-95). From 1024 envs "
                               "per GPU up the learner's grouped weight-gradient GEMM is captured in its co-run form (one "
                               "workgroup per CU: slower alone, but the other replica's rollout keeps moving beside it)"}
    line["schedule"] = first
    line["schedule_note"] = sched_note[first]
    line["config"]["schedule"] = first
    if want_pipelined:
        # a watchdog around the second schedule: if it does not come back (an N>1 run must never hang the driver), rank 0
        # prints the line as assembled so far — the synchronous numbers — and every rank leaves
        import threading

        def give_up():
            if rank == 0:
                line["pipelined"] = {"error": "the pipelined measurements did not complete within %.0f s; value is the "
                                              "synchronous schedule's" % a.pipelined_timeout}
                print(json.dumps(line), flush=True)
            os._exit(0)
        dog = threading.Timer(a.pipelined_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            res1, _, _, _ = measure_set("pipelined", keep_player=False)
            dog.cancel()
            w1 = res1["weak"]
            if w1["value"] < weak["value"]:      # the line's value is the faster schedule's; the other one rides along
                line["pipelined"] = {"value": w1["value"], "ms_per_step": w1["ms_per_step"],
                                     "ms_per_iteration": w1["ms_per_iteration"], "spread": w1["spread"],
                                     "shards": res1["shards"], "strong": res1["strong"], "note": sched_note["pipelined"],
                                     "stream_trials_ms_per_iteration": w1["stream_trials_ms_per_iteration"]}
                raise StopIteration
            line["synchronous"] = {"value": weak["value"], "ms_per_step": weak["ms_per_step"],
                                   "ms_per_iteration": weak["ms_per_iteration"], "spread": weak["spread"],
                                   "timed_gpu_seconds": weak["timed_gpu_seconds"], "repeats_run": weak["repeats"],
                                   "allreduce_us": weak["allreduce_us"], "shards": shards, "weak": line["weak"],
                                   "strong": line["strong"], "note": sched_note["synchronous"]}
            line.update(value=w1["value"], spread=w1["spread"], ms_per_step=w1["ms_per_step"],
                        ms_per_iteration=w1["ms_per_iteration"], timed_gpu_seconds=w1["timed_gpu_seconds"],
                        repeats_run=w1["repeats"], shards=res1["shards"], allreduce_us=w1["allreduce_us"],
                        stream_trials_ms_per_iteration=w1["stream_trials_ms_per_iteration"],
                        weak={k: w1[k] for k in ("value", "ms_per_step", "envs_per_gpu", "global_envs", "allreduce_us")},
                        strong={k: res1["strong"][k] for k in ("value", "ms_per_step", "envs_per_gpu", "global_envs",
                                                               "allreduce_us", "spread") if k in res1["strong"]},
                        schedule="pipelined", schedule_note=sched_note["pipelined"])
            line["config"]["schedule"] = "pipelined"
        except StopIteration:
            pass
        except Exception as ex:
            dog.cancel()
            line["pipelined"] = {"error": repr(ex)}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            # separate process: the baseline forks Hogwild workers, which must not inherit autograd/HIP state.
            # --suite = BASELINE.md section 3: the headline config and config 1 (BlockPartialRam, maze-lstm, --aux none,
            # train-mode 0), each 16 workers + 1 evaluator pinned one per physical core, + the env alone on 1 / 16 procs
            import subprocess
            r = subprocess.run([sys.executable, "-m", "oracle.cpu_a3c", "--suite", "--workers", "16", "--seconds",
                                str(a.cpu_seconds)], cwd=ROOT, capture_output=True, text=True, timeout=900,
                               env=dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES=""))
            rows = json.loads(r.stdout.strip().splitlines()[-1])

            def entry(cb):
                return {"value": cb["value"], "unit": "env steps/s", "cores": cb["cores"], "kind": "port",
                        "config": cb["config"], "cpu_model": cb["cpu_model"], "physical_cores": cb["physical_cores"],
                        "logical_cpus": cb["host_cpus"], "evaluator_cores": cb["evaluator_cores"], "pinning": cb["pinning"],
                        "env_only_1proc": cb["env_only_1proc"], "env_only_16proc": cb["env_only_16proc"],
                        "sample": "%d Hogwild workers + %d evaluator x %.0f s of reference-shaped A3C (%s, %s, --aux %s, "
                                  "train-mode %d; 1 oracle env + batch-1 policy on torch-CPU per worker, <=20-step "
                                  "rollouts, SharedAdam); env alone (random actions): %.0f steps/s on 1 process, %.0f on 16"
                                  % (cb["cores"], cb["evaluator_cores"], cb["seconds"], cb["env"], cb["network"], cb["aux"],
                                     cb["train_mode"], cb["env_only_1proc"], cb["env_only_16proc"])}
            line["cpu_baseline"] = entry(rows[0])              # the headline workload's CPU form
            line["cpu_baselines"] = [entry(cb) for cb in rows]  # + BASELINE config 1
        except Exception as ex:  # the GPU numbers must still be printed
            line["cpu_baseline"] = {"value": None, "error": repr(ex)}
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
