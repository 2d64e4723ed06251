"""train — the worker loop of train.py:15-113, one process per GPU instead of 16 Hogwild CPU workers.

    train(rank, args, shared_model, optimizer, train_modes, n_iters, env=None)

keeps the reference signature. Semantics per iteration are the reference's (sync weights -> <= num_steps env
steps -> optimize -> log), with these MI355X-first substitutions:
  * each rank owns `args.num_envs` envs (a shard of the global batch; Philox streams are keyed by the global env
    id so the union of shards equals the unsharded batch) stepped by the HIP kernels;
  * `shared_model` is this rank's replica (weights stay bit-identical across ranks because every rank applies
    the same all-reduced gradient), so `load_state_dict(shared_model.state_dict())` (train.py:71) is a no-op
    and is skipped;
  * the rollout is always `num_steps` long — finished envs auto-reset in the step launch (see player_util).
`train_modes` / `n_iters` may be plain lists (single process) or mp.Manager lists as in main.py:103-105.
"""
import os
import time

import torch

from .environment import create_env
from .model import build_model
from .player_util import Agent
from .shared_optim import SharedAdam, make_optimizer  # noqa: F401  (SharedAdam re-exported for callers)


def default_args(**over):
    """The flag defaults of main.py:16-50 as a namespace (plus num_envs / max_grad_norm / obs_u8: keep the training
    env's observations as bytes up to the policy's conv stem where the kernels allow it, see environment.VecEnv)."""
    import argparse
    d = dict(lr=0.001, gamma=0.9, tau=1.00, entropy=0.01, entropy_target=0.2, seed=1, workers=1, num_steps=20,
             test_eps=100, env='Track2D-BlockPartialPZR-v0', env_base='Track2D-BlockPartialNav-v0', optimizer='Adam',
             amsgrad=True, load_model_dir=None, log_dir='logs/', network='tat-maze-lstm', aux='reward', gpu_ids=[0],
             obs='img', single=False, gray=False, crop=False, inv=False, rescale=False, render=False,
             shared_optimizer=True, split=False, train_mode=-1, stack_frames=1, input_size=80, rnn_out=128,
             sleep_time=0, max_step=150000, init_step=-1, adv_step=None, num_envs=4096, max_grad_norm=None, obs_u8=True)
    d.update(over)
    return argparse.Namespace(**d)


def select_params(model, train_mode):
    """train.py:39-44: which half of the two-player model the optimizer owns."""
    if train_mode == 0:
        return list(model.player0.parameters())
    if train_mode == 1:
        return list(model.player1.parameters())
    return list(model.parameters())


def make_player(args, device, rank=0, world_size=1, env=None, model=None, optimizer=None):
    """Build env shard + replica + optimizer + Agent for one rank."""
    if device.type == 'cuda':
        from . import gemm_tuning
        gemm_tuning.enable()                  # read-only TunableOp picks for the policy GEMMs (no-op without the file)
    torch.manual_seed(args.seed)          # same init on every rank (replicas must start identical)
    if env is None:
        env = create_env(args.env, args, num_envs=args.num_envs, device=str(device),
                         env_id_base=rank * args.num_envs)
    if model is None:
        model = build_model(env.observation_space, env.action_space, args, device).to(device)
    model.train()
    if optimizer is None:
        optimizer = make_optimizer(select_params(model, args.train_mode), args)
    torch.manual_seed(args.seed + rank)   # per-rank action sampling (train.py:20)
    if device.type == 'cuda':
        torch.cuda.manual_seed(args.seed + rank)
    player = Agent(model, env, args, None, device)
    player.w_entropy_target = args.entropy_target
    player.reset()
    return player, optimizer


def rollout(player, num_steps, fast=True):
    """train.py:79-88 without the early break (done is a per-env mask). fast=True: actor/learner split — the policy
    steps without autograd (action_rollout) and Agent.loss_recompute re-evaluates the stored rollout time-batched;
    fast=False: the reference-shaped per-step autograd path (action_train + loss)."""
    if hasattr(player.model, "cache_dense"):
        player.model.cache_dense(True)   # expand conv weights once per rollout (released in compute_grads)
    if fast:
        player.begin_rollout(num_steps)
        for _ in range(num_steps):
            player.action_rollout()
        player.end_rollout()
    else:
        player.update_rnn_hiden()
        for _ in range(num_steps):
            player.action_train()
    # A rollout that is not a whole number of generator stamp cycles restarts the stamps (so that a captured rollout
    # replays consistently); otherwise forked generator launches stay in flight under the learner's kernels.
    if hasattr(player.env, "flush") and num_steps % getattr(player.env, "generator_cycle", 1) != 0:
        player.env.flush()


def capture_allreduce_default():
    """Whether the gradient all-reduce is captured INSIDE the update graph (one graph replay per update: all-reduce, clip,
    optimizer step) or issued eagerly between the learner's graph and the update graph. ATR_CAPTURE_ALLREDUCE=1 / 0 overrides;
    the default is the measured winner (profiles/r04_multirank_1gpu.txt)."""
    v = os.environ.get("ATR_CAPTURE_ALLREDUCE")
    return CAPTURE_ALLREDUCE_DEFAULT if v is None else v == "1"


CAPTURE_ALLREDUCE_DEFAULT = False
PREGROW_MODE = os.environ.get("ATR_PREGROW_MODE", "inline")    # pipelined schedule: "inline" on the learner's stream | "fork" | "off"
CORUN_MIN_ENVS = 1024      # PipelinedIteration: the learner's dW kernel in its one-workgroup-per-CU form from this shard size up


def clip_flat_grad_(optimizer, max_norm, eps=1e-6):
    """torch.nn.utils.clip_grad_norm_(params, max_norm) (what player_util.py:157 asks for; a no-op in the reference, SURVEY
    quirk 5, hence off unless --max-grad-norm is given) on the flat gradient bucket: the total norm over all parameters is the
    norm of the one flat tensor (the bucket's padding stays zero). No host synchronisation: capturable in the update graph."""
    bucket = getattr(optimizer, "bucket", None)
    if not max_norm or bucket is None:
        return
    g = bucket.grad
    coef = (float(max_norm) / (g.norm(2) + eps)).clamp(max=1.0)
    g.mul_(coef)


class GraphedIteration(object):
    """One A3C iteration (20-step rollout -> loss -> backward | all-reduce | SharedAdam) replayed as two hipGraphs.

    The rollout is launch-bound in eager mode (~60 small kernels per env step); capturing it removes the host
    from the loop. The gradient all-reduce stays an eager RCCL call between the two graphs, so the captured
    regions contain no collective. State carried between iterations (obs, LSTM state, done, episode lengths) lives
    in static tensors that the captured region reads first and writes last; the env state itself is device-resident
    inside the HIP library.

    The training mode (which player's loss is differentiated, player_util.py:147-152) is a constant of the captured
    loss, so there is ONE rollout graph PER MODE, captured the first time `run(mode)` sees it: the evaluator's
    `train_modes[rank]` schedule (test.py:84-92: tracker-only until --init-step) selects the graph per iteration.
    All graphs read and write the same carry tensors, so switching modes does not disturb the env/LSTM state."""

    def __init__(self, player, optimizer, args, warmup=2, fast=True, mode=None, keep_warmup_updates=False):
        """mode: the training mode of the first iterations (main.py starts in mode 0 under --init-step); the eager
        warm-up iterations and the first captured graph use it. The warm-up iterations are real rollouts + updates run to
        settle allocations before the capture; unless keep_warmup_updates is set their effect on the parameters and the
        optimizer state (step counter, moments) is rolled back, so that iteration 0 of the run is the first replay."""
        self.player, self.optimizer, self.args, self.fast = player, optimizer, args, fast
        self.mode0 = args.train_mode if mode is None else int(mode)
        dev = player.device
        snap = None if keep_warmup_updates else self._optimizer_tensors()
        saved = [t.clone() for t in snap] if snap is not None else None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        if saved is not None:
            with torch.no_grad():
                for t, v in zip(snap, saved):
                    t.copy_(v)
        self.carry = dict(state=player.state.clone(), hxs=player.hxs.detach().clone(),
                          cxs=player.cxs.detach().clone(), done=player.done.clone(), eps_len=player.eps_len.clone())
        self.g_rolls, self.stats_by_mode = {}, {}
        self._capture(self.mode0)
        self.capture_allreduce = capture_allreduce_default()
        self.g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_opt, capture_error_mode="thread_local"):
            if self.capture_allreduce:
                player.allreduce_grads(optimizer)
            clip_flat_grad_(optimizer, getattr(args, "max_grad_norm", None))
            optimizer.step()
        if saved is not None:          # (capturing a graph does not execute it; restore anyway in case a backend ran it)
            with torch.no_grad():
                for t, v in zip(snap, saved):
                    t.copy_(v)

    def _optimizer_tensors(self):
        """Every tensor an update writes: the flat parameter bucket and the optimizer's own state tensors."""
        opt = self.optimizer
        bucket = getattr(opt, "bucket", None)
        if bucket is None:
            return None
        seen, out = set(), []
        for t in [bucket.flat] + [v for v in vars(opt).values() if isinstance(v, torch.Tensor)]:
            if t.data_ptr() not in seen and t.numel() > 0:
                seen.add(t.data_ptr())
                out.append(t)
        return out

    def _bind_carry(self):
        p = self.player
        p.state, p.hxs, p.cxs = self.carry["state"], self.carry["hxs"], self.carry["cxs"]
        p.done, p.eps_len = self.carry["done"], self.carry["eps_len"]

    def _capture(self, mode):
        player, args = self.player, self.args
        if hasattr(player.env, "flush"):
            player.env.flush()           # no generator launch in flight and stamp 0 when the capture starts
        torch.cuda.synchronize(player.device)
        g = torch.cuda.CUDAGraph()
        # thread_local: an RCCL watchdog thread polling events must not invalidate the capture
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._bind_carry()
            player.carry_out, player.carry_written = self.carry, ()   # the rollout's epilogue kernel publishes the carry
            rollout(player, args.num_steps, fast=self.fast)
            # (growing the next passes' Maze maps on a forked stream under the learner — env.pregrow(fork=True) here — was
            # measured: -2 % on configs[3] / [4]: on one stream the growth only moves from the pass to the fork, and the fork /
            # join inside the graph costs more than the overlap returns. The pipelined schedule has a second stream to put it on.)
            if PREGROW_MODE == "fork" and hasattr(player.env, "pregrow"):
                player.env.pregrow(fork=True)
            stats = player.compute_grads(self.optimizer, mode)
            if hasattr(player.env, "generator_join"):
                player.env.generator_join()   # the env's forked generator launches end inside the captured region
            for k, src in (("state", player.state), ("hxs", player.hxs.detach()), ("cxs", player.cxs.detach()),
                           ("done", player.done), ("eps_len", player.eps_len)):
                if k in player.carry_written:
                    continue
                if self.carry[k].data_ptr() != src.data_ptr():    # (the epilogue kernel advances eps_len in place)
                    self.carry[k].copy_(src)
            player.carry_out = None
        self._bind_carry()
        self.g_rolls[mode], self.stats_by_mode[mode] = g, stats
        self.g_roll, self.stats = g, stats          # the most recently captured pair (kept for callers/tools)
        return g

    def _eager(self):
        rollout(self.player, self.args.num_steps, fast=self.fast)
        self.player.optimize(None, self.optimizer, self.player.model, self.mode0, self.player.device)

    def burn_in(self, iters, mode=None):
        """`iters` iterations whose updates are DISCARDED (weights, optimizer state and step counter restored): only the env
        shard moves on. Why a driver wants this: every env of a fresh shard starts an episode at step 0 together, so for the first
        ~25 rollouts the batch is perfectly correlated in episode phase (all trackers next to their targets, no episode end in
        any of them), and with 1024+ envs per update the first Adam steps fit that one phase. Measured on
        Track2D-MazePartialNav-v0 / 1024 envs / train-mode 0 over 8 seeds x 3000 iterations: 3 of 8 runs reach the +0.63
        plateau from a synchronised start, 6 of 8 after 300 discarded iterations (profiles/r05_learning_seeds_*.txt). The
        reference's 16 asynchronous one-env workers drift apart on their own (train.py:71-95)."""
        if iters <= 0:
            return
        tensors = self._optimizer_tensors()
        if tensors is None:
            raise RuntimeError("burn_in needs the flat-bucket optimizer (weights and optimizer state as a few tensors to restore); "
                               "this optimizer has none: its updates would be KEPT")
        saved = [t.clone() for t in tensors]
        n0 = self.player.n_steps
        for _ in range(int(iters)):
            self.run(mode)
        torch.cuda.synchronize(self.player.device)
        if saved is not None:
            with torch.no_grad():
                for t, v in zip(tensors, saved):
                    t.copy_(v)
        self.player.n_steps = n0
        torch.cuda.synchronize(self.player.device)

    def run(self, mode=None):
        mode = self.mode0 if mode is None else int(mode)
        g = self.g_rolls.get(mode)
        if g is None:
            g = self._capture(mode)
        g.replay()
        if not self.capture_allreduce:
            self.player.allreduce_grads(self.optimizer)
        self.g_opt.replay()
        self.player.n_steps += self.args.num_steps
        self.stats = self.stats_by_mode[mode]
        return self.stats


_hip, _masked_streams = None, {}
MODEL_SWITCHES = ("fused_sampling", "fused_actor_step", "fused_env_step", "pair_gemm_max_rows", "mfma_step_min_rows",
                  "cat_gate_gemm", "cat_gemm_min_rows", "coop_step", "coop_max_rows", "gate_cell_kernel", "gate_cell_min_rows")


def device_cus(device):
    """Compute units of `device` as the HIP runtime enumerates them (hipDeviceProp.multiProcessorCount: 256 on an MI355X)."""
    return int(torch.cuda.get_device_properties(device).multi_processor_count)


def cu_partition(device):
    """(first CU, CUs) of the rollout half and of the learner half of an even CU split, or None when this part cannot be
    split evenly over its shader engines: the runtime deals a mask's bits round-robin over the chip's 32 shader engines, so
    only runs that are multiples of 32 CUs give every engine the same share (cu_masked_stream)."""
    total = device_cus(device)
    if total < 64 or total % 64 != 0:
        return None
    return (total // 2, total // 2), (0, total // 2)


def cu_masked_stream(device, first_cu, n_cus, total_cus=None):
    """A HIP stream whose kernels are dispatched only to compute units [first_cu, first_cu + n_cus) of the runtime's CU
    enumeration (hipExtStreamCreateWithCUMask), wrapped for torch. The runtime deals the mask's bits round-robin over the
    chip's 32 shader engines (bit k = CU k / 32 of engine k % 32), so a run of 32 m bits is m of the 8 CUs of EVERY engine of
    every XCD: both halves of a split see all eight L2s and stay balanced under the round-robin workgroup placement; counts
    that are not multiples of 32 leave some engines a CU short and the whole stream waits for those (measured: 120 or 136 CUs
    are slower than 96). total_cus: the device's CU count (read from the device properties when None). One stream per
    (device, range) per process: each masked stream takes a hardware queue of its own, and a process that oversubscribes
    the queues gets time-sliced (measured 2-3x slower iterations)."""
    global _hip
    import ctypes as C
    if total_cus is None:
        total_cus = device_cus(device)
    if n_cus <= 0 or first_cu < 0 or first_cu + n_cus > total_cus:
        raise ValueError("CU range [%d, %d) does not fit a %d-CU device" % (first_cu, first_cu + n_cus, total_cus))
    key = (str(device), int(first_cu), int(n_cus))
    if key in _masked_streams:
        return _masked_streams[key]
    if _hip is None:
        path = None
        for ln in open("/proc/self/maps"):
            if "libamdhip64" in ln:
                path = ln.split()[-1]
                break
        _hip = C.CDLL(path or "libamdhip64.so")
        _hip.hipExtStreamCreateWithCUMask.restype = C.c_int
        _hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    words = (total_cus + 31) // 32
    mask = (C.c_uint32 * words)()
    for k in range(first_cu, min(first_cu + n_cus, total_cus)):
        mask[k // 32] |= 1 << (k % 32)
    st = C.c_void_p()
    with torch.cuda.device(device):
        rc = _hip.hipExtStreamCreateWithCUMask(C.byref(st), words, mask)
    if rc != 0 or not st.value:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed (%d)" % rc)
    _masked_streams[key] = torch.cuda.ExternalStream(st.value, device=device)
    from . import fused
    fused.register_stream_cus(_masked_streams[key], n_cus)       # (launches whose workgroups wait for each other size their grid by it)
    return _masked_streams[key]


class _BucketOnly(object):
    """What Agent.compute_grads needs of an optimizer: the flat bucket the gradients land in."""

    def __init__(self, bucket):
        self.bucket = bucket


class PipelinedIteration(object):
    """A3C iterations with the rollout of iteration i + 1 running WHILE the learner of iteration i runs: two HIP streams, each
    replaying hipGraphs, events between them — no host synchronisation, no fork / join inside a graph.

    The reference's workers are asynchronous by construction (Hogwild: a worker pulls the shared weights, spends a rollout +
    backward on them while 15 others update the shared model, then pushes its gradient: main.py:86-116, train.py:71-95,
    utils.py:36-44). The synchronous schedule of GraphedIteration (every gradient applied to exactly the weights it was
    computed on) is one member of that family; this is another — ONE update of delay, fixed:

        weights of rollout i = theta_{i-1};   g_i = gradient at theta_{i-1};   theta_{i+1} = update(theta_i, g_i)

    Why: at the strong-scaling shard sizes an iteration is two chains of small dependent launches (rollout: 4 per env step;
    learner: ~30 after the launch diet), each launch ~4.5 us before it does anything; the two chains of DIFFERENT iterations have no dependence on
    each other except through the weights, so they overlap (measured: two graph replays on two streams take 0.70 x their
    serial time at 512 envs, 0.81 x at 1024, 0.91 x at 4096 — tools/overlap_probe.py), and with N > 1 the gradient all-reduce
    runs under the next rollout as well.

    Mechanics. Two replicas of the policy (M0, M1: own flat weight buffers F0, F1, own rollout / activation stores) over
    ONE env shard; the optimizer owns the master weights theta. Iteration i uses replica k = i & 1; one call of run() is one
    PHASE of the pipeline:
        stream L:  [wait R(i-1)]  L_k': loss + backward through replica k' = (i-1) & 1's stores with F_k' -> gradient bucket
                                  all-reduce (eager, RCCL) ; O_k': optimizer step on theta, then theta -> F_k'
        stream R:  [wait O(i-2)]  R_k: rollout i with F_k -> stores of k; carry (obs, LSTM state, done) handed to the next rollout
    F_k is next read by rollout i + 2, which waits for O(i); rollout i + 1 reads F_{1-k}, written by O(i-1): every kernel sees
    exactly the weights of the schedule above, whatever the timing (serial=True replays the same graphs in program order on
    one stream: bit-identical weights, tests/test_drivers_gpu.py). finish() issues the learner still owed and joins both
    streams; sync() only joins (a phase boundary). tune_streams() picks the stream pair — and whether each chain gets its own
    half of the CUs — by trial."""

    def __init__(self, player, optimizer, args, warmup=2, mode=None, serial=False):
        from .player_util import Agent
        from .shared_optim import FlatParams
        self.args, self.optimizer, self.master = args, optimizer, player
        self.mode0 = args.train_mode if mode is None else int(mode)
        self.serial = bool(serial)
        # Whole-map ('Full') observations go through F.conv2d, i.e. MIOpen's kernels: replayed next to another stream's graph
        # they were observed to hang the device (their workgroups wait on each other and need the chip to themselves). Those
        # ids (no BASELINE configuration uses them) run the same schedule in program order on one stream.
        from .model import CNN_maze
        if any(isinstance(m, CNN_maze) and not m.small for m in player.model.modules()):
            self.serial = True
        dev = self.dev = player.device
        env = player.env
        # eager warm-up on the master (allocator, GEMM workspaces), its updates rolled back
        tensors = [optimizer.bucket.flat] + [v for v in vars(optimizer).values() if isinstance(v, torch.Tensor)]
        saved = [t.clone() for t in tensors]
        for _ in range(warmup):
            rollout(player, args.num_steps)
            player.optimize(None, optimizer, player.model, self.mode0, dev)
        torch.cuda.synchronize(dev)
        with torch.no_grad():
            for t, v in zip(tensors, saved):
                t.copy_(v)
        # the two replicas: same architecture, weights re-homed in flat buffers laid out like the optimizer's bucket
        self.players, self.buckets = [], []
        for k in range(2):
            m = build_model(env.observation_space, env.action_space, args, dev).to(dev)
            m.load_state_dict(player.model.state_dict())
            m.train()
            for attr, val in vars(player.model).items():     # instance-level kernel switches set on the master (A/B runs:
                if attr in MODEL_SWITCHES:                    # bench.py --actor-step) hold for the replicas that do the work
                    setattr(m, attr, val)
            if getattr(player.model, "_lt_ws", None) is not None:       # library-GEMM scratch of this replica's own chain of
                m._lt_ws = torch.empty_like(player.model._lt_ws)        # launches (allocated here, outside any capture)
            bucket = FlatParams(select_params(m, args.train_mode))
            assert bucket.flat.numel() == optimizer.bucket.flat.numel()
            # a draw stream of its own: the learner's bootstrap step reads the replica's counter, which only the replica's
            # own rollouts advance (they wait for this learner), so its draws do not depend on the timing of the other stream
            from . import fused
            m._sampler = fused.ActionSampler(dev, seed=(int(torch.initial_seed()) + 7919 * (k + 1)) & 0xFFFFFFFFFFFFFFFF)
            a = Agent(m, env, args, None, dev)
            a.w_entropy_target = player.w_entropy_target
            self.players.append(a)
            self.buckets.append(bucket)
        self.carry = dict(state=player.state.clone(), hxs=player.hxs.detach().clone(), cxs=player.cxs.detach().clone(),
                          done=player.done.clone(), eps_len=player.eps_len.clone())
        # (stream priorities were measured and make no difference to the overlap: ATR_PIPE_PRIO=1 / 2 raises the rollout's /
        # the learner's stream for experiments; what matters is which hardware queues and CUs the pair gets — tune_streams)
        prio = int(os.environ.get("ATR_PIPE_PRIO", "0"))
        self.sR = torch.cuda.Stream(device=dev, priority=-1 if prio == 1 else 0)
        self.sL = torch.cuda.Stream(device=dev, priority=-1 if prio == 2 else 0)
        part = os.environ.get("ATR_PIPE_CU_SPLIT")        # force a CU partition: this many CUs for the rollout stream, the
        self.cu_split = int(part) if part else 0          # rest for the learner's (tune_streams tries 128 / 128 by itself)
        if self.cu_split:
            total = device_cus(dev)
            self.sR = cu_masked_stream(dev, total - self.cu_split, self.cu_split, total)
            self.sL = cu_masked_stream(dev, 0, total - self.cu_split, total)
        self.ev_r = [torch.cuda.Event() for _ in range(2)]
        self.ev_o = [torch.cuda.Event() for _ in range(2)]
        # The learner's grouped weight-gradient GEMM is captured in its co-run form (one workgroup per CU: fused.gemm_tn_corun)
        # from 1024 envs up, where learner and rollout share the whole chip: it runs a fifth of the learner's time, its
        # workgroups live for hundreds of microseconds, and two of them per CU leave the other replica's short rollout kernels
        # queueing (4096 envs: 15.75 -> 16.6-16.75 M env steps/s, 2048: 13.95 -> 14.5, 1024: 11.3 -> 11.7). Below that the CU
        # partition of tune_streams does the separating and the kernel keeps its full occupancy (512 envs: 9.1 M against 8.4).
        # ATR_PIPE_CORUN=0 / 1 overrides.
        env_corun = os.environ.get("ATR_PIPE_CORUN")
        self.corun = (int(env_corun) != 0) if env_corun is not None else (int(args.num_envs) >= CORUN_MIN_ENVS and not self.cu_split)
        self.capture_allreduce = capture_allreduce_default()
        self.pending = None       # (replica, learner graph) of the rollout whose learner has not been issued yet
        self.graphs = {}          # (mode, k) -> (rollout graph, learner graph, stats)
        self.g_opt = []
        for k in range(2):        # O_k: the update on theta, then theta -> F_k
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                if self.capture_allreduce:
                    player.allreduce_grads(optimizer)
                clip_flat_grad_(optimizer, getattr(args, "max_grad_norm", None))
                optimizer.step()
                self.buckets[k].flat.copy_(optimizer.bucket.flat)
            self.g_opt.append(g)
        with torch.no_grad():
            for t, v in zip(tensors, saved):
                t.copy_(v)
            for b in self.buckets:
                b.flat.copy_(optimizer.bucket.flat)
        self.i = 0
        self._coop_wg = None
        self._coop_master = bool(getattr(player.model, "coop_step", False))     # (the master's switch: ATR_COOP_STEP / A-B runs)
        self._set_coop_grid()
        for k in range(2):
            self._capture(self.mode0, k)
        torch.cuda.synchronize(dev)

    def _set_coop_grid(self):
        """The cooperative rollout step (model._act_step -> fused.coop_env_step) launches one workgroup per CU of the stream it
        RUNS on, and its workgroups wait for each other at barriers: every one of them must be resident at the same time. That
        holds when the rollout has its CUs to itself — a CU-masked rollout stream (the partition tune_streams tries), or the
        one-stream (serial) form — and NOT on a chip shared with the learner's stream, whose long-lived workgroups hold CUs the
        step's last workgroups are waiting for (measured: barrier time-outs). So: on a masked rollout stream the replicas are
        told its CU count (a rollout graph captured for the whole chip must not be replayed on half of it), on a shared pair
        they keep the four-launch step. Returns True when the setting changed (graphs captured earlier are then stale)."""
        from . import fused
        masked = int(self.sR.cuda_stream) in fused._stream_cus
        wg = fused.stream_cus(self.dev, self.sR) if (masked or self.serial) else 0
        changed = wg != self._coop_wg
        self._coop_wg = wg
        for a in self.players:
            a.model.coop_workgroups = wg if wg else None
            a.model.coop_step = bool(wg) and self._coop_master
        return changed

    def _use_streams(self, sR, sL):
        """Switch the stream pair; rollout graphs whose cooperative step was sized for another CU count are captured again."""
        self.finish()
        torch.cuda.synchronize(self.dev)
        self.sR, self.sL = sR, sL
        if self._set_coop_grid() and self._coop_master:
            for (mode, k) in list(self.graphs.keys()):
                self._capture(mode, k)
            torch.cuda.synchronize(self.dev)

    def _bind_carry(self, p):
        p.state, p.hxs, p.cxs = self.carry["state"], self.carry["hxs"], self.carry["cxs"]
        p.done, p.eps_len = self.carry["done"], self.carry["eps_len"]

    def _capture(self, mode, k):
        p, args = self.players[k], self.args
        torch.cuda.synchronize(self.dev)
        if hasattr(p.env, "flush"):
            p.env.flush()
        torch.cuda.synchronize(self.dev)
        g_r = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_r, capture_error_mode="thread_local"):
            self._bind_carry(p)
            p.carry_out, p.carry_written = self.carry, ()             # the rollout's epilogue kernel publishes the carry
            rollout(p, args.num_steps)
            for key, src in (("state", p.state), ("hxs", p.hxs.detach()), ("cxs", p.cxs.detach()), ("done", p.done),
                             ("eps_len", p.eps_len)):
                if key in p.carry_written:
                    continue
                if self.carry[key].data_ptr() != src.data_ptr():
                    self.carry[key].copy_(src)
            p.carry_out = None
            if hasattr(p.env, "generator_join"):
                p.env.generator_join()        # the env's forked generator launches (opt-in) end inside the captured region
        # the learner of THIS rollout reads the replica's own end-of-rollout tensors (last observation slot, LSTM state, done):
        # the carry belongs to the next rollout by then
        g_l = torch.cuda.CUDAGraph()
        from . import fused
        with fused.gemm_tn_corun(self.corun), torch.cuda.graph(g_l, capture_error_mode="thread_local"):
            if PREGROW_MODE != "off" and hasattr(p.env, "pregrow"):
                # the Maze maps the NEXT rollouts' generator passes will ask for, grown on the learner's stream beside the
                # rollout that is running now: off the rollout chain, which bounds this schedule (no ordering needed between
                # the two: vec_env.VecTrack2D.pregrow). The pass itself: 206 -> 104 us at 1024 Maze + Nav envs.
                p.env.pregrow(fork=PREGROW_MODE == "fork")
            stats = p.compute_grads(_BucketOnly(self.buckets[k]), mode)
            if PREGROW_MODE == "fork" and hasattr(p.env, "generator_join"):
                p.env.generator_join()
            self.optimizer.bucket.grad.copy_(self.buckets[k].grad)
        self.graphs[(mode, k)] = (g_r, g_l, stats)
        return self.graphs[(mode, k)]

    def run(self, mode=None):
        mode = self.mode0 if mode is None else int(mode)
        k = self.i & 1
        entry = self.graphs.get((mode, k))
        if entry is None:            # a training mode seen for the first time (the evaluator's schedule): capture its pair
            self.finish()
            entry = self._capture(mode, k)
        g_r, g_l, stats = entry
        cur = torch.cuda.current_stream(self.dev)
        if self.serial:
            # the same graphs in program order on the caller's stream: R(i), L(i), O(i). The one-update delay lives in the
            # double-buffered weights (rollout i + 1 reads F_{1-k}, which O(i) does not touch), not in the timing, so this
            # is the same dataflow as the two-stream schedule below
            g_r.replay()
            g_l.replay()
            if not self.capture_allreduce:
                self.master.allreduce_grads(self.optimizer)
            self.g_opt[k].replay()
        else:
            # One call = one PHASE of the pipeline: the learner + update of the PREVIOUS rollout go out on stream L, then this
            # rollout on stream R. (Issuing rollout i and its own learner in one call would be the same dataflow, but then a
            # caller that synchronises after every call — bench.py brackets K = 20 env steps, i.e. one call — would wait for
            # the learner before the next rollout could be issued, and nothing would ever overlap.)
            if self.i == 0:
                self.sR.wait_stream(cur)
                self.sL.wait_stream(cur)
            self._issue_pending()
            with torch.cuda.stream(self.sR):
                if self.i >= 2:
                    self.sR.wait_event(self.ev_o[k])      # F_k holds theta_{i-1}; the stores of replica k are free again
                g_r.replay()
                self.ev_r[k].record(self.sR)
            self.pending = (k, g_l)
        self.master.n_steps += self.args.num_steps
        self.i += 1
        self.stats = stats
        return stats

    def _issue_pending(self):
        """Learner, all-reduce and update of the rollout issued by the previous call, on stream L."""
        if self.pending is None:
            return
        k, g_l = self.pending
        self.pending = None
        with torch.cuda.stream(self.sL):
            self.sL.wait_event(self.ev_r[k])
            g_l.replay()
            if not self.capture_allreduce:
                self.master.allreduce_grads(self.optimizer)   # (RCCL on this stream: under the next rollout)
            self.g_opt[k].replay()
            self.ev_o[k].record(self.sL)

    def sync(self):
        """Make the caller's stream wait for everything ISSUED so far (both streams). The learner of the latest rollout is
        not issued by this: between two calls of run() this is a phase boundary of the pipeline (what bench.py's timed
        regions end on: every region holds K env steps of rollouts and K / T learner updates)."""
        if not self.serial and self.i > 0:
            cur = torch.cuda.current_stream(self.dev)
            cur.wait_stream(self.sR)
            cur.wait_stream(self.sL)

    def finish(self):
        """Issue what is still owed (the learner + update of the latest rollout) and make the caller's stream wait for both
        streams: afterwards theta has received every rollout's gradient."""
        if not self.serial and self.i > 0:
            self._issue_pending()
            self.sync()

    def _schedule_tensors(self):
        """Every tensor an update of this schedule writes: master weights, optimizer state, the replicas' weight copies."""
        opt = self.optimizer
        seen, out = set(), []
        for t in [opt.bucket.flat, opt.bucket.grad] + [v for v in vars(opt).values() if isinstance(v, torch.Tensor)] \
                + [b.flat for b in self.buckets]:
            if t.data_ptr() not in seen and t.numel() > 0:
                seen.add(t.data_ptr())
                out.append(t)
        return out

    def burn_in(self, iters, mode=None):
        """GraphedIteration.burn_in for this schedule: `iters` (rounded up to whole phase pairs) iterations whose updates are
        discarded — master weights, optimizer state, both replicas' weight copies, the phase counter and the step count are
        restored; the env shard and the carried LSTM state move on. tune_streams() has this effect as well (its trial
        iterations are rolled back the same way), which is why the two-stream schedule looked like the better LEARNER in
        round 4's seed table: it was the only one whose envs had drifted apart before training began."""
        if iters <= 0:
            return
        self.finish()
        torch.cuda.synchronize(self.dev)
        tensors = self._schedule_tensors()
        saved = [t.clone() for t in tensors]
        i0, n0 = self.i, self.master.n_steps
        for _ in range(int(iters) + (int(iters) & 1)):
            self.run(mode)
        self.finish()
        torch.cuda.synchronize(self.dev)
        with torch.no_grad():
            for t, v in zip(tensors, saved):
                t.copy_(v)
        self.i, self.master.n_steps = i0, n0
        torch.cuda.synchronize(self.dev)

    def tune_streams(self, candidates=4, iters=None, partitions=("half",), keep_updates=False):
        """Pick the stream pair the two chains overlap best on. Two things are not in the application's hands and are settled
        by trial — `iters` iterations of the schedule per candidate (None: as many as fill ~50 ms, 8 to 64, from a pilot), timed on
        the host clock, the fastest kept:
          * HIP multiplexes its streams onto a few hardware queues (4 by default) in an order the application does not
            control: two streams that land on one queue run the two chains back to back (measured at 512 envs: 1.73 ms per
            iteration against 1.33 ms on distinct queues, 1.62 ms synchronous), and the mapping depends on how many streams
            the process created before — `candidates` alternative learner streams are tried;
          * a CU PARTITION (`partitions`: "half" = an even split, or a CU count for the rollout stream, the rest to the
            learner's; cu_masked_stream; skipped on parts whose CU count does not split evenly over the shader engines). On a
            shared chip a rollout kernel's workgroups queue for CU resources behind the learner's resident workgroups (a
            workgroup that needs 72 KB of LDS next to two 64 KB GEMM workgroups waits out the whole GEMM: tools/microbench/
            two_queue.hip; the rollout chain runs at 40-50 % speed next to any chip-filling learner kernel: tools/
            corun_kernels.py). With half the CUs each neither chain ever waits for the other: both run slower but
            fully concurrently — 1.15 against 1.28 ms per iteration at 512 envs; from 2048 envs up both chains are
            throughput-bound and the shared chip wins, which is what the trial then finds.
        The trial iterations are NOT training: unless keep_updates is set, the master weights, the optimizer state (step
        counter, moments), the replicas' weight copies, the phase counter and the step count are restored afterwards, so that
        iteration 0 of the caller's run is its first (the env shard and the carried LSTM state simply continue — experience
        nobody learned from). Multi-rank runs: every trial holds all-reduces, so all ranks must run the SAME candidate list
        (a rank whose CU-masked streams cannot be made would otherwise run fewer collectives and hang the job: the partition
        candidates are kept only if every rank has them, all-reduce MIN) and must KEEP the same pair (the per-candidate times
        are all-reduced MAX — an iteration is as slow as its slowest rank — before the argmin).
        Returns [(ms per iteration, chosen, label)] per candidate."""
        import time as _time
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if multi:
            # Every trial below holds collectives, so the ranks must agree on what they are about to do BEFORE any of them
            # leaves: a rank with serial=True (or another forced CU split, another candidate count — per-rank environment
            # variables) would return here while the others wait in an all-reduce for ever.
            cfg = torch.tensor([1 if self.serial else 0, int(self.cu_split), int(candidates), -1 if iters is None else int(iters),
                                len(partitions)],
                               dtype=torch.int64, device=self.dev)
            lo, hi = cfg.clone(), cfg.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            if not torch.equal(lo, hi):
                raise RuntimeError("PipelinedIteration.tune_streams: ranks disagree on (serial, cu_split, candidates, iters, "
                                   "partitions): min %s max %s — set ATR_PIPE_CU_SPLIT / the schedule identically on every rank"
                                   % (lo.tolist(), hi.tolist()))
        if self.serial:
            return []
        pairs = [(self.sR, self.sL, "as constructed")]
        if not self.cu_split:
            pairs += [(self.sR, torch.cuda.Stream(device=self.dev), "learner stream %d" % (c + 1)) for c in range(candidates)]
            total = device_cus(self.dev)
            for part in partitions:
                made = None
                try:
                    if part == "half":
                        halves = cu_partition(self.dev)
                        if halves is not None:
                            (r0, rn), (l0, ln) = halves
                            made = (cu_masked_stream(self.dev, r0, rn, total), cu_masked_stream(self.dev, l0, ln, total),
                                    "CU partition %d / %d" % (rn, ln))
                    elif 0 < int(part) < total:
                        r_cus = int(part)
                        made = (cu_masked_stream(self.dev, total - r_cus, r_cus, total),
                                cu_masked_stream(self.dev, 0, total - r_cus, total), "CU partition %d / %d" % (r_cus, total - r_cus))
                except (RuntimeError, OSError, AttributeError, ValueError):   # a runtime without CU masks
                    made = None
                if multi:       # the candidate exists on every rank or on none
                    ok = torch.tensor([1 if made is not None else 0], dtype=torch.int32, device=self.dev)
                    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                    if int(ok.item()) == 0:
                        made = None
                if made is not None:
                    pairs.append(made)
        self.finish()
        torch.cuda.synchronize(self.dev)
        tensors = None if keep_updates else self._schedule_tensors()
        saved = [t.clone() for t in tensors] if tensors is not None else None
        i0, n0 = self.i, self.master.n_steps
        # A trial must be long enough for the steady state to show: one call of run() is one PHASE, a trial starts with an empty
        # pipeline and ends with finish(), and at 512 envs eight iterations are 10 ms of which the fill / drain is a fifth — the
        # CU partition (1.13 ms per iteration in steady state against 1.24 on shared streams) then only TIES its trial and loses
        # it every other run (one collection pass of round 5: 8.3 M instead of 9.1). So the trial length is set from a pilot: ~50 ms per
        # candidate, 8 to 64 iterations (the same on every rank: the pilot time is all-reduced MAX), unless the caller fixes `iters`.
        if iters is None:
            self._use_streams(pairs[0][0], pairs[0][1])
            for _ in range(2):
                self.run()
            self.finish()
            torch.cuda.synchronize(self.dev)
            t0 = _time.perf_counter()
            for _ in range(4):
                self.run()
            self.finish()
            torch.cuda.synchronize(self.dev)
            pilot_ms = (_time.perf_counter() - t0) / 4 * 1e3
            if multi:
                pm = torch.tensor([pilot_ms], dtype=torch.float64, device=self.dev)
                dist.all_reduce(pm, op=dist.ReduceOp.MAX)
                pilot_ms = float(pm.item())
            iters = max(8, min(64, int(50.0 / max(pilot_ms, 1e-3))))
        iters += iters & 1              # whole pairs of phases per candidate: the replica parity is the same afterwards
        # two passes over the list, the better of a candidate's two times counts: one 8-iteration sample is noisy enough to
        # lose the CU partition its trial at 512 envs in two runs of six (8.2 M env steps/s instead of 8.9)
        times = [float("inf")] * len(pairs)
        for _pass in range(2):
            for j, (sR, sL, label) in enumerate(pairs):
                self._use_streams(sR, sL)
                for _ in range(2):
                    self.run()
                self.finish()
                torch.cuda.synchronize(self.dev)
                t0 = _time.perf_counter()
                for _ in range(iters):
                    self.run()
                self.finish()
                torch.cuda.synchronize(self.dev)
                times[j] = min(times[j], (_time.perf_counter() - t0) / iters * 1e3)
        self.finish()
        torch.cuda.synchronize(self.dev)
        if multi:
            tt = torch.tensor(times, dtype=torch.float64, device=self.dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            times = tt.tolist()
        best = min(range(len(pairs)), key=lambda j: times[j])
        self._use_streams(pairs[best][0], pairs[best][1])
        self.stream_choice = pairs[best][2]
        if saved is not None:
            with torch.no_grad():
                for t, v in zip(tensors, saved):
                    t.copy_(v)
            assert (self.i - i0) % 2 == 0      # (pilot 2 + 4, each candidate 2 x (2 + iters) phases: replica i0 & 1 is next, as before)
            self.i, self.master.n_steps = i0, n0
            torch.cuda.synchronize(self.dev)
        return [(times[j], j == best, pairs[j][2]) for j in range(len(pairs))]


def sync_train_modes(train_modes, device, src=0):
    """Every rank must differentiate the same loss before the all-reduce: rank `src` (the only one running the
    evaluator, which owns the schedule — test.py:84-92,129-134) broadcasts its train_modes list."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return train_modes
    t = torch.tensor([int(m) for m in train_modes], dtype=torch.int64, device=device)
    dist.broadcast(t, src=src)
    train_modes[:] = [int(v) for v in t.tolist()]
    return train_modes


def train(rank, args, shared_model, optimizer, train_modes, n_iters, env=None):
    gpu_id = args.gpu_ids[rank % len(args.gpu_ids)]
    device = torch.device('cuda:%d' % gpu_id)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    training_mode = args.train_mode
    while len(train_modes) <= rank:
        train_modes.append(training_mode)
        n_iters.append(0)
    player, optimizer = make_player(args, device, rank, world, env=env, model=shared_model, optimizer=optimizer)
    from .utils import ScalarWriter, log_train_scalars
    writer = ScalarWriter(os.path.join(args.log_dir, 'Agent:{}'.format(rank)))
    n_iter = 0
    try:
        while True:
            t0 = time.time()
            rollout(player, args.num_steps)
            training_mode = train_modes[rank]
            policy_loss, value_loss, entropies, pred_loss = player.optimize(
                None, optimizer, shared_model, training_mode, device)
            n_iter += 1
            n_iters[rank] = n_iter
            if n_iter % 10 == 0:
                torch.cuda.synchronize(device)
                fps = args.num_steps * player.num_envs / (time.time() - t0)
                log_train_scalars(writer, (policy_loss, value_loss, entropies, pred_loss), training_mode, fps, player.n_steps,
                                  player.num_agents)
                writer.flush()
            if train_modes[rank] == -100 or n_iter * world > args.max_step:   # test.py:129-134 stop rule
                break
        player.env.close()
    except KeyboardInterrupt:
        player.env.close()
    return player
