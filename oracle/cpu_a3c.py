"""CPU baseline leg (TEST INFRASTRUCTURE / bench.py cpu_baseline only): a reference-shaped A3C run —
W Hogwild worker processes, ONE env each (the C oracle, numpy-legacy RNG mode), batch-1 policy on torch-CPU,
<= 20-step rollouts that break at `done`, the reference loss, a per-iteration gradient hand-off to a shared
model and SharedAdam numerics on shared-memory state (README.md:52-57; train.py:69-110; player_util.py:108-161;
utils.py:36-44; shared_optim.py:90-175). OMP_NUM_THREADS=1 per worker as main.py:3 sets.

The reference's own Python cannot travel to the GPU box, so this port is what is timed there (kind: "port").
"""
import os
import time

import numpy as np
import torch
import torch.multiprocessing as mp

from oracle import oracle as orc


def cpu_model():
    """CPU model string of the host (BASELINE.md section 3 asks for it beside every CPU number)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def physical_cores():
    """One logical CPU per physical core, from /proc/cpuinfo's (physical id, core id) pairs restricted to the CPUs this
    process may run on: workers pinned to these never share a core with an SMT sibling."""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    seen, picks = set(), []
    try:
        cur, phys, core = None, None, None
        rows = []
        for line in list(open("/proc/cpuinfo")) + ["\n"]:
            if not line.strip():
                if cur is not None:
                    rows.append((cur, phys, core))
                cur, phys, core = None, None, None
                continue
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                cur = int(v)
            elif k == "physical id":
                phys = int(v)
            elif k == "core id":
                core = int(v)
        for cpu, ph, co in rows:
            key = (ph, co) if ph is not None and co is not None else ("cpu", cpu)
            if cpu in allowed and key not in seen:
                seen.add(key)
                picks.append(cpu)
    except Exception:
        picks = []
    return picks or allowed


class OracleVecEnv(object):
    """N=1 VecEnv-protocol adapter over the C oracle (CPU tensors)."""

    def __init__(self, env_id, seed):
        from active_tracking_rl_amd import registry
        from active_tracking_rl_amd.environment import _spaces
        sp = registry.spec(env_id)
        self.num_envs = 1
        self.observation_space, self.action_space = _spaces()
        self.env = orc.OracleEnv(sp["map_type"], sp["target_mode"], sp["level"], sp["max_episode_steps"],
                                 orc.RNG_NP, seed)
        self.env.seed_np(seed)

    def reset(self):
        o = self.env.reset().astype(np.float32)
        return torch.from_numpy(o).view(1, 2, 1, 1, 13, 13)

    def step(self, actions):
        a = [int(x.reshape(-1)[0]) for x in actions]
        o, r, d, _ = self.env.step(a)
        return (torch.from_numpy(o.astype(np.float32)).view(1, 2, 1, 1, 13, 13),
                torch.from_numpy(r.astype(np.float32)).view(1, 2), torch.tensor([1 if d else 0], dtype=torch.uint8), {})

    def close(self):
        pass


def _pin(cpu):
    os.environ["OMP_NUM_THREADS"] = "1"
    torch.set_num_threads(1)
    try:
        os.sched_setaffinity(0, {cpu})
    except Exception:
        pass


def _evaluator(args, shared_model, seconds, barrier, cpu):
    """The reference's one `test` process (main.py:106, test.py:16-136) beside the workers: argmax-policy episodes of
    --env-base on the shared model, back to back, on its own core. Its env steps are not part of the reported rate."""
    _pin(cpu)
    from active_tracking_rl_amd.model import CNN_maze, build_model
    from active_tracking_rl_amd.player_util import Agent
    CNN_maze.forward = CNN_maze.forward_conv2d
    env = OracleVecEnv(args.env_base or args.env, args.seed + 1000)
    model = build_model(env.observation_space, env.action_space, args, torch.device("cpu"))
    player = Agent(model, env, args, None, torch.device("cpu"))
    barrier.wait()
    t_end = time.time() + seconds
    while time.time() < t_end:
        model.load_state_dict(shared_model.state_dict())                      # test.py:63-64
        player.reset()
        for _ in range(500):
            player.action_test()
            if bool(player.done[0]) or time.time() >= t_end:
                break


def _worker(rank, args, shared_model, opt_state, seconds, counter, barrier, cpu):
    _pin(cpu)
    from active_tracking_rl_amd.model import CNN_maze, build_model
    from active_tracking_rl_amd.player_util import Agent
    from active_tracking_rl_amd.train import select_params
    CNN_maze.forward = CNN_maze.forward_conv2d   # the reference's F.conv2d stem (perception.py:86-92) on CPU
    torch.manual_seed(args.seed + rank)
    device = torch.device("cpu")
    env = OracleVecEnv(args.env, args.seed + rank)
    model = build_model(env.observation_space, env.action_space, args, device)
    player = Agent(model, env, args, None, device)
    player.reset()
    shared_params = select_params(shared_model, args.train_mode)
    local_params = select_params(model, args.train_mode)
    exp_avg, exp_avg_sq, max_sq, step_t = opt_state
    beta1, beta2, eps, lr = 0.9, 0.999, 1e-3, args.lr
    barrier.wait()
    t_end = time.time() + seconds
    steps = 0
    while time.time() < t_end:
        model.load_state_dict(shared_model.state_dict())                    # train.py:71
        if bool(player.done[0]):
            player.reset()                                                    # train.py:73-74
        player.update_rnn_hiden()
        for _ in range(args.num_steps):
            player.action_train()
            steps += 1
            if bool(player.done[0]):
                break                                                         # train.py:85-88
        loss, *_ = player.loss(args.train_mode)
        model.zero_grad()
        loss.backward()
        player.clear_actions()
        with torch.no_grad():                                                 # SharedAdam.step on shared state
            step_t += 1
            t = float(step_t.item())
            step_size = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
            for sp, lp, m, v, vm in zip(shared_params, local_params, exp_avg, exp_avg_sq, max_sq):
                if lp.grad is None:
                    continue
                g = lp.grad
                m.mul_(beta1).add_(g, alpha=1 - beta1)
                v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
                torch.max(vm, v, out=vm)
                sp.addcdiv_(m, vm.sqrt().add_(eps), value=-step_size)
    counter[rank] = steps


def run(env_id="Track2D-BlockPartialPZR-v0", workers=16, seconds=12.0, network="tat-maze-lstm", aux="reward",
        train_mode=-1, seed=1, evaluator=True):
    """Returns dict(value=env steps/s summed over the training workers, cores=workers (+1 evaluator core), ...).
    Workers and the evaluator are pinned one per PHYSICAL core (README.md:52-57 prescribes 16 workers + 1 test process)."""
    from active_tracking_rl_amd.model import build_model
    from active_tracking_rl_amd.train import default_args, select_params
    from active_tracking_rl_amd.environment import _spaces
    os.environ["OMP_NUM_THREADS"] = "1"
    ncpu = os.cpu_count() or 1
    cores = physical_cores()
    workers = max(1, min(workers, len(cores) - (1 if evaluator and len(cores) > 1 else 0)))
    evaluator = evaluator and len(cores) > workers
    args = default_args(env=env_id, network=network, aux=aux, train_mode=train_mode, seed=seed, num_envs=1)
    torch.manual_seed(seed)
    obs_space, act_space = _spaces()
    shared_model = build_model(obs_space, act_space, args, torch.device("cpu"))
    shared_model.share_memory()
    params = select_params(shared_model, train_mode)
    mk = lambda: [torch.zeros_like(p).share_memory_() for p in params]
    opt_state = (mk(), mk(), mk(), torch.zeros(1).share_memory_())
    ctx = mp.get_context("fork")
    counter = ctx.Array("q", workers)
    barrier = ctx.Barrier(workers + 1 + (1 if evaluator else 0))
    procs = [ctx.Process(target=_worker, args=(r, args, shared_model, opt_state, seconds, counter, barrier, cores[r]))
             for r in range(workers)]
    if evaluator:
        procs.append(ctx.Process(target=_evaluator, args=(args, shared_model, seconds, barrier, cores[workers])))
    for p in procs:
        p.start()
    barrier.wait()
    t0 = time.time()
    for p in procs:
        p.join()
    dt = time.time() - t0
    total = sum(counter[:])
    return dict(value=total / seconds, cores=workers, evaluator_cores=1 if evaluator else 0, seconds=seconds, wall=dt,
                steps=int(total), host_cpus=ncpu, physical_cores=len(cores), cpu_model=cpu_model(),
                pinning="one process per physical core (/proc/cpuinfo core ids)", env=env_id, network=network, aux=aux,
                train_mode=train_mode)


def _env_proc(env_id, seconds, seed, counter, idx, barrier, cpu):
    _pin(cpu)
    env = OracleVecEnv(env_id, seed).env
    rs = np.random.RandomState(seed)
    env.reset()
    barrier.wait()
    n, t_end = 0, time.time() + seconds
    while time.time() < t_end:
        for _ in range(256):
            _, _, d, _ = env.step(rs.randint(0, 4, 2))
            n += 1
            if d:
                env.reset()
    counter[idx] = n


def env_only_mp(env_id="Track2D-BlockPartialPZR-v0", procs=16, seconds=3.0, seed=1):
    """The restated env alone (random actions, resets included) on `procs` processes pinned to physical cores:
    aggregate env steps/s (BASELINE.md section 3 asks for 1 and 16 processes)."""
    cores = physical_cores()
    procs = max(1, min(procs, len(cores)))
    ctx = mp.get_context("fork")
    counter = ctx.Array("q", procs)
    barrier = ctx.Barrier(procs)
    ps = [ctx.Process(target=_env_proc, args=(env_id, seconds, seed + i, counter, i, barrier, cores[i])) for i in range(procs)]
    for p in ps:
        p.start()
    for p in ps:
        p.join()
    return dict(value=sum(counter[:]) / seconds, procs=procs, seconds=seconds)


def env_only(env_id="Track2D-BlockPartialPZR-v0", seconds=3.0, seed=1):
    """Oracle env alone, one core, random actions (steps/s incl. resets)."""
    env = OracleVecEnv(env_id, seed).env
    rs = np.random.RandomState(seed)
    env.reset()
    n, t_end = 0, time.time() + seconds
    while time.time() < t_end:
        for _ in range(256):
            _, _, d, _ = env.step(rs.randint(0, 4, 2))
            n += 1
            if d:
                env.reset()
    return n / seconds


if __name__ == "__main__":
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="Track2D-BlockPartialPZR-v0")
    ap.add_argument("--network", default="tat-maze-lstm")
    ap.add_argument("--aux", default="reward")
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--seconds", type=float, default=12.0)
    ap.add_argument("--train-mode", type=int, default=-1)
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--suite", action="store_true",
                    help="BASELINE.md section 3: the headline config AND config 1 (BlockPartialRam, maze-lstm, --aux none, "
                         "train-mode 0), each 16 workers + 1 evaluator, plus the env alone on 1 and 16 processes")
    a = ap.parse_args()
    if a.suite:
        out = []
        for tag, env_id, net, aux, tm in (
                ("headline: BASELINE config 3 shape on CPU", "Track2D-BlockPartialPZR-v0", "tat-maze-lstm", "reward", -1),
                ("BASELINE config 1 (README.md:71)", "Track2D-BlockPartialRam-v0", "maze-lstm", "none", 0)):
            r = run(env_id, a.workers, a.seconds, net, aux, tm)
            r["config"] = tag
            r["env_only_1proc"] = env_only_mp(env_id, 1, 2.0)["value"]
            r["env_only_16proc"] = env_only_mp(env_id, 16, 2.0)["value"]
            out.append(r)
        print(json.dumps(out))
    else:
        res = run(a.env, a.workers, a.seconds, a.network, a.aux, a.train_mode)
        res["env_only"] = env_only(a.env, 2.0)
        print(json.dumps(res) if a.json else res)
