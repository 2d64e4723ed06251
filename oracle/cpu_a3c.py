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


def _worker(rank, args, shared_model, opt_state, seconds, counter, barrier):
    os.environ["OMP_NUM_THREADS"] = "1"
    torch.set_num_threads(1)
    try:
        os.sched_setaffinity(0, {rank % (os.cpu_count() or 1)})
    except Exception:
        pass
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
        train_mode=-1, seed=1):
    """Returns dict(value=env steps/s summed over workers, cores=workers, seconds=...)."""
    from active_tracking_rl_amd.model import build_model
    from active_tracking_rl_amd.train import default_args, select_params
    from active_tracking_rl_amd.environment import _spaces
    os.environ["OMP_NUM_THREADS"] = "1"
    ncpu = os.cpu_count() or 1
    workers = max(1, min(workers, ncpu))
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
    barrier = ctx.Barrier(workers + 1)
    procs = [ctx.Process(target=_worker, args=(r, args, shared_model, opt_state, seconds, counter, barrier))
             for r in range(workers)]
    for p in procs:
        p.start()
    barrier.wait()
    t0 = time.time()
    for p in procs:
        p.join()
    dt = time.time() - t0
    total = sum(counter[:])
    return dict(value=total / seconds, cores=workers, seconds=seconds, wall=dt, steps=int(total), host_cpus=ncpu)


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
    a = ap.parse_args()
    res = run(a.env, a.workers, a.seconds, a.network, a.aux, a.train_mode)
    res["env_only"] = env_only(a.env, 2.0)
    print(json.dumps(res) if a.json else res)
