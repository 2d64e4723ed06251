"""End-to-end A3C throughput of the other BASELINE.json configurations at their per-GPU size on ONE MI355X (the
headline configuration is bench.py's). Same drivers as bench.py: the synchronous schedule (GraphedIteration) and the pipelined
one (PipelinedIteration), 20-step rollouts, SharedAdam."""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")   # (see bench.py: room for the CU-partitioned stream pair)

import time

import numpy as np
import torch

from active_tracking_rl_amd import registry
from active_tracking_rl_amd.environment import VecEnv
from active_tracking_rl_amd.train import GraphedIteration, PipelinedIteration, default_args, make_player

CASES = [
    ("configs[1] BlockPartialRam, 1024 envs, maze-lstm, train-mode 0",
     dict(env="Track2D-BlockPartialRam-v0", num_envs=1024, network="maze-lstm", aux="none", train_mode=0), None),
    ("configs[2] BlockPartialPZR, 4096 envs, tat-maze-lstm, train-mode -1 (= bench.py)",
     dict(env="Track2D-BlockPartialPZR-v0", num_envs=4096, network="tat-maze-lstm", aux="reward", train_mode=-1), None),
    ("configs[3] MazePartialNav, 1024 envs per GPU (8192 over 8), maze-lstm, train-mode 0",
     dict(env="Track2D-MazePartialNav-v0", num_envs=1024, network="maze-lstm", aux="none", train_mode=0), None),
    ("configs[4] BlockPartialAdv, 2048 envs per GPU (16384 over 8), maze-lstm, train-mode -1, Block/Maze 50/50",
     dict(env="Track2D-BlockPartialAdv-v0", num_envs=2048, network="maze-lstm", aux="none", train_mode=-1), "mixed"),
]
dev = torch.device("cuda:0")
print("# episodes of these runs come from the device's Philox generators: the scripted Ram target's plans agree with the reference's in "
      "distribution and the Nav target's paths in LENGTH (BFS field), not draw for draw — the draw-for-draw device mode (t2d_np_attach: "
      "RamAgent, Navigator + heap A*) is the parity tool, tests/test_drivers_gpu.py", flush=True)
import sys
only = [int(x) for x in sys.argv[1:]]              # e.g. `config_sweep.py 3 4`: just those configurations
for ci, (name, over, special) in enumerate(CASES):
    if only and (ci + 1) not in only:
        continue
    res = []
    for cls in (GraphedIteration, PipelinedIteration):
        args = default_args(**over)
        env = None
        if special == "mixed":
            n = args.num_envs
            maps = np.array([registry.MAP_CODE["Block"] if i % 2 == 0 else registry.MAP_CODE["Maze"] for i in range(n)], np.uint8)
            env = VecEnv(args.env, n, device="cuda:0", seed=args.seed, map_type_per_env=maps)
        player, opt = make_player(args, dev, 0, 1, env=env)
        g = cls(player, opt, args)
        drain = getattr(g, "finish", lambda: None)
        if hasattr(g, "tune_streams"):
            g.tune_streams()
        for _ in range(5):
            g.run()
        drain()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):                          # (best of three 100-iteration regions)
            t0 = time.time()
            iters = 100
            for _ in range(iters):
                g.run()
            drain()
            torch.cuda.synchronize()
            dt = (time.time() - t0) / iters
            best = dt if best is None else min(best, dt)
        res.append(best)
        player.env.close()
        del g, player, opt
        torch.cuda.empty_cache()
    print("%-100s synchronous %7.3f ms/iter %6.2f M env steps/s | pipelined %7.3f ms/iter %6.2f M env steps/s" % (
        name, res[0] * 1e3, args.num_envs * args.num_steps / res[0] / 1e6, res[1] * 1e3,
        args.num_envs * args.num_steps / res[1] / 1e6), flush=True)
