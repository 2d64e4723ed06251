"""End-to-end A3C throughput of the headline workload (BlockPartialPZR, tat-maze-lstm, train-mode -1) at the shard sizes
of the strong-scaling form on ONE MI355X: 4096 envs over 8/4/2/1 GPUs = 512/1024/2048/4096 envs per GPU.
  python tools/shard_sweep.py [sizes...]"""
import sys
import time

import torch

from active_tracking_rl_amd.train import GraphedIteration, default_args, make_player

sizes = [int(x) for x in sys.argv[1:]] or [512, 1024, 2048, 4096]
dev = torch.device("cuda:0")
for n in sizes:
    args = default_args(num_envs=n)
    player, opt = make_player(args, dev)
    g = GraphedIteration(player, opt, args)
    for _ in range(5):
        g.run()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        iters = 60
        for _ in range(iters):
            g.run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        best = dt if best is None else min(best, dt)
    print("shard %5d envs  %7.3f ms/iter  %6.2f M env steps/s" % (n, best * 1e3, n * args.num_steps / best / 1e6), flush=True)
    player.env.close()
    del g, player, opt
    torch.cuda.empty_cache()
