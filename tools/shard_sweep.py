"""End-to-end A3C throughput of the headline workload (BlockPartialPZR, tat-maze-lstm, train-mode -1) at the shard sizes
of the strong-scaling form on ONE MI355X: 4096 envs over 8/4/2/1 GPUs = 512/1024/2048/4096 envs per GPU, under the synchronous
(GraphedIteration) and the pipelined (PipelinedIteration + tune_streams) schedule; the env handle's fault word is read at the
end (bits 1-3: the cooperative rollout step's placement / barrier / shape faults).
  python tools/shard_sweep.py [--schedule synchronous|pipelined|both] [sizes...]        (ATR_COOP_STEP=0: the 4-launch step)"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")   # (see bench.py: room for the CU-partitioned stream pair)
import argparse
import time

import torch

from active_tracking_rl_amd.train import GraphedIteration, PipelinedIteration, default_args, make_player

ap = argparse.ArgumentParser()
ap.add_argument("sizes", type=int, nargs="*", default=[512, 1024, 2048, 4096])
ap.add_argument("--schedule", choices=("synchronous", "pipelined", "both"), default="both")
ap.add_argument("--env", default="Track2D-BlockPartialPZR-v0")
ap.add_argument("--network", default="tat-maze-lstm")
ap.add_argument("--train-mode", type=int, default=-1)
a = ap.parse_args()
dev = torch.device("cuda:0")
for n in a.sizes:
    for sched in (("synchronous", "pipelined") if a.schedule == "both" else (a.schedule,)):
        args = default_args(num_envs=n, env=a.env, network=a.network, aux="reward" if "tat" in a.network else "none",
                            train_mode=a.train_mode)
        player, opt = make_player(args, dev)
        if sched == "pipelined":
            g = PipelinedIteration(player, opt, args)
            choice = [c for c in g.tune_streams() if c[1]]
            label = choice[0][2] if choice else "one stream"
        else:
            g = GraphedIteration(player, opt, args)
            label = ""
        for _ in range(6):
            g.run()
        if sched == "pipelined":
            g.finish()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            iters = 100
            for _ in range(iters):
                g.run()
            if sched == "pipelined":
                g.finish()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / iters
            best = dt if best is None else min(best, dt)
        coop = any(getattr(p.model, "coop_step_seen", False) for p in ([player] + list(getattr(g, "players", []))))
        print("shard %5d envs  %-11s %7.3f ms/iter  %6.2f M env steps/s   coop step %s  faults %d  %s" % (
            n, sched, best * 1e3, n * args.num_steps / best / 1e6, "on" if coop else "off", player.env.core.faults(), label), flush=True)
        player.env.close()
        del g, player, opt
        torch.cuda.empty_cache()
