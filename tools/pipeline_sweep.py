"""Synchronous (GraphedIteration) vs pipelined (PipelinedIteration: rollout i + 1 under learner i, one update of delay)
iterations of the headline workload at the shard sizes.   python tools/pipeline_sweep.py [sizes...]"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")   # (see bench.py: room for the CU-partitioned stream pair)

import sys
import time

import torch

from active_tracking_rl_amd.train import GraphedIteration, PipelinedIteration, default_args, make_player

import os

sizes = [int(x) for x in sys.argv[1:]] or [512, 1024, 2048, 4096]
ORDER = os.environ.get("SWEEP_ORDER", "sp")          # s = synchronous, p = pipelined: which, and in which order
dev = torch.device("cuda:0")
for n in sizes:
    res = []
    for cls in [{"s": GraphedIteration, "p": PipelinedIteration}[c] for c in ORDER]:
        args = default_args(num_envs=n)
        player, opt = make_player(args, dev)
        g = cls(player, opt, args)
        if hasattr(g, "tune_streams") and os.environ.get("SWEEP_TUNE", "1") == "1":
            print("  stream trials (ms/iter):", " ".join("%.3f%s%s" % (ms, "*" if c else "", "(part)" if "partition" in lb else "") for ms, c, lb in g.tune_streams()), flush=True)
        for _ in range(6):
            g.run()
        if hasattr(g, "finish"):
            g.finish()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(200):
                g.run()
            if hasattr(g, "finish"):
                g.finish()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 200
            best = dt if best is None else min(best, dt)
        res.append(best)
        assert torch.isfinite(opt.bucket.flat).all()
        player.env.close()
        del g, player, opt
        torch.cuda.empty_cache()
    print("shard %5d envs | " % n + " | ".join("%s %7.3f ms/iter %6.2f M/s" % (
        {"s": "synchronous", "p": "pipelined"}[c], r * 1e3, n * 20 / r / 1e6) for c, r in zip(ORDER, res)), flush=True)
