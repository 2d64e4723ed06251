"""PipelinedIteration: host issue time vs total time per iteration, in blocks (does the overlap hold over time?).
  python tools/pipe_probe.py N [serial]"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")   # (see bench.py: room for the CU-partitioned stream pair)

import sys
import time

import torch

from active_tracking_rl_amd.train import PipelinedIteration, default_args, make_player

n = int(sys.argv[1])
serial = len(sys.argv) > 2
dev = torch.device("cuda:0")
args = default_args(num_envs=n)
player, opt = make_player(args, dev)
g = PipelinedIteration(player, opt, args, serial=serial)
for _ in range(6):
    g.run()
g.finish()
torch.cuda.synchronize()
for blk in range(6):
    t0 = time.perf_counter()
    for _ in range(100):
        g.run()
    t1 = time.perf_counter()
    g.finish()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("n=%d serial=%s block %d: host issue %.3f ms/iter, total %.3f ms/iter" % (n, serial, blk, (t1 - t0) * 10, (t2 - t0) * 10), flush=True)
# the two chains alone (state is garbage afterwards: a probe)
(g_r, g_l, _), g_o = g.graphs[(g.mode0, 0)], g.g_opt[0]
for name, fn in (("rollout chain", lambda: g_r.replay()), ("learner + optimizer chain", lambda: (g_l.replay(), g_o.replay()))):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    print("n=%d %s alone: %.3f ms" % (n, name, (time.perf_counter() - t0) * 10), flush=True)
player.env.close()
