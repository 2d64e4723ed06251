"""Where the time of a pipelined iteration goes, without a profiler: HIP timing events around the rollout graph (stream R) and
around learner + update (stream L) of consecutive iterations; prints per iteration the start / end of both relative to a base.
  python tools/pipe_timeline.py N"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")   # (see bench.py: room for the CU-partitioned stream pair)

import sys

import torch

from active_tracking_rl_amd.train import PipelinedIteration, default_args, make_player

n = int(sys.argv[1])
dev = torch.device("cuda:0")
args = default_args(num_envs=n)
player, opt = make_player(args, dev)
g = PipelinedIteration(player, opt, args)
g.tune_streams()
for _ in range(20):
    g.run()
g.finish()
torch.cuda.synchronize()
IT = 12
E = lambda: torch.cuda.Event(enable_timing=True)
ev = [[E() for _ in range(4)] for _ in range(IT)]
base = E()
base.record(g.sR)
g.sL.wait_stream(g.sR)
for i in range(IT):
    k = g.i & 1
    g_r, g_l, _ = g.graphs[(g.mode0, k)]
    with torch.cuda.stream(g.sR):
        if g.i >= 2:
            g.sR.wait_event(g.ev_o[k])
        ev[i][0].record(g.sR)
        g_r.replay()
        ev[i][1].record(g.sR)
        g.ev_r[k].record(g.sR)
    with torch.cuda.stream(g.sL):
        g.sL.wait_event(g.ev_r[k])
        ev[i][2].record(g.sL)
        g_l.replay()
        g.g_opt[k].replay()
        ev[i][3].record(g.sL)
        g.ev_o[k].record(g.sL)
    g.i += 1
g.finish()
torch.cuda.synchronize()
print("n=%d  iteration: rollout [start, end] (dur) | learner+update [start, end] (dur)   (us since base)" % n)
for i in range(IT):
    t = [base.elapsed_time(e) * 1e3 for e in ev[i]]
    print("  %2d: R [%8.1f, %8.1f] (%6.1f) | L [%8.1f, %8.1f] (%6.1f)" % (i, t[0], t[1], t[1] - t[0], t[2], t[3], t[3] - t[2]))
player.env.close()
