"""How much does the rollout graph (the chain of small dependent launches) slow down next to a synthetic co-runner on another
stream?  python tools/corun_probe.py N"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")   # (see bench.py: room for the CU-partitioned stream pair)

import sys
import time

import torch

from active_tracking_rl_amd import fused
from active_tracking_rl_amd.train import PipelinedIteration, default_args, make_player

n = int(sys.argv[1])
dev = torch.device("cuda:0")
args = default_args(num_envs=n)
player, opt = make_player(args, dev)
g = PipelinedIteration(player, opt, args)
g.tune_streams()
for _ in range(6):
    g.run()
g.finish()
torch.cuda.synchronize()
g_r = g.graphs[(g.mode0, 0)][0]
g_l = g.graphs[(g.mode0, 0)][1]
a = torch.randn(4096, 4096, device=dev)
b = torch.randn(4096, 4096, device=dev)
c = torch.empty(4096, 4096, device=dev)
x = torch.randn(64 << 20, device=dev)
y = torch.empty_like(x)
fused.use_uncached = True
yu = fused.uc_empty("probe", (64 << 20,), dev)
small = torch.randn(1 << 16, device=dev)
small2 = torch.empty_like(small)


def load_mm():
    torch.mm(a, b, out=c)


def load_copy():
    y.copy_(x)


def load_copy_uc():
    yu.copy_(x)


def load_sum():
    x.sum()


def load_small_writes():
    for _ in range(20):
        small2.copy_(small)


E = lambda: torch.cuda.Event(enable_timing=True)
for name, load in (("nothing", None), ("4096^3 matmuls", load_mm), ("256 MB copies", load_copy),
                   ("256 MB copies into uncached memory", load_copy_uc), ("256 MB reductions", load_sum),
                   ("chains of 256 KB copies", load_small_writes), ("the learner graph", lambda: g_l.replay())):
    torch.cuda.synchronize()
    e0, e1 = E(), E()
    reps = 20
    # keep the co-runner busy for the whole measurement: enqueue plenty, then the rollouts, then drain
    if load is not None:
        with torch.cuda.stream(g.sL):
            for _ in range(3):
                load()
    with torch.cuda.stream(g.sR):
        e0.record(g.sR)
    for _ in range(reps):
        if load is not None:
            with torch.cuda.stream(g.sL):
                load()
                load()
        with torch.cuda.stream(g.sR):
            g_r.replay()
    with torch.cuda.stream(g.sR):
        e1.record(g.sR)
    torch.cuda.synchronize()
    print("n=%d rollout graph next to %-40s %8.1f us per rollout" % (n, name + ":", e0.elapsed_time(e1) * 1e3 / reps), flush=True)
player.env.close()
