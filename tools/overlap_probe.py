"""Do two hipGraph replays on two streams overlap on this stack? Two independent training instances (own envs, models,
optimizers) of N envs each: iterations issued back to back on one stream vs alternately on two streams.
  python tools/overlap_probe.py [N]"""
import sys
import time

import torch

from active_tracking_rl_amd.train import GraphedIteration, default_args, make_player

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
inst = []
for k in range(2):
    args = default_args(num_envs=n, seed=1 + k)
    player, opt = make_player(args, dev)
    inst.append(GraphedIteration(player, opt, args))
for g in inst:
    for _ in range(3):
        g.run()
torch.cuda.synchronize()


def timed(fn, iters=60):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


one = timed(lambda: inst[0].run())
seq = timed(lambda: (inst[0].run(), inst[1].run()))
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    with torch.cuda.stream(s0):
        inst[0].run()
    with torch.cuda.stream(s1):
        inst[1].run()


par = timed(both)
print("N=%d: one instance %.3f ms/iter | two instances, one stream %.3f ms | two instances, two streams %.3f ms "
      "(perfect overlap = %.3f)" % (n, one, seq, par, one))
