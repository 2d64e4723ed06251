"""Where the time of the cooperative rollout step (k_coop_step, csrc/track2d_hip.hip) goes: one eager rollout with the kernel's
phase stamps on (wall_clock64, 100 MHz), per step the latest workgroup's time at every phase boundary relative to the first
workgroup's start; then the step's cost inside a replayed rollout graph against the four-launch step it replaces.
    python tools/coop_step_timeline.py [num_envs]"""
import os
import sys
import time
os.environ["ATR_COOP_STEP"] = "1"        # (the step is off by default: this tool is about it)

import torch

from active_tracking_rl_amd import fused
from active_tracking_rl_amd.train import GraphedIteration, default_args, make_player, rollout

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
args = default_args(num_envs=n)
player, opt = make_player(args, dev)
for _ in range(2):
    rollout(player, args.num_steps)
    player.optimize(None, opt, player.model, args.train_mode, dev)
torch.cuda.synchronize()
assert player.model.coop_step_seen, "the cooperative step did not run at this size"
G = fused.stream_cus(dev)
names = ["ticket", "fc tiles", "barrier 1", "gate tiles", "barrier 2", "cells + env"]
acc = torch.zeros(6, dtype=torch.float64)
worst = torch.zeros(6, dtype=torch.float64)
steps = 0
fused.COOP_PROBE = torch.zeros((G * 24,), dtype=torch.int64, device=dev)    # [G][8] phase stamps, then [G][2 layers][8] engine stamps
gacc = torch.zeros((2, 4), dtype=torch.float64)
player.begin_rollout(args.num_steps)
for t in range(args.num_steps):
    fused.COOP_PROBE.zero_()
    player.action_rollout()
    torch.cuda.synchronize()
    raw = fused.COOP_PROBE.cpu().double()
    p = raw[:G * 8].view(G, 8)
    ge = raw[G * 8:].view(G, 2, 8)
    # inside the tile engine (wave 0 of every workgroup): enter -> first K block parked -> its loop done -> all waves done -> stored
    gacc += (ge[:, :, 1:5] - ge[:, :, 0:4]).mean(0) / 100.0
    t0 = p[:, 0].min()
    ends = (p[:, 1:7].max(0).values - t0) / 100.0          # us: the LAST workgroup to pass each boundary
    starts = (p[:, 0].max() - t0) / 100.0
    if t == 0:
        xcd = (p[:, 7].long() >> 32)
        print("workgroups per XCC_ID:", torch.bincount(xcd, minlength=8).tolist(), " last workgroup starts %.2f us after the first" % starts)
    seg = torch.cat([ends[:1], ends[1:] - ends[:-1]])
    acc += seg
    worst = torch.maximum(worst, seg)
    steps += 1
player.end_rollout()
fused.COOP_PROBE = None
print("%d envs, %d workgroups: mean (max) us per phase, measured at the last workgroup to finish it" % (n, G))
for i, nm in enumerate(names):
    print("  %-12s %6.2f (%6.2f)" % (nm, acc[i] / steps, worst[i]))
print("  %-12s %6.2f" % ("kernel", acc.sum() / steps))
for i, nm in enumerate(("fc tiles", "gate tiles")):
    print("  inside %-10s (wave 0, mean over workgroups): first block's data %.2f us, rest of its loop %.2f, waiting for the other waves "
          "%.2f, sum + store %.2f" % ((nm,) + tuple((gacc[i] / steps).tolist())))
player.optimize(None, opt, player.model, args.train_mode, dev)
player.env.close()
for coop in (True, False):
    args = default_args(num_envs=n)
    player, opt = make_player(args, dev)
    player.model.coop_step = coop
    g = GraphedIteration(player, opt, args)
    for _ in range(5):
        g.run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(100):
            g.run()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 100)
    print("synchronous iteration, cooperative step %-3s: %.3f ms  (%.2f M env steps/s)  faults %d" % (
        "on" if coop else "off", best * 1e3, n * args.num_steps / best / 1e6, player.env.core.faults()))
    player.env.close()
