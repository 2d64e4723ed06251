"""The stem launches of a rollout, as the rollout issues them (tools/stem_bench.py re-launches one problem on one buffer):
stem_into2 on the two agents' byte frames of an [N, 2, 13, 13] observation tensor (row stride 338 B) — maze-lstm pairs: N + N frames;
the tracker-aware target (tat-maze-lstm, the headline) sees BOTH frames: N + 2 N — the outputs going to the step's slice of a
[T, ., 512] buffer per player, T = 20 launches in one hipGraph; optionally a pass over a large scratch buffer
between the launches (what the other kernels of a step do to the caches).     python tools/stem_rollout_bench.py [N ...]"""
import sys

import torch

from active_tracking_rl_amd import fused
from active_tracking_rl_amd.model import CNN_maze

dev = torch.device("cuda:0")
torch.manual_seed(0)
encs = [CNN_maze((1, 13, 13), 1).to(dev) for _ in range(2)]
T = 20
for N, tat in [(int(v), t) for v in (sys.argv[1:] or [1024, 2048, 4096]) for t in (False, True)]:
    obs = torch.randint(0, 5, (T, N, 2, 13, 13), device=dev).to(torch.uint8)
    out = [torch.empty((T, N, 512), device=dev), torch.empty((T, 2 * N if tat else N, 512), device=dev)]
    scratch = torch.empty(96 << 20, device=dev)
    row = "N=%5d (%d + %d frames per launch%s)" % (N, N, 2 * N if tat else N, ": tat" if tat else "")
    for wash in (False, True):
        def step(t):
            fused.stem_into2(obs[t, :, 0], encs[0], out[0][t], obs[t] if tat else obs[t, :, 1], encs[1], out[1][t])
            if wash:
                scratch.add_(1.0)
        step(0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for t in range(T):
                step(t)
        g.replay(); g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = None
        for _ in range(5):
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (5 * T)
            best = us if best is None else min(best, us)
        row += " | %s %7.2f us per step" % ("with a 384 MB read+write pass between launches" if wash else "launches back to back", best)
    print(row, flush=True)
