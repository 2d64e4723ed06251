"""k_act_step alone (atr_act_env_step: both players' cells + heads + draws + env step) in a 9-launch hipGraph, at the shard
sizes; the launches it replaces (2 x atr_lstm_cell_forward_act1 + t2d_step_u8) the same way.
  python tools/act_step_bench.py [rows...]"""
import sys

import torch

from active_tracking_rl_amd import fused as fz
from active_tracking_rl_amd.vec_env import VecTrack2D

dev = torch.device("cuda:0")
import os
rows = [int(x) for x in sys.argv[1:]] or [512, 1024, 2048, 4096]
MODE = os.environ.get("ACT_BENCH_MODE", "all")      # "sep": only k_act_step with separate ig / hg (for the PMC passes);
#                                                      "one": only the round-4 form of the timed region above 768 rows — one gate
#                                                      tensor, bias added in the kernel, masked hidden rows written (hm_out);
#                                                      "pre": that form WITHOUT the activated-gates store (round 5: the rollout
#                                                      keeps the gate GEMM's output for the learner instead)
R = 128
for n in rows:
    core = VecTrack2D("Track2D-BlockPartialPZR-v0", num_envs=n, seed=1)
    core.reset()
    actors = [torch.nn.Linear(R, 4).to(dev) for _ in range(2)]
    smp = fz.ActionSampler(dev, seed=5)
    emb = torch.randn(4, 4 * R, device=dev)
    cprev, hout, cout = (torch.zeros(2, n, R, device=dev) for _ in range(3))
    acts = torch.empty(2, n, 4 * R, device=dev)
    actn = torch.empty(2, n, dtype=torch.int64, device=dev)
    out8 = (torch.empty((n, 2, 13, 13), dtype=torch.uint8, device=dev), torch.empty((n, 2), device=dev),
            torch.zeros((n,), dtype=torch.uint8, device=dev))
    res = []
    rows_next = torch.empty(2, n, 384, device=dev)
    for separate in ((True,) if MODE == "sep" else ((False,) if MODE in ("one", "pre") else (True, False))):
        g = torch.randn(2, n, 4 * R, device=dev)
        hg = torch.randn(2, n, 4 * R, device=dev) if separate else None
        bs = [torch.zeros(4 * R, device=dev) for _ in range(2)] if (separate or MODE in ("one", "pre")) else None
        hm = [rows_next[0][:, 256:], rows_next[1][:, 256:]] if MODE in ("one", "pre") else None

        def fused_launch():
            fz.act_env_step(core, [g[0], g[1]], [hg[0], hg[1]] if hg is not None else None, bs, [cprev[0], cprev[1]], out8[2],
                            [hout[0], hout[1]], [cout[0], cout[1]], None if MODE == "pre" else [acts[0], acts[1]], smp, actors, actn,
                            emb=emb, env_out=out8, hm_out=hm)

        def three_launches():
            for p in range(2):
                fz.lstm_cell_act_into(g[p], hg[p], cprev[p], out8[2], hout[p], cout[p], acts[p], smp, actors[p], actn[p],
                                      emb=emb if p == 1 else None, act_in=actn[0] if p == 1 else None, bias=bs[p])
            core.step_u8(actn[0], actn[1], out=out8)
        for fn in ([fused_launch] if MODE in ("sep", "one", "pre") else ([fused_launch, three_launches] if separate else [fused_launch])):
            core.flush()
            smp.begin_block()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                fn()
                core.flush()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(9):
                    fn()
            smp.end_block()
            core.flush()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tot = 0.0
            for _ in range(40):
                e0.record()
                gr.replay()
                e1.record()
                torch.cuda.synchronize()
                tot += e0.elapsed_time(e1)
            res.append(tot * 1e3 / (40 * 9))
    per_player = lambda sep: 4 * R * 4 * (2 if sep else 1) + 3 * R * 4 + 4 * R * 4
    b_sep, b_one = 709 + 2 * per_player(True) + 17, 709 + 2 * per_player(False) + 17
    if MODE == "pre":
        b_r5 = b_one + 2 * R * 4 - 2 * 4 * R * 4
        print("rows %5d | k_act_step (one gate tensor + bias + masked hidden rows, NO activated-gates store) %6.2f us = %5.0f GB/s of %d B "
              "per env-step" % (n, res[0], b_r5 * n / res[0] / 1e3, b_r5), flush=True)
    elif MODE == "one":
        b_r4 = b_one + 2 * R * 4
        print("rows %5d | k_act_step (one gate tensor + bias + masked hidden rows) %6.2f us = %5.0f GB/s of %d B per env-step" % (
            n, res[0], b_r4 * n / res[0] / 1e3, b_r4), flush=True)
    elif MODE == "sep":
        print("rows %5d | k_act_step (ig + hg) %6.2f us = %5.0f GB/s" % (n, res[0], b_sep * n / res[0] / 1e3), flush=True)
    else:
        print("rows %5d | k_act_step (ig + hg) %6.2f us = %5.0f GB/s | the 3 launches it replaces %6.2f us | k_act_step (one gate "
              "tensor) %6.2f us = %5.0f GB/s" % (n, res[0], b_sep * n / res[0] / 1e3, res[1], res[2], b_one * n / res[2] / 1e3), flush=True)
    core.close()
