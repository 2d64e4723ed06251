"""atr_lstm_bptt (one launch) against the per-step path (cell backward + batched GEMM per step) inside hipGraphs.
  python tools/bptt_bench.py [rows...]"""
import sys

import torch

from active_tracking_rl_amd import fused, gemm_tuning

gemm_tuning.enable()
dev = torch.device("cuda:0")
R, T, P = 128, 20, 2
for N in ([int(x) for x in sys.argv[1:]] or [512, 1024, 2048, 4096]):
    whh = torch.randn(P, R, 4 * R, device=dev) * 0.08
    keep = (torch.rand(T, N, device=dev) > 0.1).float()
    h_all = torch.randn(P, T + 1, N, R, device=dev)
    c_all = torch.randn(P, T + 1, N, R, device=dev)
    acts = torch.rand(P, T, N, 4 * R, device=dev)
    dhs = [torch.randn(T, N, R, device=dev) for _ in range(P)]
    out = []
    for flag in (True, False):
        fused.use_fused_bptt = flag
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(2):
                fused._lstm_bptt(whh, keep, h_all, c_all, acts, list(dhs))
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(3):
                fused._lstm_bptt(whh, keep, h_all, c_all, acts, list(dhs))
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / 30)
    fused.use_fused_bptt = True
    print("rows %5d  T=%d P=%d | BPTT + dW_hh: one launch %7.1f us, per-step path %7.1f us" % (N, T, P, out[0], out[1]), flush=True)
