"""The rollout step's library GEMMs above the pair-kernel sizes, in a replayed hipGraph of 20 launches each: fused.linear_lt
(hipBLASLt called directly, csrc/lt_gemm.cpp: per-problem kernel tuning, strided output) against the torch calls it replaces —
the one-GEMM LSTMCell (K = 384, both players batched) against an untuned torch.bmm of the same shape, the two fc + ReLU layers
written into [features | k h] rows (ldc = 384) against torch._addmm_activation into a dense tensor (TunableOp picks of
tunableop_gfx950.csv).   python tools/lt_gemm_bench.py"""
import torch
from active_tracking_rl_amd import gemm_tuning, fused
gemm_tuning.enable()
dev = torch.device("cuda:0")
def bench(fn, reps=200):
    for _ in range(3): fn()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps // 20): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps // 20 * 20)
ws = torch.empty(32 << 20, dtype=torch.uint8, device=dev)
for N in (2048, 4096):
    fh = torch.randn(2, 3, N, 384, device=dev)[:, 1]
    wc = torch.randn(2, 512, 384, device=dev); g = torch.empty(2, N, 512, device=dev)
    t_cat = bench(lambda: fused.linear_lt(fh, wc, g, workspace=ws))
    i_cat = fused.linear_lt_info(fh, wc, g, workspace=ws)
    fhc = fh.contiguous(); wct = wc.transpose(1, 2).contiguous()
    t_bmm = bench(lambda: torch.bmm(fhc, wct, out=g))
    y0 = torch.randn(N, 512, device=dev); y1 = torch.randn(N, 1024, device=dev)
    w0 = torch.randn(256, 512, device=dev); w1 = torch.randn(256, 1024, device=dev); b = torch.randn(256, device=dev)
    rows = torch.empty(N, 384, device=dev); fo = torch.empty(N, 256, device=dev)
    t0 = bench(lambda: fused.linear_lt(y0, w0, rows[:, :256], bias=b, relu=True, workspace=ws)); i0 = fused.linear_lt_info(y0, w0, rows[:, :256], bias=b, relu=True, workspace=ws)
    t1 = bench(lambda: fused.linear_lt(y1, w1, rows[:, :256], bias=b, relu=True, workspace=ws)); i1 = fused.linear_lt_info(y1, w1, rows[:, :256], bias=b, relu=True, workspace=ws)
    t0t = bench(lambda: torch._addmm_activation(b, y0, w0.t(), out=fo)); t1t = bench(lambda: torch._addmm_activation(b, y1, w1.t(), out=fo))
    print("N=%d gates lt %.2f (%s) torch.bmm-untuned %.2f | fc0 lt %.2f (%s) torch %.2f | fc1 lt %.2f (%s) torch %.2f" % (N, t_cat, i_cat, t_bmm, t0, i0, t0t, t1, i1, t1t), flush=True)
