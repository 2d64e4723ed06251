"""The step's LSTMCell product with the cell as its epilogue (atr_gate_cell, csrc/gate_cell_hip.hip) against the path it replaces
(hipBLASLt batched product -> gate tensor): correctness against plain PyTorch fp32 / float64, then both timed from a hipGraph of
`reps` launches (HIP events around the replay).
  python tools/gate_cell_bench.py [rows ...] [--timeline]"""
import sys
import torch
from active_tracking_rl_amd import fused

rows = [int(x) for x in sys.argv[1:] if x.isdigit()] or [4096]
dev = torch.device("cuda:0")
torch.manual_seed(1)
R, F = 128, 256
K = F + R
for N in rows:
    fh = torch.randn(2, N, K, device=dev) * 0.5
    fh[:, :, :F].clamp_(min=0)                       # features are ReLU outputs
    w = torch.randn(2, 4 * R, K, device=dev) * 0.05
    bias = [torch.randn(4 * R, device=dev) * 0.1 for _ in range(2)]
    c_prev = [torch.randn(N, R, device=dev) for _ in range(2)]
    done = (torch.rand(N, device=dev) < 0.1).to(torch.uint8)
    pre = torch.empty(2, N, 4 * R, device=dev)
    h = [torch.empty(N, R, device=dev) for _ in range(2)]
    c = [torch.empty(N, R, device=dev) for _ in range(2)]
    for cell in ((True, True), (True, False)):
        pre.fill_(float("nan")); [t.fill_(float("nan")) for t in h + c]
        fused.gate_cell(fh, w, bias, pre, c_prev, done, h, c, cell=cell)
        torch.cuda.synchronize()
        ref64 = torch.bmm(fh.double(), w.double().transpose(1, 2))
        e_pre = (pre.double() - ref64).abs().max().item()
        msg = "rows %5d cell %s: |pre - f64| max %.2e" % (N, cell, e_pre)
        for p in range(2):
            if not cell[p]:
                continue
            g = ref64[p] + bias[p].double()
            i_, f_, g_, o_ = g[:, :R].sigmoid(), g[:, R:2 * R].sigmoid(), g[:, 2 * R:3 * R].tanh(), g[:, 3 * R:].sigmoid()
            k = (done == 0).double().unsqueeze(1)
            cn = f_ * (k * c_prev[p].double()) + i_ * g_
            hn = o_ * cn.tanh()
            msg += "  p%d |h| %.2e |c| %.2e" % (p, (h[p].double() - hn).abs().max().item(), (c[p].double() - cn).abs().max().item())
        print(msg, flush=True)
    # timing: graph of reps launches
    reps = 20
    ws = torch.empty(32 << 20, dtype=torch.uint8, device=dev)
    def timed(fn, label):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
        flops = 2.0 * 2 * N * 4 * R * K
        print("rows %5d  %-46s %7.2f us per launch  %6.1f TFLOP/s  (%.2f of 157.3)" % (N, label, best, flops / best / 1e6, flops / best / 1e6 / 157.3),
              flush=True)
    if fused.lt_available():
        timed(lambda: fused.linear_lt(fh, w, pre, workspace=ws), "hipBLASLt batched product (gate tensor only)")
    timed(lambda: fused.gate_cell(fh, w, bias, pre, c_prev, done, h, c, cell=(False, False)), "gate_cell, product only (both players)")
    timed(lambda: fused.gate_cell(fh, w, bias, pre, c_prev, done, h, c, cell=(True, False)), "gate_cell, tracker's cell in the epilogue")
    timed(lambda: fused.gate_cell(fh, w, bias, pre, c_prev, done, h, c, cell=(True, True)), "gate_cell, both cells in the epilogue")
    for tl_cell in (((False, False), (True, False)) if "--timeline" in sys.argv else ()):
        print("timeline, cell", tl_cell)
        wgs = fused.lib().atr_gate_cell_workgroups(N)
        probe = torch.zeros(wgs, 4, dtype=torch.int64, device=dev)
        for _ in range(3):
            fused.gate_cell(fh, w, bias, pre, c_prev, done, h, c, cell=tl_cell, probe=probe)
        torch.cuda.synchronize()
        t = probe.cpu().double()
        idx = torch.arange(t.shape[0])
        live = t[:, 0] > 0
        loop = (t[:, 1] - t[:, 0]) * 0.01
        j = idx // 8
        per_player = 8 * ((((N + 127) // 128) + 7) // 8)
        for name, key in (("XCD (hw)", t[:, 3].long()), ("player", (j >= per_player).long()), ("column block", j % 8)):
            parts = []
            for v in sorted(set(key[live].tolist())):
                m = live & (key == v)
                parts.append("%d: %.1f / %.1f" % (v, loop[m].median(), loop[m].max()))
            print("  loop length median / max by %-14s %s" % (name, "  ".join(parts)))
        qs = torch.quantile(loop[live], torch.tensor([0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99, 1.0], dtype=torch.float64))
        print("  loop length quantiles 0/10/25/50/75/90/99/100 %%: " + " ".join("%.1f" % q for q in qs.tolist()))
        t = t[t[:, 0] > 0]
        t0 = t[:, 0].min()
        tick = 0.01            # wall_clock64: 100 MHz
        print("timeline (us from the first workgroup's start; %d workgroups): start  median %.2f max %.2f | main loop ends  median %.2f "
              "max %.2f | ends  median %.2f max %.2f | loop length median %.2f  epilogue median %.2f" % (
                  t.shape[0], (t[:, 0] - t0).median() * tick, (t[:, 0] - t0).max() * tick, (t[:, 1] - t0).median() * tick,
                  (t[:, 1] - t0).max() * tick, (t[:, 2] - t0).median() * tick, (t[:, 2] - t0).max() * tick,
                  (t[:, 1] - t[:, 0]).median() * tick, (t[:, 2] - t[:, 1]).median() * tick))
