"""The learner's six weight-gradient products (headline network) as ONE grouped launch (atr_gemm_tn_grouped) against six
single launches (atr_gemm_tn), per shard size; ATR_GEMM_TN_SLICES=s forces the K-split.   python tools/gemm_group_bench.py [envs...]"""
import sys

import torch

from active_tracking_rl_amd import fused
from active_tracking_rl_amd.shared_optim import FlatParams

dev = "cuda"
sizes = [int(x) for x in sys.argv[1:]] or [512, 1024, 2048, 4096]
shapes = [(512, 256), (512,), (512,), (512, 128), (256, 512), (256,), (256, 1024), (256,), (512, 256), (512,), (512,), (512, 128)]
params = [torch.nn.Parameter(torch.randn(*sh, device=dev)) for sh in shapes]
bucket = FlatParams(params)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for n in sizes:
    K = 20 * n
    keep = (torch.rand(K, device=dev) > 0.1).float()
    dG = [torch.randn(K, 512, device=dev) for _ in range(2)]
    feat = [torch.randn(K, 256, device=dev) for _ in range(2)]
    h = [torch.randn(K, 128, device=dev) for _ in range(2)]
    dpre = [torch.randn(K, 256, device=dev) for _ in range(2)]
    y0, y1 = torch.randn(K, 512, device=dev), torch.randn(K, 1024, device=dev)

    def grouped():
        with fused.deferred_weight_grads(bucket) as q:
            q.add(dG[0], feat[0], params[0], biases=(params[1], params[2]))
            q.add(dG[0], h[0], params[3], row_scale=keep, shift=n)
            q.add(dpre[0], y0, params[4], biases=(params[5],))
            q.add(dpre[1], y1, params[6], biases=(params[7],))
            q.add(dG[1], feat[1], params[8], biases=(params[9], params[10]))
            q.add(dG[1], h[1], params[11], row_scale=keep, shift=n)
            q.flush()

    def single():
        for p in range(2):
            fused.gemm_tn(dG[p], feat[p], colsum=True)
            fused.gemm_tn(h[p], dG[p], row_scale=keep)
        fused.gemm_tn(dpre[0], y0, colsum=True)
        fused.gemm_tn(dpre[1], y1, colsum=True)

    flop = 2.0 * K * (2 * 512 * 256 + 2 * 512 * 128 + 256 * 512 + 256 * 1024)
    tg, ts = timed(grouped), timed(single)
    print("envs %5d K %6d | grouped %7.1f us (%5.1f TFLOP/s) | six single launches %7.1f us (%5.1f TFLOP/s)" % (
        n, K, tg, flop / tg / 1e6, ts, flop / ts / 1e6), flush=True)
