"""Micro-benchmark of the fused conv stem (csrc/stem_hip.hip) vs the GEMM formulation, forward and backward."""
import time
import torch
from active_tracking_rl_amd import fused
from active_tracking_rl_amd.model import CNN_maze

torch.manual_seed(0)
enc = CNN_maze((1, 13, 13), 1).cuda()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for M in (4096, 8192, 81920, 163840):
    x = torch.randint(0, 5, (M, 169), device="cuda").float()
    g = torch.randn(M, 512, device="cuda")
    params = [enc.conv1.weight, enc.conv1.bias, enc.conv2.weight, enc.conv2.bias]
    def f_fused(): return fused.stem(x, enc.conv1, enc.conv2)
    def f_gemm(): return enc.forward_dense_stem(x)
    y = f_fused()
    def b_fused():
        y = f_fused(); torch.autograd.grad(y, params, g)
    def b_gemm():
        y = f_gemm(); torch.autograd.grad(y, params, g)
    tf, tg = timeit(lambda: f_fused()), timeit(lambda: f_gemm())
    tbf, tbg = timeit(b_fused), timeit(b_gemm)
    mac = M * (512 * 144 + 784 * 9)
    print("M=%7d  fwd fused %8.1f us (%.1f TMAC/s)  gemm %8.1f us | fwd+bwd fused %8.1f us  gemm %8.1f us" % (
        M, tf, mac / tf / 1e6, tg, tbf, tbg))
