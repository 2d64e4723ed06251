"""A/B of the stem kernels on byte vs float frames (same values): forward and backward alone, HIP-event timed."""
import torch
from active_tracking_rl_amd import fused
from active_tracking_rl_amd.model import CNN_maze

torch.manual_seed(0)
enc = CNN_maze((1, 13, 13), 1).cuda()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


prm = [enc.conv1.weight.detach().contiguous(), enc.conv1.bias.detach(), enc.conv2.weight.detach().contiguous(),
       enc.conv2.bias.detach()]
for M in (81920, 163840):
    xu = torch.randint(0, 3, (M, 169), device="cuda", dtype=torch.uint8)
    xf = xu.float()
    y = torch.empty(M, 512, device="cuda")
    dy = torch.randn(M, 512, device="cuda")
    for name, x in (("u8 ", xu), ("f32", xf), ("u8 ", xu), ("f32", xf)):
        tf = timeit(lambda: fused.stem_into(x, enc.conv1, enc.conv2, y))
        tb = timeit(lambda: fused._stem_backward(x, y, dy, prm[0], prm[1], prm[2], (prm[0].shape, prm[2].shape)))
        print("M=%7d %s  fwd %7.1f us   bwd %7.1f us" % (M, name, tf, tb))
