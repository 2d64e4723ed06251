"""Measured ceilings to put beside the step kernel's numbers: pure-write (fill), pure-read (sum) and copy rates of the
chip at the byte counts one batched step moves (the obs write dominates: 1352 B x N)."""
import torch

dev = torch.device("cuda:0")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def t_us(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for n in (4096, 65536, 262144, 1048576):
    nbytes = 1352 * n
    x = torch.empty(nbytes // 4, device=dev)
    y = torch.empty_like(x)
    f = t_us(lambda: x.fill_(1.0))
    c = t_us(lambda: y.copy_(x))
    r = t_us(lambda: x.sum())
    print("N=%8d  %8.1f MB  fill %8.1f us = %6.0f GB/s   copy %8.1f us = %6.0f GB/s (r+w)   read(sum) %8.1f us = %6.0f GB/s"
          % (n, nbytes / 1e6, f, nbytes / f / 1e3, c, 2 * nbytes / c / 1e3, r, nbytes / r / 1e3))
