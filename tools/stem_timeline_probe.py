"""Phase timeline of k_stem_fwd16 (probe build: hipcc -DSTEM_PROBE on csrc/stem_hip.hip, loaded through T2D_LIB_PATH): wave 0 of
every workgroup stamps the 100 MHz wall clock at entry, after the prologue (weights staged, first x tile in LDS), after conv1,
after conv2, after the output left, and after its stores were acknowledged. One launch of the rollout's shape (two problems of N
frames) after a cache-washing pass; microseconds relative to the earliest entry.
    T2D_LIB_PATH=scratch_exp/libstemprobe.so python tools/stem_timeline_probe.py [N]"""
import ctypes as C
import sys

import numpy as np
import torch

from active_tracking_rl_amd import fused, vec_env
from active_tracking_rl_amd.model import CNN_maze

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
encs = [CNN_maze((1, 13, 13), 1).to(dev) for _ in range(2)]
obs = torch.randint(0, 5, (N, 2, 13, 13), device=dev).to(torch.uint8)
out = [torch.empty((N, 512), device=dev) for _ in range(2)]
scratch = torch.empty(96 << 20, device=dev)
lib = C.CDLL(vec_env.LIB_PATH)
for rep in range(3):
    scratch.add_(1.0)
    fused.stem_into2(obs[:, 0], encs[0], out[0], obs[:, 1], encs[1], out[1])
    torch.cuda.synchronize()
wgs = min(2 * ((N + 15) // 16), 512)
buf = np.zeros(2048 * 8, dtype=np.uint64)
assert lib.atr_stem_probe_read(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
t = buf.reshape(2048, 8)[:wgs, :6].astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0
names = ["entry", "prologue done", "conv1 done", "conv2 done", "output left", "stores acknowledged"]
print("%d workgroups (2 problems x %d frames); microseconds after the first workgroup's entry" % (wgs, N))
for i, n in enumerate(names):
    print("  %-20s min %6.2f  median %6.2f  max %6.2f" % (n, us[:, i].min(), np.median(us[:, i]), us[:, i].max()))
d = np.diff(us, axis=1)
for i in range(5):
    print("  %-20s -> %-20s median %6.2f us (min %.2f max %.2f)" % (names[i], names[i + 1], np.median(d[:, i]), d[:, i].min(), d[:, i].max()))
