"""k_stem_fwd16 at the headline's rollout shape (N tracker frames + 2 N target frames in one launch): per-workgroup wall-clock stamps of
a probe build (-DSTEM_PROBE; ATR_STEM_FWD16_MIN=2048 forces the 16-frame kernel): when workgroups start, how many passes each makes,
when its first pass and its last pass end.   T2D_LIB_PATH=scratch_exp/libstemprobe.so ATR_STEM_FWD16_MIN=2048 python tools/stem_tat_timeline.py [N]"""
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
out = [torch.empty((N, 512), device=dev), torch.empty((2 * N, 512), device=dev)]
scratch = torch.empty(96 << 20, device=dev)
lib = C.CDLL(vec_env.LIB_PATH)
for rep in range(3):
    scratch.add_(1.0)
    fused.stem_into2(obs[:, 0], encs[0], out[0], obs, encs[1], out[1])
    torch.cuda.synchronize()
buf = np.zeros(2048 * 8, dtype=np.uint64)
assert lib.atr_stem_probe_read(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
t = buf.reshape(2048, 8).astype(np.int64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
us = lambda c: (t[:, c] - t0) / 100.0
passes = t[:, 6]
print("%d workgroups; passes per workgroup: %s" % (len(t), dict(zip(*np.unique(passes, return_counts=True)))))
for name, v in (("entry", us(0)), ("prologue done", us(1)), ("first pass ends", us(7)), ("all stores acknowledged", us(5))):
    print("  %-26s min %6.2f  median %6.2f  max %6.2f" % (name, v.min(), np.median(v), v.max()))
b2 = np.zeros(2048 * 4, dtype=np.uint64)
assert lib.atr_stem_probe2_read(b2.ctypes.data_as(C.c_void_p), b2.size) == 0
b2 = b2.reshape(2048, 4)[: len(t)].astype(np.int64)
hw, xcc = b2[:, 0] & 0xffffffff, b2[:, 0] >> 32
cu = (xcc & 0xf) * 1024 + ((hw >> 13) & 7) * 64 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xf)        # (xcc, se, sh, cu)
where = {}
for i, u in enumerate(cu):
    where.setdefault(int(u), []).append(i)
print("  distinct CUs %d; workgroups per CU: %s; passes per CU: %s" % (
    len(where), dict(zip(*np.unique([len(v) for v in where.values()], return_counts=True))),
    dict(zip(*np.unique([int(passes[v].sum()) for v in where.values()], return_counts=True)))))
pairs = [v for v in where.values() if len(v) == 2]
print("  id distance of a CU's two workgroups: %s" % dict(zip(*np.unique([abs(v[0] - v[1]) for v in pairs], return_counts=True))))
for k in np.unique(passes):
    m = passes == k
    f = lambda a: np.median(a[m]) / 100.0
    print("  workgroups with %d pass(es): first pass: conv1 %5.2f conv2 %5.2f output %5.2f (ends %5.2f); last pass: conv1 %5.2f conv2 %5.2f "
          "output %5.2f; stores acknowledged median %6.2f max %6.2f" % (
              k, f(b2[:, 1] - t[:, 1]), f(b2[:, 2] - b2[:, 1]), f(t[:, 7] - b2[:, 2]), f(t[:, 7] - t0),
              f(t[:, 2] - (t[:, 7] if k > 1 else t[:, 1])), f(t[:, 3] - t[:, 2]), f(t[:, 4] - t[:, 3]), np.median(us(5)[m]), us(5)[m].max()))
