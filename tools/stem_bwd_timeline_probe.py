"""Phase timeline of k_stem_bwd16's LAST pass per workgroup (probe build: hipcc -DSTEM_PROBE, loaded through T2D_LIB_PATH): wave 0
stamps the 100 MHz wall clock at the pass's start, after conv1's barrier, after the next pass's dz2 tile is written and the loads
of the pass after it are issued, after the MFMA phase's barrier; every wave stamps the end of its own MFMA phase (the roles'
balance) and, in SHADER-clock counts (s_memtime), its progress inside the phase: products (1) issued, the next pass's dz2 tile written,
last MFMA issued — whose ratio to the wall clock is the clock the chip actually runs this kernel at.   T2D_LIB_PATH=... python tools/stem_bwd_timeline_probe.py [M]"""
import ctypes as C
import sys

import numpy as np
import torch

from active_tracking_rl_amd import fused, vec_env
from active_tracking_rl_amd.model import CNN_maze

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 81920
enc = CNN_maze((1, 13, 13), 1).to(dev)
x = torch.randint(0, 5, (M, 169), device=dev).to(torch.uint8)
y = fused.stem(x, enc.conv1, enc.conv2)
dy = torch.randn_like(y)
prm = [enc.conv1.weight.detach().contiguous(), enc.conv1.bias.detach(), enc.conv2.weight.detach().contiguous()]
for _ in range(3):
    fused._stem_backward(x, y, dy, prm[0], prm[1], prm[2], (prm[0].shape, prm[2].shape))
torch.cuda.synchronize()
lib = C.CDLL(vec_env.LIB_PATH)
buf = np.zeros(2048 * 8, dtype=np.uint64)
assert lib.atr_stem_probe_read(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
t = buf.reshape(2048, 8).astype(np.int64)
wg = 256
a, w = t[:wg, :4], t[1024:1024 + wg]
print("k_stem_bwd16, %d frames on 256 workgroups: the last pass of every workgroup, microseconds" % M)
names = ["pass start", "conv1 done (barrier)", "next dz2 written, loads issued", "MFMA phase done (barrier)"]
d = np.diff(a, axis=1) / 100.0
for i in range(3):
    print("  %-26s -> %-26s median %6.2f us (min %.2f max %.2f)" % (names[i], names[i + 1], np.median(d[:, i]), d[:, i].min(), d[:, i].max()))
rel = (w - a[:, 1:2]) / 100.0
print("  per role, MFMA phase (conv1's barrier -> the wave's last MFMA issued), median us: " + " ".join("%.2f" % np.median(rel[:, k]) for k in range(8)))
b2 = np.zeros(2048 * 4, dtype=np.uint64)
if hasattr(lib, "atr_stem_probe2_read") and lib.atr_stem_probe2_read(b2.ctypes.data_as(C.c_void_p), b2.size) == 0:
    # per wave, inside the MFMA phase, in SHADER-clock counts (s_memtime): start, products (1) issued, next dz2 written, last MFMA issued
    sc = b2.reshape(256, 8, 4).astype(np.int64)
    ok = sc[:, 0, 0] > 0
    sc = sc[ok]
    cyc = sc - sc[:, :, :1]
    # the shader clock during the phase: counts of wave 0 over the same interval on the 100 MHz wall clock (phase start stamp -> its last MFMA)
    wall_us = (w[ok, 0] - a[ok, 2]) / 100.0
    ghz = np.median(cyc[:, 0, 3] / (wall_us * 1e3))
    print("  shader clock during the MFMA phase (s_memtime counts / wall time of wave 0): %.2f GHz (nominal 2.4: the f32 MFMA peak at this clock is %.1f TFLOP/s)" % (ghz, 157.3 * ghz / 2.4))
    print("  per wave, shader cycles since its own start of the MFMA phase (medians over %d workgroups; 32 cycles per MFMA at full rate):" % len(sc))
    for k, name in ((1, "(1) issued"), (2, "next dz2 written"), (3, "last MFMA issued")):
        print("    %-18s %s" % (name, " ".join("%6.0f" % np.median(cyc[:, r, k]) for r in range(8))))
