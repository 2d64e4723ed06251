"""Timeline of one k_step2 launch from s_memtime stamps (library built with -DT2D_EXP=6, see tools/exp_variants.sh).
Stamps per env (leader lane): 0 entry, 1 state arrived, 2 rows + reward table arrived / wall vote done, 3 state stored,
4 LDS stage written + synced, 5 obs stores issued, 6 all stores acknowledged."""
import ctypes as C
import sys

import numpy as np
import torch

from active_tracking_rl_amd.vec_env import VecTrack2D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = VecTrack2D("Track2D-BlockPartialPZR-v0", num_envs=n, seed=1)
out = (env.reset(), torch.empty((n, 2), device="cuda"), torch.empty((n,), dtype=torch.uint8, device="cuda"))
env.step_random(37, 1, out)
torch.cuda.synchronize()
env.L.t2d_debug_tile_words.argtypes = [C.c_void_p, C.c_void_p]
acc = []
for rep in range(20):
    env.step_random(1, 1, out)
    buf = np.zeros((n, 256), np.uint32)
    assert env.L.t2d_debug_tile_words(env.h, buf.ctypes.data_as(C.c_void_p)) == 0
    st = buf[:, 246:253].astype(np.int64)
    st = st[st[:, 0] != 0]
    t0 = st[:, 0].min()
    rel = (st - t0) & 0xffffffff
    acc.append(rel)
rel = np.concatenate(acc)
names = ["entry", "state", "rows+lut", "state_st", "lds", "obs_issued", "acked"]
print("N=%d  waves x reps = %d; cycles since the first wave's entry (median / p90 / max):" % (n, len(rel)))
for i, nm in enumerate(names):
    print("  %-10s %8.0f %8.0f %8.0f" % (nm, np.median(rel[:, i]), np.percentile(rel[:, i], 90), rel[:, i].max()))
d = np.diff(rel, axis=1)
print("per-wave phase durations (median / p90):")
for i in range(6):
    print("  %-22s %8.0f %8.0f" % (names[i] + "->" + names[i + 1], np.median(d[:, i]), np.percentile(d[:, i], 90)))
print("span (max acked): %d cycles" % rel[:, 6].max())
