"""Where a generator pass spends its time (library built with -DT2D_EXP=9, tools/build_probes.sh): s_memtime stamps of
every wave that generated a slot — 0 start of generation, 1 map generated, 2 free-cell index built, 3 spawns / goals
picked, 4 target plan made (Nav: flood + direction field stored), 5 end; 6 = the wave's entry into the kernel (before the
Nav plan prefetch of the current episode)."""
import ctypes as C
import sys

import numpy as np
import torch

from active_tracking_rl_amd.vec_env import VecTrack2D

env_id = sys.argv[1] if len(sys.argv) > 1 else "Track2D-MazePartialNav-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
env = VecTrack2D(env_id, num_envs=n, seed=1)
out = (env.reset(), torch.empty((n, 2), device="cuda"), torch.empty((n,), dtype=torch.uint8, device="cuda"))
env.L.t2d_debug_tile_words.argtypes = [C.c_void_p, C.c_void_p]
names = ["map", "free index", "spawns/goals", "target plan", "stores"]
rows = []
for rep in range(12):
    env.step_random(10, 1, out)                      # one generator window
    torch.cuda.synchronize()
    buf = np.zeros((n, 256), np.uint32)
    assert env.L.t2d_debug_tile_words(env.h, buf.ctypes.data_as(C.c_void_p)) == 0
    st = buf[:, 246:253].astype(np.int64)
    st = st[st[:, 5] != 0]
    if len(st) == 0:
        continue
    # stamps are per pass; keep the waves of the latest pass (largest entry stamps cluster): all stamps within 2^22 of the max
    st = st[(st[:, 5].max() - st[:, 5]) % (1 << 32) < (1 << 22)]
    d = (np.diff(st[:, :6], axis=1)) % (1 << 32)
    total = (st[:, 5] - st[:, 6]) % (1 << 32)
    slow = np.argmax(total)
    rows.append((len(st), total.max(), d[slow], ((st[:, 0] - st[:, 6]) % (1 << 32))[slow], np.median(d, axis=0), np.median(total)))
print("%s N=%d: per generator pass, in s_memtime ticks" % (env_id, n))
for cnt, tmax, dslow, pre, dmed, tmed in rows[-6:]:
    print("  waves generating %4d | slowest wave %7d ticks = prefetch/entry %6d + %s | median wave %7d = %s" % (
        cnt, tmax, pre, " + ".join("%s %d" % (nm, v) for nm, v in zip(names, dslow)), tmed,
        " + ".join("%d" % v for v in dmed)))
