"""Where a Nav generator pass (k_gen_nav: one workgroup per generated slot, three floods on three waves) spends its time.
Library built with -DT2D_EXP=9 (T2D_LIB_PATH=scratch_exp/libexp9.so): s_memtime stamps of wave 0 of every generating
workgroup, left in the slot's spare tile words: 6 entry | 0 start | 1 map generated | 2 free-cell index | 3 spawns / goals |
4 front done | 7 the two queued goals drawn | 8 own flood done (Navigator.reset's plan) | 9 all three floods done (barrier) |
5 end (validated, scalars stored). Ticks are s_memtime counts = SHADER-CLOCK cycles on this part (348 k ticks inside a 206 us kernel: ~1.7 GHz under this load), not the 100 MHz reference clock."""
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
order = [6, 0, 1, 2, 3, 4, 7, 8, 9, 5]
names = ["entry->start", "map", "free index", "spawns/goals", "front end", "2 goal draws", "own flood", "wait floods", "finish"]
print("%s N=%d: per generator pass, wave 0 of the generating workgroups, in s_memtime ticks (shader-clock cycles, ~1.7-2.1 per ns)" % (env_id, n))
for rep in range(10):
    env.step_random(20, 1, out)                      # one generator cycle
    torch.cuda.synchronize()
    buf = np.zeros((n, 256), np.uint32)              # (slot 0's tiles: the first n of n_maps)
    assert env.L.t2d_debug_tile_words(env.h, buf.ctypes.data_as(C.c_void_p)) == 0
    st = buf[:, 246:256].astype(np.int64)
    st = st[st[:, 5] != 0]
    if len(st) == 0:
        continue
    st = st[(st[:, 5].max() - st[:, 5]) % (1 << 32) < (1 << 22)]
    seq = st[:, order]
    d = np.diff(seq, axis=1) % (1 << 32)
    total = (st[:, 5] - st[:, 6]) % (1 << 32)
    span = (st[:, 5].max() - st[:, 6].min()) % (1 << 32)
    slow = np.argmax(total)
    if rep >= 4:
        print("  blocks %4d | pass span %6d | slowest %6d = %s | median %6d = %s" % (
            len(st), span, total.max(), " + ".join("%s %d" % (nm, v) for nm, v in zip(names, d[slow])), np.median(total),
            " + ".join("%d" % v for v in np.median(d, axis=0))))
