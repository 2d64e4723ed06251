"""Per-workgroup timeline of k_gemm_tn on the learner's six weight-gradient products: start / end (100 MHz wall clock), shader
cycles and the CU of every workgroup, from a PROBE build of the library (-DATR_TN_PROBE=1, tools/build_probes.sh ->
scratch_exp/libtnprobe.so):   T2D_LIB_PATH=scratch_exp/libtnprobe.so python tools/gemm_tn_timeline.py [envs] [workgroups]"""
import ctypes as C
import sys
import numpy as np
import torch
from active_tracking_rl_amd import fused, vec_env
from active_tracking_rl_amd.shared_optim import FlatParams
dev = "cuda"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
shapes = [(512, 256), (512,), (512,), (512, 128), (256, 512), (256,), (256, 1024), (256,), (512, 256), (512,), (512,), (512, 128)]
params = [torch.nn.Parameter(torch.randn(*sh, device=dev)) for sh in shapes]
bucket = FlatParams(params)
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
for _ in range(40):          # (steady state: the first launches after idle run at other clocks)
    grouped()
torch.cuda.synchronize()
L = vec_env.load_library()
W = 1536 if len(sys.argv) < 3 else int(sys.argv[2])
buf = np.zeros((W, 4), np.uint64)
L.atr_tn_stamps.argtypes = [C.c_void_p, C.c_int]
rc = L.atr_tn_stamps(buf.ctypes.data, W)
assert rc == 0, rc
t0 = buf[:, 0].min()
st, en = (buf[:, 0] - t0).astype(np.float64) / 100.0, (buf[:, 1] - t0).astype(np.float64) / 100.0     # us (100 MHz)
cyc = buf[:, 2].astype(np.float64)
hw = (buf[:, 3] & np.uint64(0xffffffff)).astype(np.uint64)
xcc = (buf[:, 3] >> np.uint64(32)).astype(np.uint64) & np.uint64(0xf)
cu = (hw >> np.uint64(8)) & np.uint64(0xf); sh = (hw >> np.uint64(12)) & np.uint64(1); se = (hw >> np.uint64(13)) & np.uint64(7)
dur = en - st
print("workgroups %d  span %.1f us  duration mean %.1f  min %.1f  max %.1f  p5 %.1f p95 %.1f us; clock %.0f MHz" % (
    W, en.max(), dur.mean(), dur.min(), dur.max(), np.percentile(dur, 5), np.percentile(dur, 95), (cyc / dur).mean()))
print("sum of durations / (span * 512 slots) = %.3f" % (dur.sum() / (en.max() * 512)))
order = np.argsort(st)
print("start times: first 512 by %.1f us; 513th at %.1f; 1025th at %.1f; last start %.1f; first end %.1f" % (
    np.sort(st)[511], np.sort(st)[512], np.sort(st)[1024], st.max(), en.min()))
for x in range(8):
    m = xcc == x
    print("xcc %d: wgs %d  mean dur %.1f  last end %.1f  blocks mod 8: %s" % (x, m.sum(), dur[m].mean() if m.any() else 0, en[m].max() if m.any() else 0,
          np.unique(np.nonzero(m)[0] % 8)[:8]))
# duration by round (start order thirds)
for r in range(3):
    idx = order[r * 512:(r + 1) * 512]
    print("start-order third %d: dur mean %.1f  end min %.1f max %.1f" % (r, dur[idx].mean(), en[idx].min(), en[idx].max()))
key = (xcc.astype(np.int64) << 8) | (se.astype(np.int64) << 5) | (sh.astype(np.int64) << 4) | cu.astype(np.int64)
u, cnt = np.unique(key, return_counts=True)
print("distinct (xcc, se, sh, cu): %d; workgroups per CU min %d max %d" % (len(u), cnt.min(), cnt.max()))
ends = np.array([en[key == k].max() for k in u])
print("per-CU last end: min %.1f mean %.1f max %.1f" % (ends.min(), ends.mean(), ends.max()))
tile = (np.arange(W) >> 3) % 48
names = [("dW_ih0", 0, 8), ("dW_hh0", 8, 12), ("fc0", 12, 20), ("fc1", 20, 36), ("dW_ih1", 36, 44), ("dW_hh1", 44, 48)]
for nm, a, b in names:
    m = (tile >= a) & (tile < b)
    print("%-7s wgs %4d  dur mean %.1f  min %.1f max %.1f   mean start %.1f" % (nm, m.sum(), dur[m].mean(), dur[m].min(), dur[m].max(), st[m].mean()))
sl = (np.arange(W) & 7) + 8 * ((np.arange(W) >> 3) // 48)
for s_ in range(0, 32, 4):
    m = sl == s_
    print("slice %2d: start mean %.1f  dur mean %.1f" % (s_, st[m].mean(), dur[m].mean()))
# the slots: per CU, sort its 6 WGs by start: gaps between an end and the next start
gaps = []
for k in u:
    m = np.nonzero(key == k)[0]
    o = m[np.argsort(st[m])]
    print_once = False
    gaps.append((st[o[2:]] .min() - en[o].min()))
print("first refill delay after first end on a CU: mean %.2f us" % np.mean(gaps))
