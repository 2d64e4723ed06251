"""Per-wave timeline of atr_actor_step from s_memtime stamps (library built with -DATR_EXP=1, tools/build_probes.sh):
0 wave entry, 1 first operand block staged (before the first MFMA), 2 accumulators complete, 3 outputs stored and acknowledged."""
import numpy as np
import torch
from active_tracking_rl_amd import fused

N = 4096
lstm = torch.nn.LSTMCell(256, 128).cuda()
f = torch.relu(torch.randn(N, 256, device="cuda"))
h, c = torch.randn(N, 128, device="cuda"), torch.randn(N, 128, device="cuda")
done = torch.zeros(N, dtype=torch.uint8, device="cuda")
bsum = (lstm.bias_ih + lstm.bias_hh).detach()
ho, co, acts = torch.empty_like(h), torch.empty_like(c), torch.empty(N, 512, device="cuda")
rows = []
with torch.no_grad():
    for rep in range(30):
        fused.actor_step_into(f, h, c, done, lstm, bsum, ho, co, acts)
        torch.cuda.synchronize()
        st = acts.view(torch.int32)[::32, :4].cpu().numpy().astype(np.int64) & 0xffffffff     # one stamp set per 32-row wave tile
        if rep >= 5:
            rows.append(st)
st = np.concatenate(rows)
names = ["entry", "first block staged", "accumulators done", "stored + acked"]
print("atr_actor_step, N=%d: s_memtime ticks per wave over %d wave tiles (384 MFMAs x 64 cycles = 24 576 of pure matrix-pipe time)" % (N, len(st)))
d = np.diff(st, axis=1) & 0xffffffff
print("  whole wave                                 median %7.0f  p90 %7.0f" % (np.median(d.sum(1)), np.percentile(d.sum(1), 90)))
for i in range(3):
    print("  phase %-34s median %7.0f  p90 %7.0f" % (names[i] + " -> " + names[i + 1], np.median(d[:, i]), np.percentile(d[:, i], 90)))
