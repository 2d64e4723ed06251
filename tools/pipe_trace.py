"""Per-queue timeline summary of a pipelined run from a rocprofv3 --kernel-trace CSV: for each hardware queue the kernels,
busy time, idle time between consecutive kernels, and the distribution of those gaps, over the last second of the trace.
  python tools/pipe_trace.py DIR"""
import csv
import glob
import sys
from collections import defaultdict

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the steady state of the pipelined blocks: between the 3rd and the 5th sixth of the k_rollout_begin launches
marks = [int(r["Start_Timestamp"]) for r in rows if "k_rollout_begin" in r["Kernel_Name"]]
lo, hi = marks[len(marks) // 4], marks[len(marks) // 2]
n_iter = len(marks) // 2 - len(marks) // 4
rows = [r for r in rows if lo <= int(r["Start_Timestamp"]) < hi]
print("# %d iterations in the window: %.1f us per iteration" % (n_iter, (hi - lo) / 1e3 / n_iter))
print("# columns:", ",".join(k for k in rows[0].keys())[:300])
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
by_q = defaultdict(list)
for r in rows:
    by_q[(r.get("Queue_Id", "?"), r.get("Stream_Id", "?"))].append(r)
for q, rs in by_q.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e3
    gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rs[:-1], rs[1:])]
    gaps_s = sorted(gaps)
    names = defaultdict(lambda: [0, 0.0])
    for r in rs:
        n = names[r["Kernel_Name"][:60]]
        n[0] += 1
        n[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("queue/stream %s: %d kernels over %.0f us: busy %.0f us, gaps %.0f us (median %.2f, p90 %.2f, max %.1f)" % (
        q, len(rs), span, busy, sum(gaps), gaps_s[len(gaps_s) // 2], gaps_s[int(len(gaps_s) * 0.9)], gaps_s[-1]))
    for k, (c, us) in sorted(names.items(), key=lambda kv: -kv[1][1])[:8]:
        print("      %5d x %7.2f us  %s" % (c, us / c, k))

ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
depth, last, hist = 0, ev[0][0], defaultdict(float)
for t, d in ev:
    hist[depth] += (t - last) / 1e3
    depth += d
    last = t
print("time with k kernels in flight (us):", {k: round(v) for k, v in sorted(hist.items())})

# ---- phases of the schedule: per stream, runs of kernels separated by gaps > 15 us; who waits for whom
print("# phases (start offset, duration, kernels, busy) per stream, first 12 of the window")
t0 = int(rows[0]["Start_Timestamp"])
for q, rs in by_q.items():
    phases, cur = [], [rs[0]]
    for a, b in zip(rs[:-1], rs[1:]):
        if int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) > 15000:
            phases.append(cur)
            cur = []
        cur.append(b)
    phases.append(cur)
    print("stream", q, "phases:", len(phases))
    for ph in phases[:12]:
        s, e = int(ph[0]["Start_Timestamp"]), int(ph[-1]["End_Timestamp"])
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ph)
        print("   %9.1f  dur %8.1f  kernels %4d  busy %8.1f   first %s | last %s" % (
            (s - t0) / 1e3, (e - s) / 1e3, len(ph), busy / 1e3, ph[0]["Kernel_Name"][:28], ph[-1]["Kernel_Name"][:28]))
