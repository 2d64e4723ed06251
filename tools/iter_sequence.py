"""The ordered kernel sequence of ONE replayed iteration from a rocprofv3 --kernel-trace CSV: kernels between the last two
atr::k_rollout_begin launches, with start offset, duration and the gap to the previous kernel's end.
  python tools/iter_sequence.py DIR"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_rollout_begin" in r["Kernel_Name"]]
a, b = marks[-3], marks[-2]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
print("# %d kernels, %.1f us wall" % (b - a, (int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:110]))
    prev_end = e
