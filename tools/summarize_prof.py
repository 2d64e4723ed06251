"""Condense rocprofv3 CSV output into the small text summaries committed under profiles/."""
import csv
import glob
import sys
from collections import defaultdict


def kernel_stats(path, top=60, iterations=None):
    rows = list(csv.DictReader(open(path)))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    print("# %s\n# total kernel time %.3f ms over %d kernel names" % (path, tot / 1e6, len(rows)))
    if iterations:
        print("# iterations %d  (kernel time per iteration %.3f ms)" % (iterations, tot / 1e6 / iterations))
    print("%-110s %8s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
    for r in rows[:top]:
        print("%-110s %8s %12.3f %12.2f %7.2f" % (r["Name"][:110], r["Calls"], int(r["TotalDurationNs"]) / 1e6,
                                                  float(r["AverageNs"]) / 1e3, float(r["Percentage"])))


def pmc(path, match):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if match in r["Kernel_Name"]:
            a = acc[(r["Kernel_Name"][:60], r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for (k, c), (s, n) in sorted(acc.items()):
        print("%-62s %-14s dispatches=%6d  mean=%.1f" % (k, c, n, s / n))


if __name__ == "__main__":
    mode, root = sys.argv[1], sys.argv[2]
    if mode == "stats":
        its = int(sys.argv[3]) if len(sys.argv) > 3 else None     # iterations the profiled command ran (incl. eager warm-up)
        for f in glob.glob(root + "/**/*kernel_stats.csv", recursive=True):
            kernel_stats(f, iterations=its)
    else:
        for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
            pmc(f, sys.argv[3] if len(sys.argv) > 3 else "k_env")
