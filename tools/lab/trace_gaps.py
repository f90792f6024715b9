"""Reads a rocprofv3 kernel_trace.csv and prints, for the blocked loop, the average duration of each
kernel and the average gap (previous kernel's end -> this kernel's start) in front of it, split by
what the previous kernel was. Usage: python tools/lab/trace_gaps.py <..._kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()


def short(n):
    for k in ("k_blk_pick_generic", "k_blk_pick", "k_blk_prep", "k_blk_sweep_full", "k_blk_sweep", "k_blk_budget",
              "k_build", "k_reset_loop", "k_init_basis", "k_blk_stage"):
        if k in n:
            return k
    return n[:40]


dur = defaultdict(list)
gap = defaultdict(list)
prev = None
for s, e, n in rows:
    k = short(n)
    dur[k].append(e - s)
    if prev is not None:
        gap[(short(prev[2]), k)].append(s - prev[1])
    prev = (s, e, n)
print("%-24s %8s %10s" % ("kernel", "calls", "avg ns"))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("%-24s %8d %10.0f" % (k, len(v), sum(v) / len(v)))
print("\n%-24s -> %-24s %8s %10s %10s" % ("previous", "next", "n", "avg gap ns", "median"))
for (a, b), v in sorted(gap.items(), key=lambda kv: -len(kv[1]))[:14]:
    v = sorted(v)
    print("%-24s -> %-24s %8d %10.0f %10.0f" % (a, b, len(v), sum(v) / len(v), v[len(v) // 2]))
total = rows[-1][1] - rows[0][0]
print("\nspan %.1f us, sum of durations %.1f us" % (total / 1e3, sum(e - s for s, e, _ in rows) / 1e3))
