#!/usr/bin/env python3
"""Developer: GPU-side step periods (k_scan start to k_scan start) and per-kernel mean durations over a window of steps,
from a rocprofv3 --kernel-trace CSV.  usage: step_periods.py <kernel_trace.csv> <first step> <last step>"""
import collections
import csv
import sys

path, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
idx = [i for i, r in enumerate(rows) if "k_scan" in r[2]]
per = [(rows[idx[k + 1]][0] - rows[idx[k]][0]) / 1e3 for k in range(a, b)]
print("steps %d..%d: mean period %.1f us (min %.1f, max %.1f)" % (a, b, sum(per) / len(per), min(per), max(per)))
dur = collections.defaultdict(float)
busy = 0.0
for s, e, n in rows[idx[a]:idx[b]]:
    dur[n] += (e - s) / 1e3
    busy += (e - s) / 1e3
for n, v in sorted(dur.items(), key=lambda x: -x[1]):
    print("  %-28s %8.1f us per step" % (n, v / (b - a)))
print("  %-28s %8.1f us per step" % ("(sum of kernels)", busy / (b - a)))
