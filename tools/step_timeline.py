#!/usr/bin/env python3
"""Developer: the timeline of ONE step out of a rocprofv3 --kernel-trace CSV: every dispatch between two consecutive
launches of an anchor kernel (default k_scan), with its start offset from the anchor, its duration and its stream, plus
the gaps nothing ran in.  Used to see where a slab step's extra time over the plain step goes.
usage: step_timeline.py <kernel_trace.csv> [step number (default 25)] [anchor substring]"""
import csv
import sys

path = sys.argv[1]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 25
anchor = sys.argv[3] if len(sys.argv) > 3 else "k_scan"
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""),
                 r.get("Queue_Id", "?")))
rows.sort()
idx = [i for i, r in enumerate(rows) if anchor in r[2]]
a, b = idx[k], idx[k + 1]
t0 = rows[a][0]
busy_until = t0
print("step %d: %.1f us from %s to the next %s" % (k, (rows[b][0] - t0) / 1e3, anchor, anchor))
for s, e, name, q in rows[a:b]:
    gap = (s - busy_until) / 1e3
    print("%9.1f us  +%7.1f us  queue %-3s %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name,
                                                  "   (idle %.1f us before)" % gap if gap > 1.0 else ""))
    busy_until = max(busy_until, e)
