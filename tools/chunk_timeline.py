#!/usr/bin/env python3
"""Developer: the last step's kernels of a rocprofv3 --kernel-trace of tools/chunk_ab.py as a timeline (start / end in
microseconds from the step's k_scan; queue id), to see what overlaps with what.  usage: chunk_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "?"), int(r.get("Grid_Size", 0) or 0)))
rows.sort()
scans = [i for i, r in enumerate(rows) if r[2].startswith("k_scan")]
for which in (-2, -1):
    a = scans[which]
    b = scans[which + 1] if which != -1 else len(rows)
    t0 = rows[a][0]
    print("step starting at scan #%d" % (len(scans) + which))
    for s, e, name, q, g in rows[a:b]:
        print("  %8.1f .. %8.1f  (%7.1f us)  q%-3s grid %-9d %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, g, name))
    print("  step span %.1f us" % ((max(r[1] for r in rows[a:b]) - t0) / 1e3))
