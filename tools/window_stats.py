#!/usr/bin/env python3
"""Per-window kernel durations from a rocprofv3 --kernel-trace CSV of `bench.py`.

bench.py runs several windows in one process (steps W..W+K, the settled window 400..500, the IEEE-division repeat),
and rocprofv3 --stats averages every launch of a kernel over the whole process.  This tool cuts the per-dispatch
trace of the SAME run into bench.py's windows (by launch number of each kernel) so that the bench line's
roofline.avg_launch_ms can be recomputed from profiles/ for exactly the launches it timed.

usage: window_stats.py <kernel_trace.csv> <warmup> <steps> [settled_from settled_steps] > summary.json"""
import collections
import csv
import json
import sys

path, warm, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
s_from, s_steps = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (400, 100)
per = collections.defaultdict(list)  # kernel name -> durations in launch order (ns)
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    per[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
out = {"source": "rocprofv3 --kernel-trace of `python3 bench.py --gpus 1 --steps %d --warmup %d`" % (steps, warm),
       "unit": "microseconds per launch (mean over the window's launches)", "windows": {}}
windows = {"timed steps %d..%d" % (warm, warm + steps): (warm, warm + steps),
           "settled steps %d..%d" % (s_from, s_from + s_steps): (s_from, s_from + s_steps)}
for name, rows in sorted(per.items()):
    rows.sort()
    d = [x[1] for x in rows]
    if len(d) < warm + steps:
        continue  # not a per-step kernel
    for label, (a, b) in windows.items():
        if b > a and len(d) >= b:
            w = d[a:b]
            out["windows"].setdefault(label, {})[name] = {"launches": len(w), "mean_us": sum(w) / len(w) / 1e3,
                                                          "min_us": min(w) / 1e3, "max_us": max(w) / 1e3}
    out.setdefault("all_launches", {})[name] = {"launches": len(d), "mean_us": sum(d) / len(d) / 1e3}
json.dump(out, sys.stdout, indent=1, sort_keys=True)
print()
