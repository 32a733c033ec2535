#!/usr/bin/env python3
"""Summarise tools/pmc.sh output: per-kernel counter averages over the LAST `steps` dispatches."""
import collections, csv, glob, sys
tag = sys.argv[1]; last = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for part in "abcl":
    files = glob.glob("gpurun_out/pmc_%s_%s/*/*_counter_collection.csv" % (tag, part))
    if not files: continue
    import os
    rows = list(csv.DictReader(open(max(files, key=os.path.getmtime))))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:32]
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in per.items():
        if not any(x in k for x in ("density", "force", "reorder", "scatter", "scan")): continue
        print(part, "%-30s" % k, {c: round(sum(v[-last:]) / len(v[-last:]) / 1e6, 3) for c, v in cs.items()})
