#!/usr/bin/env python3
"""Summary of a `rocprofv3 --kernel-trace --memory-copy-trace` run of tools/readback_timeline.py: per kind of kernel the
mean duration while a device->host copy is in flight and while none is, the copies themselves (kernel or SDMA record),
and the frame period.  usage: readback_trace_summary.py <trace dir> > summary.json"""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
mt = glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True)
rows = list(csv.DictReader(open(kt)))
ker = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Queue_Id"]) for r in rows]
copies = [(s, e, n) for s, e, n, q in ker if "copy" in n.lower() and (e - s) > 200e3]  # the 50 MB copies, as kernels
sdma = []
if mt:
    for r in csv.DictReader(open(mt[0])):
        if "DEVICE_TO_HOST" in r["Direction"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 200e3:
            sdma.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "sdma"))
allc = sorted(copies + sdma)


def overlaps(s, e):
    return any(cs < e and s < ce for cs, ce, _ in allc)


stat = {}
for s, e, n, q in ker:
    if "copy" in n.lower() and (e - s) > 200e3:
        continue
    if not n.startswith(("k_scan", "k_place", "k_reorder", "k_density", "k_force", "k_gather")):
        continue
    key = n.split("<")[0]
    st = stat.setdefault(key, {"during_copy": [], "alone": []})
    st["during_copy" if overlaps(s, e) else "alone"].append((e - s) / 1e3)
out = {"trace": kt, "copy_as": "shader kernel (%s)" % copies[0][2] if copies else ("sdma record" if sdma else "none seen"),
       "copies": len(allc), "copy_us_mean": sum(e - s for s, e, _ in allc) / max(len(allc), 1) / 1e3,
       "device_to_host_records_in_memory_copy_trace": len(sdma), "kernels_us": {}}
for k, v in sorted(stat.items()):
    out["kernels_us"][k] = {w: {"n": len(x), "mean": sum(x) / len(x), "max": max(x)} for w, x in v.items() if x}
gs = sorted(s for s, e, n, q in ker if n.startswith("k_gather"))
per = [(b - a) / 1e3 for a, b in zip(gs, gs[1:])]
out["gather_to_gather_period_us"] = per
json.dump(out, sys.stdout, indent=1)
