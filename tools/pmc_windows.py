#!/usr/bin/env python3
"""Fold tools/pmc.sh output (SQ / TA counter passes of ONE window) into profiles/rNN/pmc_windows.json, keyed like
traffic.json ('<config>-<dist>-w<warmup>-k<steps>'): per kernel, the mean over the window's launches of every counter
collected.  bench.py fills roofline.secondary from it on an exact window match only.
usage: pmc_windows.py <tag> <config> <dist> <warmup> <steps> <out.json>"""
import collections, csv, glob, json, os, sys
tag, cfg, dist, warm, steps, out_path = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
per = collections.defaultdict(dict)
for part in "abcl":
    files = glob.glob("gpurun_out/pmc_%s_%s/*/*_counter_collection.csv" % (tag, part))
    if not files:
        continue
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        vals[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in vals.items():
        if not k.startswith("k_"):
            continue
        for c, v in cs.items():
            if len(v) >= steps:
                per[k][c] = sum(v[-steps:]) / steps  # the timed window only
for k, c in per.items():
    if c.get("TA_BUSY_avr") is not None and c.get("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (kernel duration x 2.4 GHz x 8.07 for every kernel checked),
        # TA_BUSY_avr as the average over the TA instances: busy fraction = TA_BUSY_avr / (GRBM_GUI_ACTIVE / 8)
        c["ta_busy_frac"] = c["TA_BUSY_avr"] / (c["GRBM_GUI_ACTIVE"] / 8.0)
allw = json.load(open(out_path)) if os.path.exists(out_path) else {}
allw["%s-%s-w%d-k%d" % (cfg, dist, warm, steps)] = per
allw["_units"] = ("mean per launch over the window's launches; SQ_INSTS_* = wave-instructions; SQ_WAVE_CYCLES / SQ_WAIT_* / "
                  "SQ_ACTIVE_INST_* in quad-cycles (MI355X_MICROARCH.md); lane utilisation = SQ_THREAD_CYCLES_VALU / 64 / "
                  "SQ_ACTIVE_INST_VALU; ta_busy_frac = TA_BUSY_avr / (GRBM_GUI_ACTIVE / 8): GRBM_GUI_ACTIVE is summed over the 8 XCDs")
json.dump(allw, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps({k: {c: round(v, 1) for c, v in cs.items() if c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "ta_busy_frac")} for k, cs in per.items()}))
