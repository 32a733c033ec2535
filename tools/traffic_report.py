#!/usr/bin/env python3
"""Fold tools/traffic.sh output into profiles/r02/traffic.json (one entry per window).
FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports exactly half the bytes of a wide
coalesced streaming read (MI355X_MICROARCH.md, HBM): the raw value, the x2-corrected value and the
calibration against k_reorder (a pure streaming kernel of known byte count in this library) are
all recorded; bench.py reports the corrected figure."""
import collections, csv, glob, json, os, sys
cfg, dist, warm, steps, n = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
out_path = sys.argv[6] if len(sys.argv) > 6 else None  # default: profiles/r02/traffic.json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def per_kernel(kind, counter):
    f = max(glob.glob(os.path.join(root, "gpurun_out", "traffic_%s_%s_%s" % (cfg, dist, kind), "*", "*_counter_collection.csv")),
            key=os.path.getmtime)  # gpurun merges runs into the same directory: take the newest pass
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v[-steps:]) / len(v[-steps:]) for k, v in per.items()}  # the timed window only
fetch, write = per_kernel("fetch", "FETCH_SIZE"), per_kernel("write", "WRITE_SIZE")
# calibration: k_reorder<true> (the step loop's instantiation: predicted positions recomputed) reads slot_tmp(4) +
# id_tmp(4) + cid(4) + 2 x 16 B and writes 3 x 16 + 4 + 3 x 4 B per particle (plus the small cell-start lookups and its
# cell-mates' ids): a streaming kernel with a known byte count.  Only its two 16-B gathers fall under the half-count.
cal_read_expected, cal_write_expected = n * (4 + 4 + 4 + 32), n * (48 + 4 + 12)
cal = {"kernel": "k_reorder<true>", "reorder_fetch_raw": fetch.get("k_reorder<true>"), "reorder_read_expected": cal_read_expected,
       "reorder_write_raw": write.get("k_reorder<true>"), "reorder_write_expected": cal_write_expected}
out = {"window": {"config": cfg, "dist": dist, "warmup": warm, "steps": steps, "particles": n},
       "units": "bytes per launch, mean over the timed window",
       "fetch_raw": fetch, "write_raw": write, "calibration": cal}
force = "k_force_listed<false, false, false>"  # <IEEE, ACCEL_ONLY, CUT>: the step's launch, not the on-demand accel pass
dens = "k_density_listed<false, false>"
out["bytes_per_launch"] = {"force_integrate_bin": 2.0 * fetch[force] + write[force],
                           "density": 2.0 * fetch[dens] + write[dens]}
out["bytes_per_launch_note"] = "2 x FETCH_SIZE (gfx950 half-count correction for 16-B/lane reads) + WRITE_SIZE"
path = out_path or os.path.join(root, "profiles", "r03", "traffic.json")
allt = json.load(open(path)) if os.path.exists(path) else {}
allt["%s-%s-w%d-k%d" % (cfg, dist, warm, steps)] = out  # keyed by window: bench.py fills roofline.traffic on an exact match only
json.dump(allt, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps({k: out[k] for k in ("bytes_per_launch", "calibration")}, indent=1))
