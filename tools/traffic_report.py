#!/usr/bin/env python3
"""Fold tools/traffic.sh output into profiles/rNN/traffic.json (one entry per window).
Reads: the size-resolved fabric read-request counters of gfx950, bytes = 32 n32 + 64 n64 + 128 n128.  The calibration
on kernels of known byte counts (profiles/r04/traffic_calibration.json) shows that EVERY read request is a 128-byte line,
streamed or gathered, and that FETCH_SIZE tallies each at 64 B: bytes moved = 2 x FETCH_SIZE = the resolved figure in all
six patterns, so the doubling of rounds 1-3 stands for the gather kernels too.  What the counters measure is LINES
fetched: a scattered 16-byte gather costs a whole line.  WRITE_SIZE is exact for the stores of these kernels
(MI355X_MICROARCH.md).  All raw values stay in the file; `calibration` compares k_reorder<true> (a kernel with a known
MINIMUM byte count) in this very window -- its excess is the line over-fetch of its two 16-byte gathers.
usage: traffic_report.py <config> <dist> <warmup> <steps> <particles> [out.json]"""
import collections, csv, glob, json, os, sys
cfg, dist, warm, steps, n = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = sys.argv[6] if len(sys.argv) > 6 else os.path.join(root, "profiles", "r05", "traffic.json")


def per_kernel(kind, counter, scale):
    f = max(glob.glob(os.path.join(root, "gpurun_out", "traffic_%s_%s_%s" % (cfg, dist, kind), "*", "*_counter_collection.csv")),
            key=os.path.getmtime)  # gpurun merges runs into the same directory: take the newest pass
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]) * scale)
    return {k: sum(v[-steps:]) / len(v[-steps:]) for k, v in per.items()}  # the timed window only


fetch, write = per_kernel("fetch", "FETCH_SIZE", 1024.0), per_kernel("write", "WRITE_SIZE", 1024.0)
n32, n64, n128 = (per_kernel("rdsize", "TCC_EA0_RDREQ_%s_sum" % s, 1.0) for s in ("32B", "64B", "128B"))
read = {k: 32.0 * n32.get(k, 0.0) + 64.0 * n64.get(k, 0.0) + 128.0 * n128.get(k, 0.0) for k in set(n32) | set(n64) | set(n128)}
# calibration in this window: k_reorder<true> reads, per particle, its slot's index (4 B) and ONE 32-byte record {position,
# id | velocity, cell id} through it (round 5; rounds 1-4: index + id + cell id + two 16-byte records = 44 B, the figure
# `read_expected_r04` keeps for comparison), plus two cell starts (cached), and writes 3 x 16 + 4 + 3 x 4 B
cal_read_expected, cal_write_expected = n * (4 + 32), n * (48 + 4 + 12)
ro = "k_reorder<true>"
cal = {"kernel": ro, "read_expected_at_least": cal_read_expected, "read_resolved": read.get(ro),
       "read_resolved_over_expected": read.get(ro, 0.0) / cal_read_expected, "read_expected_r04": n * 44,
       "read_resolved_over_expected_r04": read.get(ro, 0.0) / (n * 44.0), "fetch_size_raw": fetch.get(ro),
       "two_x_fetch_size_over_expected": 2.0 * fetch.get(ro, 0.0) / cal_read_expected,
       "requests": {"32B": n32.get(ro), "64B": n64.get(ro), "128B": n128.get(ro)},
       "write_expected": cal_write_expected, "write_raw": write.get(ro)}
out = {"window": {"config": cfg, "dist": dist, "warmup": warm, "steps": steps, "particles": n},
       "units": "bytes per launch, mean over the timed window",
       "read_resolved": read, "read_requests": {"32B": n32, "64B": n64, "128B": n128},
       "fetch_raw": fetch, "write_raw": write, "calibration": cal}
def step_kernel(prefix):
    """The step's instantiation of a neighbour kernel, whatever its trailing template arguments are this round."""
    hits = [k for k in read if k.startswith(prefix)]
    assert len(hits) == 1, (prefix, hits)
    return hits[0]


force = step_kernel("k_force_listed<false, false, false")  # <IEEE, ACCEL_ONLY, CUT, ...>: the step's launch, not the on-demand accel pass
dens = step_kernel("k_density_listed<false, false")
out["bytes_per_launch"] = {"force_integrate_bin": read[force] + write[force], "density": read[dens] + write[dens]}
out["bytes_per_launch_note"] = "size-resolved read requests (32 n32 + 64 n64 + 128 n128) + WRITE_SIZE"
allt = json.load(open(out_path)) if os.path.exists(out_path) else {}
allt["%s-%s-w%d-k%d" % (cfg, dist, warm, steps)] = out  # keyed by window: bench.py fills roofline.traffic on an exact match only
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(allt, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps({k: out[k] for k in ("bytes_per_launch", "calibration")}, indent=1))
