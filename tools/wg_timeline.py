#!/usr/bin/env python3
"""Developer: when does every workgroup of K4 / K5 start and end?  (tools/ab/libwgt.so = tools/ab/patches/wg_timeline.patch:
the two neighbour kernels stamp the 100 MHz clock at workgroup start and end.)  One step from a given state; prints, per
kernel: span, the workgroup-duration distribution, how busy the machine is over time (workgroups in flight at ten
points of the span), and how long the launch runs after its last workgroup was dispatched (the drain).
usage: wg_timeline.py <config> <state step>"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import water_sandbox_amd as ws  # noqa: E402

cfg, warm = sys.argv[1], int(sys.argv[2])
L = ws.fluid.bind_library(os.path.join(ROOT, "tools", "ab", "libwgt.so"))
hip = C.CDLL("libamdhip64.so")
pos, params = ws.workloads.make_workload(cfg, "cloud")
n = pos.shape[0]
# SCHED=0 | c<classes>g<particles per group> in the environment: the tile schedule of the kernels
setting = os.environ.get("SCHED", "0")
os.environ["WS_TILE_SCHEDULE"] = "0" if setting == "0" else "1"
if setting != "0":
    os.environ["WS_SCHED_CLASSES"], os.environ["WS_SCHED_GROUP"] = setting[1:].split("g")
w = ws.FluidWorker(pos, params, library=L)
w.run(warm)
w.sync()
n4, n5 = 2 * ((n + 63) // 64) + 64, 2 * ((n + 127) // 128) + 64  # (a scheduled launch has up to 1.5 x the workgroups)
buf = C.c_void_p()
assert hip.hipMalloc(C.byref(buf), C.c_size_t(16 * (n4 + n5))) == 0
hip.hipMemset(buf, 0, C.c_size_t(16 * (n4 + n5)))
L.ws_exp_wgt.argtypes = [C.c_void_p, C.c_ulonglong]
L.ws_exp_wgt(buf, n4)
w.run(1)
w.sync()
L.ws_exp_wgt(None, 0)
host = np.zeros(2 * (n4 + n5), np.uint64)
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
assert hip.hipMemcpy(host.ctypes.data, buf, host.nbytes, 2) == 0
w.close()
for name, a, m in (("k_density_listed", 0, n4), ("k_force_listed", n4, n5)):
    t = host[2 * a:2 * (a + m)].reshape(m, 2).astype(np.int64)
    t0, t1 = t[:, 0], t[:, 1]
    ok = t1 > 0
    base = t0[ok].min()
    s, e = (t0[ok] - base) / 100.0, (t1[ok] - base) / 100.0  # microseconds
    dur = e - s
    span = e.max()
    last_dispatch = s.max()
    order = np.argsort(e)
    pts = np.linspace(0.05, 0.999, 12) * span
    inflight = [int(np.sum((s <= p) & (e > p))) for p in pts]
    # work left when the last workgroup starts: workgroup-microseconds after that moment
    rest = np.clip(e - np.maximum(s, last_dispatch), 0, None).sum()
    xcd = np.flatnonzero(ok) % 8
    out = {"kernel": name, "config": cfg, "schedule": setting, "per_xcd_workgroups": [int(np.sum(xcd == x)) for x in range(8)], "state_step": warm, "workgroups": int(ok.sum()), "span_us": round(float(span), 1),
           "last_dispatch_us": round(float(last_dispatch), 1), "drain_us": round(float(span - last_dispatch), 1),
           "wg_duration_us": {k: round(float(np.percentile(dur, q)), 1) for k, q in (("p10", 10), ("p50", 50), ("p90", 90), ("p99", 99), ("max", 100))},
           "mean_wg_us": round(float(dur.mean()), 1),
           "in_flight_at_fraction_of_span": {("%.2f" % (p / span)): v for p, v in zip(pts, inflight)},
           "wg_us_after_last_dispatch": round(float(rest), 0),
           "busy_integral_over_peak": round(float(dur.sum() / (span * max(inflight))), 3),
           "per_xcd_end_us": [round(float(e[xcd == x].max()), 1) for x in range(8)],
           "per_xcd_wg_us_sum": [round(float(dur[xcd == x].sum()), 0) for x in range(8)]}
    print(json.dumps(out))
