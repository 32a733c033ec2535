#!/usr/bin/env python3
"""Developer (round 4): an experimental kernel that a patched library switches on by an environment variable at launch
time, against the shipped kernel from the SAME states: bit-equality (and the largest difference) of the records after
two steps, and the kernels' launch times.
usage: ab_switch_check.py <library name under tools/ab> <switch variable> [config] [warm steps ...]
  k4t WS_K4_CELLS   the lanes-=-candidates K4 (tools/ab/patches/k4_cells.patch)
  k5p WS_K5_PAIRS   K5 with every particle split over a lane pair (tools/ab/patches/k5_pairs.patch)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import water_sandbox_amd as ws

libname, SWITCH = sys.argv[1], sys.argv[2]
cfg = sys.argv[3] if len(sys.argv) > 3 else "c3"
warms = [int(x) for x in sys.argv[4:]] or [10, 60, 400]
L = ws.fluid.bind_library(os.path.join(ROOT, "tools", "ab", "lib%s.so" % libname))
pos, params = ws.workloads.make_workload(cfg, "cloud")
for warm in warms:
    os.environ[SWITCH] = "0"
    w = ws.FluidWorker(pos, params, library=L)
    w.run(warm)
    state = w.read_vec("particles")
    w.close()
    out = {}
    for mode in ("0", "1", "0", "1"):
        os.environ[SWITCH] = mode
        v = ws.FluidWorker(pos, params, profile=True, library=L)
        ms = {}
        for _ in range(3):
            v.write_slice("particles", state)
            v.profile_reset()
            v.run(2)   # (the first step after an upload bins on its own; time both, report the minimum)
            v.sync()
            for k, (t, c) in v.profile().items():
                if c:
                    ms.setdefault(k, []).append(t / c)
        rec = v.read_vec("particles")
        st = v.stats()
        v.close()
        out.setdefault(mode, []).append(({k: round(min(x), 4) for k, x in ms.items()}, rec, st))
    a, b = out["0"][0][1], out["1"][0][1]
    same = {f: bool(np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32))) for f in a.dtype.names}
    diff = {f: float(np.max(np.abs(a[f].astype(np.float64) - b[f].astype(np.float64)))) for f in a.dtype.names}
    scale = {f: float(np.max(np.abs(a[f]))) for f in a.dtype.names}
    print(json.dumps({"config": cfg, "warm": warm, "switch": SWITCH, "bit_identical": same, "max_abs_diff": diff, "max_abs": scale,
                      "shipped_ms": [o[0] for o in out["0"]], "experiment_ms": [o[0] for o in out["1"]],
                      "mask_overflow": [out["0"][0][2]["mask_overflow"], out["1"][0][2]["mask_overflow"]]}), flush=True)
