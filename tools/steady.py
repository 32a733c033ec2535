#!/usr/bin/env python3
"""Developer: per-kernel launch times of variant libraries (tools/ab/lib<name>.so) in the STEADY step loop -- `steps`
steps after `warm` steps of the trajectory, each library on its own handle (tools/ablate.py times the first step after an
upload instead, which bins and places differently).
usage: steady.py <config> <warm> <steps> <lib> [<lib> ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import water_sandbox_amd as ws  # noqa: E402

cfg, warm, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
pos, params = ws.workloads.make_workload(cfg, "cloud")
ref = None
for name in sys.argv[4:]:
    L = ws.fluid.bind_library(os.path.join(ROOT, "tools", "ab", "lib%s.so" % name))
    v = ws.FluidWorker(pos, params, profile=True, library=L)
    v.run(warm)
    v.sync()
    v.profile_reset()
    v.run(steps)
    v.sync()
    ms = {k: round(t / c, 4) for k, (t, c) in v.profile().items() if c}
    rec = v.read_vec("particles")
    if ref is None:
        ref = rec
    same = all(np.array_equal(rec[f].view(np.uint32), ref[f].view(np.uint32)) for f in rec.dtype.names)
    v.close()
    print(json.dumps({"lib": name, "config": cfg, "steps": [warm, warm + steps], "ms": ms, "sum_ms": round(sum(ms.values()), 4),
                      "bit_identical_to_first": bool(same)}), flush=True)
