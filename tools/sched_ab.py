#!/usr/bin/env python3
"""Developer: the cost-guided tile schedule of K4 / K5 (WsSched) against the static equal shares, from the SAME states,
with the developer library (WS_TILE_SCHEDULE=0 is one of its environment hooks).
usage: sched_ab.py <config> <state step> <timed steps> [setting ...]
Per setting: whole-step wall clock (best of three runs of `timed steps` steps bracketed by ws_sync, no profiling) and the
per-kernel means from a profiled run; the final state must be bit-identical."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import water_sandbox_amd as ws  # noqa: E402

cfg, warm, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
L = ws.fluid.bind_library(ws.build.build_dev_library())
if cfg.startswith("pow"):  # pow19: 2^19 particles in the container the named configurations' rule gives that count
    block = ws.workloads.block_for(1 << int(cfg[3:]))
    params = ws.make_params(container_size=ws.workloads.container_for_block(block))
    pos = ws.workloads.uniform_cloud(block[0] * block[1] * block[2], 0x5EED0100 + int(cfg[3:]), list(params.ext_min), list(params.ext_max))
else:
    pos, params = ws.workloads.make_workload(cfg, "cloud")
os.environ["WS_TILE_SCHEDULE"] = "0"
w = ws.FluidWorker(pos, params, library=L)
w.run(warm)
state = w.read_vec("particles")
w.close()
ref = None
# settings: "0" = static shares; "c<classes>g<particles per group>" = the schedule with these parameters
for setting in (sys.argv[4:] or ["0", "c4g1024"]):
    os.environ["WS_TILE_SCHEDULE"] = "0" if setting == "0" else "1"
    if setting != "0":
        os.environ["WS_SCHED_CLASSES"], os.environ["WS_SCHED_GROUP"] = setting[1:].split("g")
    best = None
    v = ws.FluidWorker(pos, params, library=L)
    for rep in range(3):
        v.write_slice("particles", state)
        v.run(3)  # (the schedule needs a step's costs)
        v.sync()
        t0 = time.perf_counter()
        v.run(steps)
        v.sync()
        dt = (time.perf_counter() - t0) / steps * 1e3
        best = dt if best is None else min(best, dt)
    out = v.read_vec("particles")
    v.close()
    p = ws.FluidWorker(pos, params, library=L, profile=True)
    p.write_slice("particles", state)
    p.run(3)
    p.sync()
    p.profile_reset()
    p.run(steps)
    p.sync()
    prof = {k: round(t / c, 4) for k, (t, c) in p.profile().items() if c}
    p.close()
    same = None
    if ref is None:
        ref = out
    else:
        same = all(np.array_equal(out[f].view(np.uint32), ref[f].view(np.uint32)) for f in ("position", "velocity", "density"))
    print(json.dumps({"config": cfg, "state_step": warm, "steps": steps, "tile_schedule": setting, "ms_per_step": round(best, 4),
                      "kernel_ms": prof, "bit_identical_to_first": same}), flush=True)
