#!/usr/bin/env python3
"""Developer: steady-state A/B of library builds (tools/ab/lib<name>.so) from the SAME state: per build, the whole-step wall
clock over `steps` steps (best of three, no profiling) and the per-kernel means of a profiled run; the final states must be
bit-identical.  (tools/ablate.py times ONE step right after the state upload -- right for K4 / K5, wrong for the sort
phase, whose first step after an upload sorts from id order.)
usage: lib_ab.py <config | powK> <state step> <steps> <name> [<name> ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import water_sandbox_amd as ws  # noqa: E402

cfg, warm, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if cfg.startswith("pow"):  # pow20: 2^20 particles in the container the named configurations' rule gives that count
    block = ws.workloads.block_for(1 << int(cfg[3:]))
    params = ws.make_params(container_size=ws.workloads.container_for_block(block))
    pos = ws.workloads.uniform_cloud(block[0] * block[1] * block[2], 0x5EED0100 + int(cfg[3:]), list(params.ext_min), list(params.ext_max))
else:
    pos, params = ws.workloads.make_workload(cfg, "cloud")
w = ws.FluidWorker(pos, params)
w.run(warm)
state = w.read_vec("particles")
w.close()
ref = None
for name in sys.argv[4:]:
    L = ws.fluid.bind_library(os.path.join(ROOT, "tools", "ab", "lib%s.so" % name))
    v = ws.FluidWorker(pos, params, library=L)
    best = None
    for rep in range(3):
        v.write_slice("particles", state)
        v.run(3)
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
    print(json.dumps({"config": cfg, "state_step": warm, "steps": steps, "lib": name, "ms_per_step": round(best, 4), "kernel_ms": prof,
                      "bit_identical_to_first": same}), flush=True)
