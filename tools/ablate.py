#!/usr/bin/env python3
"""Developer: time ONE step of each variant library from the SAME state (so that a variant whose results are not
meant to be right -- an ablation -- cannot change the workload it is timed on).
usage: ablate.py <warm steps> <lib> [<lib> ...]   (libs = names under tools/ab/; ABLATE_CONFIG=c4 for another configuration)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import water_sandbox_amd as ws  # noqa: E402

warm = int(sys.argv[1])
pos, params = ws.workloads.make_workload(os.environ.get("ABLATE_CONFIG", "c3"), "cloud")
w = ws.FluidWorker(pos, params)
w.run(warm)
state = w.read_vec("particles")
w.close()
for rep in range(2):
    for name in sys.argv[2:]:
        L = ws.fluid.bind_library(os.path.join(ROOT, "tools", "ab", "lib%s.so" % name))
        v = ws.FluidWorker(pos, params, profile=True, library=L)
        ms = {}
        for _ in range(3):
            v.write_slice("particles", state)
            v.profile_reset()
            v.run(1)
            v.sync()
            for k, (t, c) in v.profile().items():
                if c:
                    ms.setdefault(k, []).append(t / c)
        over = v.stats()["mask_overflow"]
        v.close()
        print(json.dumps({"lib": name, "warm": warm, "mask_overflow_per_step": over // 3, "ms": {k: round(min(x), 4) for k, x in ms.items()}}), flush=True)
