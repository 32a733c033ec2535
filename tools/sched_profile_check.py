#!/usr/bin/env python3
"""Developer: does the tile schedule survive WS_FLAG_PROFILE (bench.py times its windows on a profiled handle)?
Whole-step wall clock of C2 from step 400, schedule on / off x profile on / off (developer library)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import water_sandbox_amd as ws
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
L = ws.fluid.bind_library(ws.build.build_dev_library())
pos, params = ws.workloads.make_workload(cfg, "cloud")
for sched in ("0", "1"):
    for prof in (False, True):
        os.environ["WS_TILE_SCHEDULE"] = sched
        w = ws.FluidWorker(pos, params, library=L, profile=prof)
        if prof:
            w.profile_select((1 << ws.fluid.KERNEL_IDS["force_integrate_bin"]) | (1 << ws.fluid.KERNEL_IDS["density"]))
        w.run(400); w.sync()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); w.run(100); w.sync(); best = min(best, (time.perf_counter() - t0) / 100 * 1e3)
        print(json.dumps({"config": cfg, "schedule": sched, "profile": prof, "ms_per_step": round(best, 4)}), flush=True)
        w.close()
