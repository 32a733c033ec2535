#!/usr/bin/env python3
"""Developer: what per-kernel timing costs the step.  C3 cloud, steps 5..25, with no kernels timed, with the two
neighbour kernels timed (what bench.py does), with K5 alone, and with all five -- ms per step, twice each."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import water_sandbox_amd as ws
pos, params = ws.workloads.make_workload("c3", "cloud")
for mask, label in ((0, "no events"), ((1 << ws.fluid.KERNEL_IDS["force_integrate_bin"]) | (1 << ws.fluid.KERNEL_IDS["density"]), "K4+K5 events"), (1 << ws.fluid.KERNEL_IDS["force_integrate_bin"], "K5 events"), (0xFFFFFFFF, "all events")):
    for rep in range(2):
        w = ws.FluidWorker(pos, params, profile=True)
        w.profile_select(mask)
        w.run(5); w.sync()
        t0 = time.perf_counter(); w.run(20); w.sync(); t = time.perf_counter() - t0
        print(label, "%.4f ms/step" % (t / 20 * 1e3), flush=True)
        w.close()
