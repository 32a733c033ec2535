#!/usr/bin/env python3
"""Developer: 3000 steps of the C3 cloud on one handle -- positions finite and inside the container every 500 steps,
the accept-mask overflow counter, finite accelerations and the density range at the end."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import water_sandbox_amd as ws
pos, params = ws.workloads.make_workload("c3", "cloud")
w = ws.FluidWorker(pos, params)
t0 = time.time()
for chunk in range(6):
    w.run(500); w.sync()
    p = w.read_positions()
    print("steps", (chunk + 1) * 500, "finite", bool(np.isfinite(p).all()), "min", p.min(0), "max", p.max(0), "elapsed %.1fs" % (time.time() - t0), "overflow", w.stats().get("mask_overflow"), flush=True)
P = w.read_vec("particles")
print("acc finite", bool(np.isfinite(P["acceleration"]).all()), "dens min/max", P["density"][:, 0].min(), P["density"][:, 0].max())
w.close()
