#!/usr/bin/env python3
"""Developer check: C4 (16.7 M particles) cut into four x-slabs on one GPU (loopback transport) against the single
handle, with and without the halo / compute overlap: every field bit for bit."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import water_sandbox_amd as ws
pos, params = ws.workloads.make_workload("c4", "cloud")
w = ws.FluidWorker(pos, params); w.run(6); want = w.read_vec("particles"); w.close()
for ov in ("0", "1"):
    os.environ["WS_SLAB_OVERLAP"] = ov
    t0 = time.time()
    got, owned = ws.slab.run_loopback(pos, params, 4, 6)
    ok = all(np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)) for f in want.dtype.names)
    print("c4 16.7M particles, 4 slabs, overlap", ov, "bit-identical:", ok, "owned", owned, "%.1fs" % (time.time() - t0), flush=True)
