#!/usr/bin/env python3
"""Developer probe: evolve a workload on the GPU and print cell-occupancy / neighbour statistics."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import water_sandbox_amd as ws  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
dist = sys.argv[2] if len(sys.argv) > 2 else "cloud"
marks = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [10, 60, 200]
pos, params = ws.workloads.make_workload(cfg, dist)
w = ws.FluidWorker(pos, params)
done = 0
h = params.smoothing_radius
for m in marks:
    w.run(m - done)
    done = m
    P = w.read_vec("particles")
    pred = P["predicted_position"][:, :3]
    cell = np.floor(pred / h).astype(np.int64)
    cell -= cell.min(0)
    dims = cell.max(0) + 1
    cid = (cell[:, 0] * dims[1] + cell[:, 1]) * dims[2] + cell[:, 2]
    occ = np.bincount(cid)
    nz = occ[occ > 0]
    # candidates per particle = sum over 27 cells of occupancy: estimate via 3D box filter
    grid = occ.reshape(-1) if occ.size == dims.prod() else np.pad(occ, (0, dims.prod() - occ.size))
    grid = grid.reshape(dims).astype(np.float64)
    pad = np.pad(grid, 1)
    box = np.zeros_like(grid)
    for a in range(3):
        for b in range(3):
            for c in range(3):
                box += pad[a:a + dims[0], b:b + dims[1], c:c + dims[2]]
    cand = box.reshape(-1)[cid]
    dens = P["density"][:, 0]
    # accepted neighbours ~ density / W-average is not exact; report density and speed instead
    spd = np.linalg.norm(P["velocity"][:, :3], axis=1)
    q = lambda a: [float(np.percentile(a, p)) for p in (50, 90, 99, 100)]
    print("step %4d  occupied cells %8d  occupancy/cell p50/90/99/max %s  candidates/particle mean %.1f p50/90/99/max %s"
          % (m, nz.size, q(nz), cand.mean(), q(cand)))
    print("           density p50/90/99/max %s  speed p50/90/99/max %s  y-range %.2f..%.2f"
          % (q(dens), q(spd), pred[:, 1].min(), pred[:, 1].max()))
w.close()
