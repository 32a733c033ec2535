#!/usr/bin/env python3
"""Developer: how evenly is the neighbour work spread over the lanes of a wave?  Takes the C3 cloud `warm` steps in,
counts every particle's neighbours (cKDTree), orders the particles as the sort does (dense cell id, z fastest; id
inside a cell), cuts that order into waves of 64 and prints mean(count) / max(count) per wave -- the lane utilisation
a one-lane-per-particle neighbour loop can reach at best -- next to global statistics.
usage: wave_balance.py [warm steps]"""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import water_sandbox_amd as ws  # noqa: E402

warm = int(sys.argv[1]) if len(sys.argv) > 1 else 400
pos, params = ws.workloads.make_workload("c3", "cloud")
w = ws.FluidWorker(pos, params)
w.run(warm)
P = w.read_vec("particles")
dims = w.grid_dims()
w.close()
q = P["predicted_position"][:, :3].astype(np.float64)
h = float(params.smoothing_radius)
tree = cKDTree(q)
cnt = tree.query_ball_point(q, h, return_length=True, workers=-1) - 1  # without the particle itself
cell = np.floor(P["predicted_position"][:, :3] / np.float32(h)).astype(np.int64)
org = np.floor(np.array([params.ext_min[i] for i in range(3)], np.float32) / np.float32(h)).astype(np.int64) - 2
g = np.clip(cell - org, 0, np.array(dims) - 1)
cid = (g[:, 0] * dims[1] + g[:, 1]) * dims[2] + g[:, 2]
order = np.lexsort((np.arange(len(cid)), cid))
c = cnt[order]
nw = len(c) // 64
waves = c[: nw * 64].reshape(nw, 64)
util = waves.mean(axis=1) / np.maximum(waves.max(axis=1), 1)
print("step %d: neighbours mean %.1f, median %.0f, 90/99/99.9 pct %.0f / %.0f / %.0f, max %d"
      % (warm, c.mean(), np.median(c), *np.percentile(c, [90, 99, 99.9]), c.max()))
print("per wave of 64 consecutive sorted particles: sum of max over waves / sum of counts = %.3f (lane utilisation %.3f)"
      % (waves.max(axis=1).sum() * 64 / waves.sum(), waves.sum() / (waves.max(axis=1).sum() * 64)))
print("utilisation per wave: mean %.3f, 10/50/90 pct %.3f / %.3f / %.3f" % (util.mean(), *np.percentile(util, [10, 50, 90])))
occ = np.bincount(cid)
print("particles per occupied cell: mean %.2f, 99 pct %.0f, max %d; occupied cells %d" % (
    occ[occ > 0].mean(), np.percentile(occ[occ > 0], 99), occ.max(), (occ > 0).sum()))
