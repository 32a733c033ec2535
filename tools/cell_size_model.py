#!/usr/bin/env python3
"""Developer: what would cells of edge h/2 (a 5x5x5 stencil, 25 z-contiguous runs of 5 cells) buy K4 over cells of
edge h (3x3x3, 9 runs of 3 cells)?  Takes the C3 cloud `warm` steps in with the library, then replays K4's phase 1 on
the host for both cell sizes: particles in the dense z-fastest cell order, waves of 64 consecutive particles, every
run walked in lockstep in trips of 4 candidates -- a wave pays, per run, the trips of its busiest lane.
Prints candidates per particle, trips per wave, accept-mask words per particle and the size of the cell table the
scan has to cover.  (A model of instruction counts, not a timing: phase 1 is ~75 VALU instructions per trip and K4
is VALU-bound in the dense state)
usage: cell_size_model.py [warm steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import water_sandbox_amd as ws  # noqa: E402

warm = int(sys.argv[1]) if len(sys.argv) > 1 else 400
pos, params = ws.workloads.make_workload("c3", "cloud")
w = ws.FluidWorker(pos, params)
w.run(warm)
P = w.read_vec("particles")
w.close()
q = P["predicted_position"][:, :3]
h = np.float32(params.smoothing_radius)
n = q.shape[0]
lo_ext = np.array([params.ext_min[i] for i in range(3)], np.float32)
hi_ext = np.array([params.ext_max[i] for i in range(3)], np.float32)


def model(div):
    """cells of edge h / div, stencil radius `div` cells"""
    edge = h / np.float32(div)
    org = np.floor(lo_ext / edge).astype(np.int64) - 2 * div
    dims = np.floor(hi_ext / edge).astype(np.int64) + 2 * div + 1 - org
    g = np.clip(np.floor(q / edge).astype(np.int64) - org, div, dims - 1 - div)
    cid = (g[:, 0] * dims[1] + g[:, 1]) * dims[2] + g[:, 2]
    ncell = int(dims.prod())
    start = np.zeros(ncell + 1, np.int64)
    np.cumsum(np.bincount(cid, minlength=ncell), out=start[1:])
    order = np.lexsort((np.arange(n), cid))
    c = cid[order]
    nw = n // 64
    cw = c[: nw * 64].reshape(nw, 64)
    trips = np.zeros(nw, np.int64)       # lockstep: sum over runs of the busiest lane's trips
    lane_trips = np.zeros((nw, 64), np.int64)
    cand = np.zeros((nw, 64), np.int64)
    for dx in range(-div, div + 1):
        for dy in range(-div, div + 1):
            cc = cw + (dx * dims[1] + dy) * dims[2]
            ln = start[cc + div + 1] - start[cc - div]
            t = (ln + 3) // 4
            trips += t.max(axis=1)
            lane_trips += t
            cand += ln
    words = (cand + 31) // 32
    print("cells of h/%d: %d x %d x %d = %.1f M cells (scan: %.0f MB read + written); runs per particle %d" % (
        div, *dims, ncell / 1e6, ncell * 8 / 1e6, (2 * div + 1) ** 2))
    print("   candidates per particle %.1f; trips per lane %.1f; trips per wave (lockstep) %.1f; lane utilisation of the "
          "trips %.2f; mask words per particle %.2f (max %d)" % (
              cand.mean(), lane_trips.mean(), trips.mean(), lane_trips.mean() / trips.mean(), words.mean(), words.max()))
    return trips.mean(), cand.mean()


print("C3 cloud, step %d" % warm)
t1, c1 = model(1)
t2, c2 = model(2)
print("h/2 against h: candidates x %.2f, phase-1 trips per wave x %.2f" % (c2 / c1, t2 / t1))
