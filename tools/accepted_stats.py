#!/usr/bin/env python3
"""Developer probe: accepted neighbours per particle (radius h) along the C3 trajectory, and their spread inside
one 64-lane wave of the sorted order (mean / max = the lane utilisation a lane-per-particle sweep can reach)."""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import water_sandbox_amd as ws  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
marks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [10, 60, 200]
pos, params = ws.workloads.make_workload(cfg, "cloud")
w = ws.FluidWorker(pos, params)
done = 0
for m in marks:
    w.run(m - done)
    done = m
    p = w.read_vec("particles")["predicted_position"][:, :3].astype(np.float64)
    c = np.floor(p / params.smoothing_radius).astype(np.int64)
    order = np.lexsort((c[:, 2], c[:, 1], c[:, 0]))  # the dense grid's z-fastest cell order
    tree = cKDTree(p)
    cnt = tree.query_ball_point(p, params.smoothing_radius, return_length=True, workers=16)[order]
    nw = cnt.size // 64
    cw = cnt[: nw * 64].reshape(nw, 64)
    print(f"step {m}: accepted/particle mean {cnt.mean():.1f} p50 {np.median(cnt):.0f} p99 {np.percentile(cnt, 99):.0f} max {cnt.max()}"
          f"  per-wave mean/max {cw.mean() / cw.max(axis=1).mean():.2f}", flush=True)
