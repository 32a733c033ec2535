#!/usr/bin/env python3
"""Developer soak run (round 3): long free-running trajectories through the paths added this round, checking what must
hold however long it runs -- ids conserved, positions finite and inside the container, slabs == single handle bitwise.
  1. C3 cloud, 6 000 steps with WS_FLAG_GRAPH (replayed steps; the 64-bit scan tickets; a parameter push every step)
  2. C2 cloud on 4 in-process slabs (C++ local transport through ctypes is not wired: the Python loopback), 1 500 steps
     with a tilted gravity, a re-cut every 100 steps, a radius change at 500 and 1 000, a reset at 750 -- against one handle
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import water_sandbox_amd as ws

t0 = time.time()
pos, params = ws.workloads.make_workload("c3", "cloud")
w = ws.FluidWorker(pos, params, graph=True)
for chunk in range(12):
    for _ in range(500):
        w.set_params(params)
        w.run(1)
    p = w.read_positions()
    mn, mx = np.float32(list(params.ext_min)[:3]), np.float32(list(params.ext_max)[:3])
    assert np.isfinite(p).all() and (p >= mn).all() and (p <= mx).all()
    print("c3 graph steps", (chunk + 1) * 500, "ok", "replayed", w.stats()["graph_steps"], "%.1fs" % (time.time() - t0), flush=True)
keys, perm, off = w.sort_view()
assert np.array_equal(np.sort(perm), np.arange(pos.shape[0], dtype=np.uint32))
w.close()

size = ws.workloads.CONFIGS["c2"][1]
params = ws.make_params(container_size=size, gravity=(5.0, -9.8, 0.0, 0.0))
pos = ws.workloads.uniform_cloud(262144, 5, list(params.ext_min), list(params.ext_max))
events = {500: ws.make_params(container_size=size, gravity=(5.0, -9.8, 0.0, 0.0), smoothing_radius=0.35),
          1000: ws.make_params(container_size=size, gravity=(-5.0, -9.8, 2.0, 0.0), smoothing_radius=0.15)}


def program(wk, rank=0):
    cur = params
    out = {}
    for f in range(1500):
        if f == 750:
            wk.reset(pos)
        if f in events:
            cur = events[f]
        if f % 100 == 0 and hasattr(wk, "rebalance"):
            wk.rebalance()
        wk.set_params(cur)
        wk.run(1)
        if f % 250 == 249:
            out[f] = wk.read_positions()
    return out, wk.read_vec("particles")


single = ws.FluidWorker(pos, params)
want, want_rec = program(single)
single.close()
print("c2 single done %.1fs" % (time.time() - t0), flush=True)
for seen, rec in ws.slab.run_loopback_program(pos, params, 4, program):
    for f in want:
        assert np.array_equal(seen[f].view(np.uint32), want[f].view(np.uint32)), f
    for name in want_rec.dtype.names:
        assert np.array_equal(rec[name].view(np.uint32), want_rec[name].view(np.uint32)), name
print("c2 on 4 slabs: 1500 steps with re-cuts, two radius changes and a reset bit-identical to one handle  %.1fs" % (time.time() - t0))
