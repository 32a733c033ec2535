#!/usr/bin/env python3
"""Developer one-off: BASELINE.json config 5 (67 108 864 particles) in its EIGHT-slab decomposition on one GPU
(loopback transport, ~60 GB of HBM): bit-identical to the single handle after a few steps, and the host's global
read returns the same id-ordered positions on every rank."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import water_sandbox_amd as ws

t0 = time.time()
pos, params = ws.workloads.make_workload("c5", "cloud")
steps = 3
w = ws.FluidWorker(pos, params)
w.run(steps)
want = w.read_positions()
w.close()
print("single handle done %.1fs" % (time.time() - t0), flush=True)


def program(s, rank):
    s.run(steps)
    got = s.read_positions(want=(rank in (0, 7)))
    return None if got is None else bool(np.array_equal(got.view(np.uint32), want.view(np.uint32))), s.num_owned()


res = ws.slab.run_loopback_program(pos, params, 8, program)
print(res, "%.1fs" % (time.time() - t0))
assert sum(r[1] for r in res) == pos.shape[0] and res[0][0] is True and res[7][0] is True
print("C5 in eight slabs: id-ordered positions after %d steps bit-identical to the single handle" % steps)
