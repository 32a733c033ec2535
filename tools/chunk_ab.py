#!/usr/bin/env python3
"""Developer: whole-step time of the chunked step (K4 / K5 of different x-chunks on two streams) against the plain
five-launch step, from the SAME states, with the developer library (WS_CHUNKS is one of its environment hooks).
usage: chunk_ab.py <config> <state step> <timed steps> <chunks> [<chunks> ...]     (chunks = 1: the plain step)
Prints one JSON line per chunk count: ms per step (wall clock around `timed steps` steps bracketed by ws_sync; best of
three) and whether the final state is bit-identical to the plain step's."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import water_sandbox_amd as ws  # noqa: E402

cfg, warm, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
chunks = [int(x) for x in sys.argv[4:]]  # (CHUNK_MODE in the environment: how the chunks are dealt to streams)
os.environ["WS_CHUNK_MODE"] = os.environ.get("CHUNK_MODE", "2")
os.environ["WS_CHUNK_COMB"] = os.environ.get("CHUNK_COMB", "0")
L = ws.fluid.bind_library(ws.build.build_dev_library())
pos, params = ws.workloads.make_workload(cfg, "cloud")
os.environ.pop("WS_CHUNKS", None)
w = ws.FluidWorker(pos, params, library=L)
w.run(warm)
state = w.read_vec("particles")
w.close()
ref = None
for c in chunks:
    os.environ["WS_CHUNKS"] = str(abs(c))
    os.environ.pop("WS_CHUNK_GRID_FRAC", None)
    if c < 0:  # negative: launch bounds of 1.6 x N / C instead of N (experiment)
        os.environ["WS_CHUNK_GRID_FRAC"] = os.environ.get("CHUNK_FRAC", "1.6")
    v = ws.FluidWorker(pos, params, library=L)
    best = None
    for rep in range(3):
        v.write_slice("particles", state)
        v.run(2)  # (the first step after an upload is the unchunked one)
        v.sync()
        t0 = time.perf_counter()
        v.run(steps)
        v.sync()
        dt = (time.perf_counter() - t0) / steps * 1e3
        best = dt if best is None else min(best, dt)
    out = v.read_vec("particles")
    v.close()
    same = None
    if ref is None:
        ref = out
    else:
        same = all(np.array_equal(out[f].view(np.uint32), ref[f].view(np.uint32)) for f in ("position", "velocity", "density"))
    print(json.dumps({"config": cfg, "state_step": warm, "steps": steps, "chunks": c, "mode": os.environ["WS_CHUNK_MODE"], "comb": os.environ["WS_CHUNK_COMB"], "ms_per_step": round(best, 4),
                      "bit_identical_to_first": same}), flush=True)
