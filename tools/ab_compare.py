#!/usr/bin/env python3
"""Developer: two builds of the library (tools/ab/lib<a>.so, lib<b>.so) stepped side by side from the same initial
state: are the records bit-identical after `steps` steps?  usage: ab_compare.py <a> <b> [config] [steps ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import water_sandbox_amd as ws

a, b = sys.argv[1], sys.argv[2]
cfg = sys.argv[3] if len(sys.argv) > 3 else "c3"
marks = [int(x) for x in sys.argv[4:]] or [12, 70, 420]
pos, params = ws.workloads.make_workload(cfg, "cloud")
libs = [ws.fluid.bind_library(os.path.join(ROOT, "tools", "ab", "lib%s.so" % n)) for n in (a, b)]
workers = [ws.FluidWorker(pos, params, library=L) for L in libs]
done = 0
for m in marks:
    for w in workers:
        w.run(m - done)
    done = m
    ra, rb = (w.read_vec("particles") for w in workers)
    same = {f: bool(np.array_equal(ra[f].view(np.uint32), rb[f].view(np.uint32))) for f in ra.dtype.names}
    print(json.dumps({"config": cfg, "steps": m, "bit_identical": same, "mask_overflow": [w.stats()["mask_overflow"] for w in workers]}), flush=True)
