#!/usr/bin/env python3
"""Developer: the product library's listed kernels against the one-lane-per-particle `simple` kernels of the same
sources (developer build, WS_VARIANT=simple) over a LONG free-running trajectory at full size: every field bit-identical
at the end (any divergence on the way would be amplified, not hidden).  Also the graph-replayed step.
usage: long_identity.py <config> <steps>"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import water_sandbox_amd as ws
cfg, steps = sys.argv[1], int(sys.argv[2])
pos, params = ws.workloads.make_workload(cfg, "cloud")
res = {}
DEV = ws.fluid.bind_library(ws.build.build_dev_library())
for name in ("listed", "graph", "simple"):
    t0 = time.time()
    if name == "simple":
        os.environ["WS_VARIANT"] = "simple"
        w = ws.FluidWorker(pos, params, library=DEV)
        os.environ.pop("WS_VARIANT")
    else:
        w = ws.FluidWorker(pos, params, graph=(name == "graph"))
    w.run(steps)
    res[name] = w.read_vec("particles")
    st = w.stats()
    w.close()
    print(json.dumps({"config": cfg, "steps": steps, "variant": name, "seconds": round(time.time() - t0, 1), "mask_overflow": st["mask_overflow"],
                      "graph_steps": st["graph_steps"]}), flush=True)
ok = {}
for name in ("graph", "simple"):
    ok[name] = all(np.array_equal(res[name][f].view(np.uint32), res["listed"][f].view(np.uint32)) for f in res["listed"].dtype.names)
print(json.dumps({"config": cfg, "steps": steps, "bit_identical_to_listed": ok}))
sys.exit(0 if all(ok.values()) else 1)
