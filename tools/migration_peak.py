#!/usr/bin/env python3
"""Developer (round 4): how many particles leave a slab per step along the benchmark trajectory?  `world` slabs of one
domain in this process (loopback transport, one GPU), the default message capacities; prints the peak
of leavers per step per slab and the default migration-message capacity next to it.
usage: migration_peak.py <config> <copies> <world> [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import water_sandbox_amd as ws

cfg, copies, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 450
block, size = ws.workloads.CONFIGS[cfg]
block = (block[0] * copies, block[1], block[2])
size = (size[0] * copies, size[1], size[2])
pos, _, n, params = ws.slab.make_dist_workload(ws, block, size, "cloud", 0, 1, seed=ws.workloads.cloud_seed(cfg))
nx = int(size[0] / 0.25) + 4


def program(s, rank):
    out, last, st = [], 0, None
    every = int(os.environ.get("MIG_EVERY", "5"))
    for k in range(steps):
        try:
            s.run(1)
            if k % every == every - 1 or k > steps - 3:
                c = s.counters()
                st = s.stats()
                out.append((k, c["left"] - last, c["owned"], c["far"]))
                last = c["left"]
        except ws.WsError as e:
            return out, st, "step %d: %s" % (k, e)
    return out, s.stats(), None


res = ws.slab.run_loopback_program(pos, params, world, program, capacity=int(n / world * 1.6))  # default message capacities
for r, (rows, st, failed) in enumerate(res):
    if failed:
        print(json.dumps({"rank": r, "FAILED": failed, "last_stats": st, "rows": rows[-4:]}), flush=True)
        continue
    peak = max(rows, key=lambda x: x[1])
    print(json.dumps({"config": "%sx%d" % (cfg, copies), "world": world, "rank": r, "particles": n,
                      "peak_leavers_per_5_steps": peak[1], "at_step": peak[0], "owned_min_max": [min(x[2] for x in rows), max(x[2] for x in rows)],
                      "far_total": rows[-1][3], "stats": st, "average_layer": n // nx,
                      "per_5_steps": [x[1] for x in rows][:90]}), flush=True)
