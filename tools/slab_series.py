#!/usr/bin/env python3
"""Developer (round 4): per-step leavers of every slab for one slab_fuzz.py case (fixed-size messages, so it never
overruns): what a sizing rule has to survive.  usage: slab_series.py <cases> <seed> <case index> [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import water_sandbox_amd as ws

os.environ["WS_SLAB_FIXED_MESSAGES"] = "1"
cases, seed, want = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed)
for k in range(cases):  # the same draws as slab_fuzz.py
    world = int(rng.integers(2, 9))
    size = (float(rng.choice([8.0, 16.0, 24.0, 40.0])), float(rng.choice([5.0, 9.0, 12.0])), float(rng.choice([5.0, 9.0])))
    g = float(rng.choice([0.0, 6.0, 60.0, 600.0, 3000.0])) * float(rng.choice([-1.0, 1.0]))
    n = int(rng.choice([3000, 20000, 65536, 150000, 600000]))
    steps = int(rng.integers(8, 70))
    if int(size[0] / 0.25) + 4 < 3 * world:
        world = 2
    gz = float(rng.choice([0.0, 40.0]))
    cloud_seed = int(rng.integers(1, 1 << 30))
    if k == want:
        break
if len(sys.argv) > 4:
    steps = int(sys.argv[4])
params = ws.make_params(container_size=size, gravity=(g, -9.8, gz, 0.0))
pos = ws.workloads.uniform_cloud(n, cloud_seed, list(params.ext_min), list(params.ext_max))


def program(s, rank):
    out, last = [], 0
    for _ in range(steps):
        s.run(1)
        c = s.counters()
        out.append(c["left"] - last)
        last = c["left"]
    return out


res = ws.slab.run_loopback_program(pos, params, world, program)
print(json.dumps({"case": want, "world": world, "container": size, "gravity_x": g, "particles": n, "steps": steps}))
for r, rows in enumerate(res):
    print("rank", r, rows)
