#!/usr/bin/env python3
"""CPU restatement (oracle) timings for C1..C3 on this host: fast sort mode with all granted threads, and
single-thread exact mode at C1.  Test-infrastructure timing, printed as JSON lines."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import water_sandbox_amd as ws
from oracle import oracle as O
from util import oracle_from_params

def run(cfg, dist, mode, threads, steps):
    O.set_threads(threads)
    pos, params = ws.workloads.make_workload(cfg, dist)
    orc = oracle_from_params(O, pos, params)
    orc.step(mode)
    t0 = time.perf_counter()
    for _ in range(steps):
        orc.step(mode)
    dt = time.perf_counter() - t0
    print(json.dumps({"config": cfg, "dist": dist, "particles": orc.n, "mode": "exact" if mode == O.SORT_EXACT else "fast",
                      "threads": threads, "steps": steps, "steps_per_s": round(steps / dt, 3)}), flush=True)

T = O.default_threads()
for dist in ("cloud", "lattice"):
    run("c1", dist, O.SORT_FAST, T, 200)
    run("c2", dist, O.SORT_FAST, T, 20)
    run("c3", dist, O.SORT_FAST, T, 3)
run("c1", "lattice", O.SORT_EXACT, 1, 50)
run("c1", "cloud", O.SORT_EXACT, 1, 50)
