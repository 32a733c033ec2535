#!/usr/bin/env python3
"""Developer probe: wall-clock ms/step without any profiling events.
usage: wall.py [config] [dist] [warmup] [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import water_sandbox_amd as ws
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
dist = sys.argv[2] if len(sys.argv) > 2 else "cloud"
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 10
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
pos, params = ws.workloads.make_workload(cfg, dist)
w = ws.FluidWorker(pos, params)
w.run(warm); w.sync()
t0 = time.perf_counter(); w.run(steps); w.sync(); dt = time.perf_counter() - t0
print(json.dumps({"config": cfg, "dist": dist, "warmup": warm, "steps": steps,
                  "ms_per_step": round(dt / steps * 1e3, 4)}))
w.close()
