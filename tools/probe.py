#!/usr/bin/env python3
"""Developer probe: run a workload for a while and print the per-kernel HIP-event breakdown and
the overflow-tile counters.  Usage: probe.py [config] [dist] [warmup] [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import water_sandbox_amd as ws  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
dist = sys.argv[2] if len(sys.argv) > 2 else "cloud"
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 10
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
pos, params = ws.workloads.make_workload(cfg, dist)
ieee = os.environ.get("WS_IEEE", "0") == "1"  # WS_FLAG_IEEE_DIVISION
w = ws.FluidWorker(pos, params, profile=True, ieee_division=ieee)
w.profile_select(0)
w.run(warm)
w.sync()
w.profile_select(0xFFFFFFFF)
w.profile_reset()
w.run(steps)
w.sync()
prof = {k: round(v[0] / max(v[1], 1), 4) for k, v in w.profile().items() if v[1]}
print(json.dumps({"variant": os.environ.get("WS_VARIANT", "listed") + ("+ieee" if ieee else ""), "config": cfg, "dist": dist, "warmup": warm,
                  "steps": steps, "ms": prof, "total_ms": round(sum(prof.values()), 4), "stats": w.stats()}))
w.close()
