#!/usr/bin/env python3
"""Developer probe (round 4): what serialises the per-frame position readback and the step?

    python3 tools/readback_timeline.py [c3] [sparse|settled]

Times, at one state of the trajectory, the step alone, the device->host copy alone and the frame pattern
(ws_read_positions_begin, ws_step, ws_read_positions_end) for each kind of host buffer.  Run it under
`rocprofv3 --kernel-trace --memory-copy-trace` to get the timeline (tools/readback_trace_summary.py reads it)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import water_sandbox_amd as ws

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
state = sys.argv[2] if len(sys.argv) > 2 else "sparse"
frames = int(os.environ.get("RB_FRAMES", "20"))
pos, params = ws.workloads.make_workload(cfg, "cloud")
n = pos.shape[0]
# RB_CHURN=k: k workers created, used (steps + an asynchronous readback: two streams each) and destroyed first, as bench.py
# does with its repetitions -- where the streams of THIS worker land among the runtime's hardware queues depends on it
for _ in range(int(os.environ.get("RB_CHURN", "0"))):
    c = ws.FluidWorker(pos, params, profile=True)
    c.run(3)
    c.read_positions_begin_owned()
    c.read_positions_end()
    c.close()
w = ws.FluidWorker(pos, params, profile=os.environ.get("RB_PROFILE") == "1")  # WS_FLAG_PROFILE: timed launches
w.run(5 if state == "sparse" else 400)
w.sync()


def timed(label, fn, reps=frames):
    w.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    w.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print("%-34s %.3f ms" % (label, ms), flush=True)
    return ms


print("config %s, %d particles, state %s (step %d), WS_FLAG_PROFILE %s, %s" % (cfg, n, state, w.steps_done(), os.environ.get("RB_PROFILE", "0"), os.environ.get("WS_COPY_STREAM_PRIORITY", "high") + " priority copy stream, churn " + os.environ.get("RB_CHURN", "0")), flush=True)
for k in sorted(os.environ):
    if any(t in k for t in ("HSA", "HIP", "ROC", "GPU_", "AMD")):
        print("env", k, "=", os.environ[k])
kinds = os.environ.get("RB_KINDS", "registered,hostmalloc,library").split(",")
for kind in kinds:
    if kind == "registered":
        buf = np.empty((n, 3), np.float32)
        w.pin_host_buffer(buf)
    elif kind == "hostmalloc":
        t = torch.empty((n, 3), dtype=torch.float32, pin_memory=True)
        buf = t.numpy()
    else:
        buf = None
    begin = (lambda: w.read_positions_begin(buf)) if buf is not None else (lambda: w.read_positions_begin_owned())
    # the step alone from this state would move the trajectory on: time it last, and only a few steps
    c = timed(kind + ": copy alone (begin, end)", lambda: (begin(), w.read_positions_end()), 10)
    f = timed(kind + ": frame (begin, step, end)", lambda: (begin(), w.run(1), w.read_positions_end()))
    s = timed(kind + ": step alone", lambda: w.run(1))
    print("%s: frame / max(step, copy) = %.2f" % (kind, f / max(s, c)), flush=True)
    if kind == "registered":
        w.unpin_host_buffer(buf)
w.close()
