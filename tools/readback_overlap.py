#!/usr/bin/env python3
"""Developer probe: frame time of the Bevy host's per-frame pattern at C3 -- positions read back every frame --
with the blocking ws_read_positions and with the begin / step / end form that overlaps the copy with the next step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import water_sandbox_amd as ws
pos, params = ws.workloads.make_workload("c3", "cloud")
n = pos.shape[0]
w = ws.FluidWorker(pos, params)
w.run(10); w.sync()
for kind in ("registered", "torch-pinned"):
    if kind == "registered":
        buf = np.empty((n, 3), np.float32); w.pin_host_buffer(buf)
    else:
        t = torch.empty((n, 3), dtype=torch.float32, pin_memory=True); buf = t.numpy()
    for label, loop in (("serial", lambda: (w.read_positions_into(buf), w.run())),
                        ("overlap", lambda: (w.read_positions_begin(buf), w.run(), w.read_positions_end())),
                        ("overlap2", lambda: (w.read_positions_begin(buf), w.run(2), w.read_positions_end()))):
        w.sync(); t0 = time.time()
        for _ in range(20): loop()
        w.sync(); print(kind, label, "%.2f ms/frame" % ((time.time() - t0) / 20 * 1e3), flush=True)
    if kind == "registered": w.unpin_host_buffer(buf)
w.close()
