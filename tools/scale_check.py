#!/usr/bin/env python3
"""Developer check: large configs on one GPU (memory, 32-bit indexing) + readback-inclusive step rate."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import water_sandbox_amd as ws

for cfg, steps in (("c3", 30), ("c4", 10), ("c5", 5)):
    t0 = time.time()
    pos, params = ws.workloads.make_workload(cfg, "cloud")
    n = pos.shape[0]
    w = ws.FluidWorker(pos, params)
    t1 = time.time()
    w.run(steps); w.sync()
    t2 = time.time()
    out = w.read_positions()
    t3 = time.time()
    mn = np.float32(list(params.ext_min)[:3]); mx = np.float32(list(params.ext_max)[:3])
    ok = bool(np.all(np.isfinite(out)) and np.all(out >= mn) and np.all(out <= mx))
    moved = float(np.abs(out - pos).max())
    print("%s n=%9d gen+create %.1fs  %d steps %.1f ms/step  read_positions %.1f ms  in-box %s max-move %.3f grid %s"
          % (cfg, n, t1 - t0, steps, (t2 - t1) / steps * 1e3, (t3 - t2) * 1e3, ok, moved, w.grid_dims()), flush=True)
    if cfg == "c3":
        # the per-frame pattern of the Bevy host: read positions, then step
        import ctypes as C
        buf = np.empty((n, 3), np.float32)  # the host's persistent position buffer, as in the Bevy update()
        L = ws.load_library()
        w.pin_host_buffer(buf)
        t4 = time.time()
        for _ in range(20):
            L.ws_read_positions(w._h, buf.ctypes.data); w.run()
        w.sync()
        print("   c3 with ws_read_positions into one persistent buffer every step: %.2f ms/frame" % ((time.time() - t4) / 20 * 1e3), flush=True)
        t5 = time.time(); L.ws_read_positions(w._h, buf.ctypes.data); print("   one pinned readback: %.2f ms" % ((time.time()-t5)*1e3))
        assert np.array_equal(buf, w.read_positions())
        w.unpin_host_buffer(buf)
    w.close()
