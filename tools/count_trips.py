#!/usr/bin/env python3
"""Developer (needs a library with the trip-count instrumentation patched in: tools/ab_build.sh count "" count_trips and
WSFLUID_LIBRARY=tools/ab/libcount.so): how many trips of four candidates a K4 wave executes and with how many lanes,
how many list entries its phase 2 walks, and the same for K5's iterator loop -- per wave, C3 cloud, sparse and settled."""
import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import water_sandbox_amd as ws
L = ws.fluid.load_library()
pos, params = ws.workloads.make_workload("c3", "cloud")
for warm in (10, 400):
    w = ws.FluidWorker(pos, params); w.run(warm); w.sync()
    out = (C.c_uint32 * 8)(); L.ws_exp_read(out)
    w.run(4); w.sync(); L.ws_exp_read(out)
    waves = pos.shape[0] / 64 * 4
    print("warm", warm, "| K5 iterator: wave-trips per wave %.1f, active lanes per trip %.1f |" % (out[5] / waves, out[6] / max(out[5], 1)), "wave-trips per wave %.1f" % (out[1] / waves), "active lanes per trip %.1f" % (out[3] / max(out[1], 1)), "phase-2 iterations per wave %.1f" % (out[2] / waves), "flushes per wave %.1f" % (out[4] / waves))
    w.close()
