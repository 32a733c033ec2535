#!/usr/bin/env python3
"""Developer: as rccl_one_rank_probe.py, without torch in the process -- the library is loaded first, so it and librccl
run on the SYSTEM's HIP runtime and librccl (with torch imported first they run on the ones torch bundles: the same
sonames, whichever is loaded first serves everybody).  usage: rccl_one_rank_probe_notorch.py [collectives: 0|1] [rccl_first]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import water_sandbox_amd as ws

run = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
L = ws.load_library()
hip = C.CDLL("libamdhip64.so.7")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipFree.argtypes = [C.c_void_p]
n = 1 << 16
src, dst, st = C.c_void_p(), C.c_void_p(), C.c_void_p()
T = ws.slab.NativeRcclTransport
rccl_first = len(sys.argv) > 2 and sys.argv[2] == "rccl_first"  # the unique id and the transport before any other HIP call of the process
if rccl_first:
    tr = T(T.unique_id(), 0, 1, 0)
assert hip.hipMalloc(C.byref(src), n * 4) == 0 and hip.hipMalloc(C.byref(dst), n * 4) == 0 and hip.hipStreamCreate(C.byref(st)) == 0
if not rccl_first:
    tr = T(T.unique_id(), 0, 1, 0)
t = tr.struct
host = np.arange(n, dtype=np.int32)
assert hip.hipMemcpy(src, host.ctypes.data, n * 4, 1) == 0
if run:
    for name, fn in (("alltoall", t.alltoall_dev), ("allgather", t.allgather_dev)):
        assert hip.hipMemcpy(dst, np.zeros(n, np.int32).ctypes.data, n * 4, 1) == 0
        rc = fn(t.ctx, src, dst, n * 4, st)
        assert hip.hipStreamSynchronize(st) == 0
        back = np.empty(n, np.int32)
        assert hip.hipMemcpy(back.ctypes.data, dst, n * 4, 2) == 0
        print(name, "rc", rc, "equal", bool(np.array_equal(back, host)), flush=True)
tr.close()
hip.hipFree(src); hip.hipFree(dst)
print("closed", flush=True)
