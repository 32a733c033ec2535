#!/usr/bin/env python3
"""Developer: the library's RCCL transport on a communicator of one rank, entry points called directly -- and what
sequences of transports in one process do at exit.
usage: rccl_one_rank_probe.py <alltoall|allgather|both|none> <null|side> [two|two_live|worker_first]
  two          a first transport is created and destroyed before the one that runs the collectives
  two_live     both alive while the collectives run
  worker_first a one-rank slab worker steps on the first transport before it is destroyed (tests/test_gpu_slab.py's order)"""
import faulthandler
import os
import sys

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import water_sandbox_amd as ws

which, stream_kind = sys.argv[1], sys.argv[2]
mode = sys.argv[3] if len(sys.argv) > 3 else ""
T = ws.slab.NativeRcclTransport
first = None
if mode in ("two", "two_live", "worker_first"):
    first = T(T.unique_id(), 0, 1, 0)
    if mode == "worker_first":
        params = ws.make_params(container_size=(16.0, 9.0, 9.0))
        pos = ws.workloads.uniform_cloud(32768, 7, list(params.ext_min), list(params.ext_max))
        w = ws.slab.SlabWorker(pos, np.arange(pos.shape[0], dtype=np.uint32), pos.shape[0], params, 0, 1, first)
        w.run(12)
        w.read()
        w.close()
    if mode != "two_live":
        first.close()
        first = None
    print("first transport done", flush=True)
tr = T(T.unique_id(), 0, 1, 0)
t = tr.struct
side = torch.cuda.Stream()
st = side if stream_kind == "side" else torch.cuda.current_stream()
with torch.cuda.stream(st):
    src = torch.arange(1 << 16, dtype=torch.int32, device="cuda")
    fns = {"alltoall": [t.alltoall_dev], "allgather": [t.allgather_dev], "both": [t.alltoall_dev, t.allgather_dev], "none": []}[which]
    for fn in fns:
        dst = torch.zeros_like(src)
        rc = fn(t.ctx, src.data_ptr(), dst.data_ptr(), src.numel() * 4, st.cuda_stream)
        torch.cuda.synchronize()
        print(which, stream_kind, "rc", rc, "equal", bool(torch.equal(dst, src)), flush=True)
tr.close()
if first is not None:
    first.close()
print("closed", flush=True)
