"""Child process of tests/test_gpu_fake_rccl.py::test_thin_and_thick_slabs_agree_on_the_halo_stream: eight slab handles in
this process on the tests' stand-in for librccl, in a container only 20 cell layers wide: the equal cuts give the slabs two
or three layers each.  Whether the halos overlap the early kernels (and so travel on the communication stream = the
transport's second communicator) must be decided alike on every rank: a two-layer slab deciding for itself, as until
round 4, would exchange its halos on the other communicator than its three-layer neighbours and nobody would ever receive
them.  Prints one JSON object."""
import ctypes as C
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import water_sandbox_amd as ws

fake = C.CDLL(os.environ["WS_RCCL_LIBRARY"])
fake.fake_rccl_errors.restype = C.c_uint32
DEV = ws.fluid.bind_library(ws.build.build_dev_library())  # the developer build: the only one that honours WS_RCCL_LIBRARY
world, steps = 8, 40
params = ws.make_params(container_size=(4.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))
pos = ws.workloads.uniform_cloud(20000, 4321, list(params.ext_min), list(params.ext_max))
single = ws.FluidWorker(pos, params)
single.run(steps)
want = single.read_vec("particles")
single.close()
owner = ws.slab.assign(params, pos, world, DEV)
uid = ws.slab.NativeRcclTransport.unique_id(DEV)
results, errors = [None] * world, []
created = threading.Barrier(world, timeout=120)


def body(r):
    try:
        tr = ws.slab.NativeRcclTransport(uid, r, world, 0, library=DEV)
        sel = np.flatnonzero(owner == r).astype(np.uint32)
        w = ws.slab.SlabWorker(pos[sel], sel, pos.shape[0], params, r, world, tr, library=DEV)
        created.wait()
        dims = np.zeros(3, np.uint32)
        w._check(w._L.ws_grid_dims(w._h, dims.ctypes.data))
        w.run(steps)
        rec = w.read_vec("particles")
        results[r] = (int(dims[0]) - 2, bool(all(np.array_equal(rec[f].view(np.uint32), want[f].view(np.uint32)) for f in want.dtype.names)),
                      w.num_owned(), int(fake.fake_rccl_errors()))
        w.close()
        tr.close()
    except Exception as e:
        errors.append((r, repr(e)))
        created.abort()


threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
for t in threads:
    t.start()
for t in threads:
    t.join()
print(json.dumps({"errors": errors, "owned_layers": [r[0] for r in results if r], "identical": [r[1] for r in results if r],
                  "owned": [r[2] for r in results if r], "fake_errors": [r[3] for r in results if r]}))
