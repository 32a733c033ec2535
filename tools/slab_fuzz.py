#!/usr/bin/env python3
"""Developer (round 4): randomised slab-identity runs -- a random container, gravity, particle count, slab count and run
length per case, `world` loopback slabs on this one GPU against the single handle, bit for bit; the counters say which
routes were used.  usage: slab_fuzz.py [cases] [seed] [exact|lagged|graph] [benign]
(graph, round 5: the CAPTURED multi-rank step -- WS_FLAG_GRAPH | WS_FLAG_GRAPH_MULTIRANK, which travels with fixed-capacity
messages -- through the library's RCCL transport and the tests' stand-in for librccl, 2 .. 4 slabs as host threads; needs
WS_RCCL_LIBRARY = tests/libfakerccl.so and GPU_MAX_HW_QUEUES=24 in the environment, and uses the developer build)
(exact: the default sizes; lagged: WS_FLAG_LAGGED_MESSAGES; benign: gravity along x up to 20 and the benchmark clouds' particle density, 50 per unit volume, instead of up to 3 000 and up to 60 times that)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import water_sandbox_amd as ws

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)
exact = not (len(sys.argv) > 3 and sys.argv[3] in ("lagged", "graph"))
graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
benign = len(sys.argv) > 4 and sys.argv[4] == "benign"
bad = over = 0


def run_captured(pos, params, world, steps, counters):
    """`world` slabs as threads of this process on the RCCL transport code + the stand-in librccl, every step but the first
    two replayed from a captured graph; returns (records by id, owned counts, graph steps per rank)."""
    import ctypes as C
    import threading

    DEV = ws.fluid.bind_library(ws.build.build_dev_library())
    fake = C.CDLL(os.environ["WS_RCCL_LIBRARY"])
    fake.fake_rccl_errors.restype = C.c_uint32
    n = pos.shape[0]
    owner = ws.slab.assign(params, pos, world, DEV)
    uid = ws.slab.NativeRcclTransport.unique_id(DEV)
    out, seen = np.zeros(n, ws.PARTICLE_DTYPE), np.zeros(n, np.int32)
    owned, gsteps, errors = [0] * world, [0] * world, []
    created = threading.Barrier(world, timeout=120)

    def body(r):
        try:
            tr = ws.slab.NativeRcclTransport(uid, r, world, 0, library=DEV)
            sel = np.flatnonzero(owner == r).astype(np.uint32)
            w = ws.slab.SlabWorker(pos[sel], sel, n, params, r, world, tr, graph=True, graph_multirank=True, library=DEV)
            created.wait()
            w.run(steps)
            rec, ids = w.read()
            out[ids] = rec
            np.add.at(seen, ids, 1)
            owned[r], gsteps[r] = len(ids), w.stats()["graph_steps"]
            counters[r] = dict(w.counters(), **{k: v for k, v in w.stats().items() if k.endswith("_peak")})
            w.close()
            tr.close()
        except Exception as e:
            errors.append((r, repr(e)))
            created.abort()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errors:
        raise RuntimeError("rank failed: %r; stand-in's error word %d" % (errors, int(fake.fake_rccl_errors())))
    assert np.all(seen == 1) and int(fake.fake_rccl_errors()) == 0
    return out, owned, gsteps


for k in range(cases):
    world = int(rng.integers(2, 5 if graph else 9))
    size = (float(rng.choice([8.0, 16.0, 24.0, 40.0])), float(rng.choice([5.0, 9.0, 12.0])), float(rng.choice([5.0, 9.0])))
    g = float(rng.choice([0.0, 6.0, 60.0, 600.0, 3000.0])) * float(rng.choice([-1.0, 1.0]))
    n = int(rng.choice([3000, 20000, 65536, 150000, 600000]))
    steps = int(rng.integers(8, 70))
    if int(size[0] / 0.25) + 4 < 3 * world:  # every slab needs a few layers
        world = 2
    if benign:
        g = float(rng.choice([0.0, 6.0, 20.0])) * float(rng.choice([-1.0, 1.0]))
        n = int(50 * size[0] * size[1] * size[2] * float(rng.choice([0.5, 1.0, 2.0])))
    params = ws.make_params(container_size=size, gravity=(g, -9.8, float(rng.choice([0.0, 40.0])), 0.0))
    pos = ws.workloads.uniform_cloud(n, int(rng.integers(1, 1 << 30)), list(params.ext_min), list(params.ext_max))
    w = ws.FluidWorker(pos, params)
    w.run(steps)
    want = w.read_vec("particles")
    w.close()
    counters = {}
    row = {"case": k, "world": world, "container": size, "gravity_x": g, "particles": n, "steps": steps}
    try:
        if graph:
            got, owned, gsteps = run_captured(pos, params, world, steps, counters)
            row["graph_steps"] = gsteps
        else:
            got, owned = ws.slab.run_loopback(pos, params, world, steps, counters=counters, lagged_messages=not exact)
        same = all(np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)) for f in want.dtype.names)
        row.update(identical=bool(same), owned_sum_ok=sum(owned) == n, left=sum(c["left"] for c in counters.values()),
                   far=sum(c["far"] for c in counters.values()),
                   peaks=[max(c[k] for c in counters.values()) for k in ("migration_peak", "far_peak", "halo_peak")])
        bad += 0 if (same and sum(owned) == n) else 1
    except (ws.WsError, RuntimeError) as e:
        row.update(error=str(e)[:300])
        # with exact sizes an overrun can only be one of a buffer's CAPACITY (a third of all particles in one cell layer
        # against a wall ...): reported, not counted as a failure of the sizing
        if (exact or graph) and "status 3" in str(e):
            over += 1
        else:
            bad += 1
    print(json.dumps(row), flush=True)
print(json.dumps({"cases": cases, "bad": bad, "capacity_overruns": over}), flush=True)
sys.exit(1 if bad else 0)
