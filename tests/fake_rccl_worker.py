"""Child process of tests/test_gpu_fake_rccl.py: WS_RCCL_LIBRARY names the tests' stand-in for librccl (tests/fake_rccl/)
and the slab handles are made by the DEVELOPER build of the library (tests/libwsfluid_dev.so: the only build that honours
that variable), so ws_rccl_transport_create binds IT -- and the library's RCCL transport (csrc/ws_rccl.cpp: peer arithmetic, segment
order, grouped send / recv, the all-to-all and the all-gather, two communicators bound to two streams, ncclCommSplit) runs with real
peers: `world` slab handles in this one process, one host thread each, on the one GPU.  Prints one JSON object."""
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
DEV = ws.fluid.bind_library(ws.build.build_dev_library())  # same sources as the product + csrc/ws_devhooks.h's hooks


def calls():
    out = (C.c_ulonglong * 4)()
    fake.fake_rccl_calls(out)
    return [int(x) for x in out]


def run_world(pos, params, world, program, graph, exact=False):
    n = pos.shape[0]
    owner = ws.slab.assign(params, pos, world, DEV)
    uid = ws.slab.NativeRcclTransport.unique_id(DEV)
    results, errors = [None] * world, []
    # Ranks share ONE process here, and HIP refuses a legacy-stream call of one thread while another thread captures a
    # graph; real ranks are processes.  The library itself no longer uses the legacy stream, but creation (allocations)
    # and capture are kept apart anyway: everybody is created before anybody steps.
    created = threading.Barrier(world, timeout=120)

    def body(r):
        try:
            tr = ws.slab.NativeRcclTransport(uid, r, world, 0, library=DEV)  # (ncclCommInitRank + ncclCommSplit: rendezvous of all ranks)
            sel = np.flatnonzero(owner == r).astype(np.uint32)
            # a captured step with peers: WS_FLAG_GRAPH | WS_FLAG_GRAPH_MULTIRANK, messages at their FIXED capacities (the
            # default of a captured multi-rank step); direct cases: lagged or exact sizes
            w = ws.slab.SlabWorker(pos[sel], sel, n, params, r, world, tr, graph=graph, graph_multirank=graph, exact_messages=exact,
                                   lagged_messages=not exact and not graph, library=DEV)
            created.wait()
            results[r] = (program(w, r), tr.communicators())
            w.close()
            tr.close()
        except Exception as e:  # surfaced below
            errors.append((r, repr(e)))
            created.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise RuntimeError("rank failed: %r; stand-in's error word %d" % (errors, int(fake.fake_rccl_errors())))
    return results


def bits_equal(a, b):
    return all(np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)) for f in a.dtype.names)


def main():
    out = {"library": os.environ["WS_RCCL_LIBRARY"], "cases": []}
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))  # tilted: migration
    pos = ws.workloads.uniform_cloud(65536, 1234, list(params.ext_min), list(params.ext_max))
    steps = int(os.environ.get("FAKE_RCCL_STEPS", "60"))
    global want
    single = ws.FluidWorker(pos, params)
    single.run(steps // 2)
    want_mid = single.read_positions()
    single.run(steps - steps // 2)
    want = single.read_vec("particles")
    single.close()

    def program(w, r):
        w.run(steps // 2)
        mid = w.read_positions()          # the frame loop's collective read in the middle (two all-gathers)
        w.run(steps - steps // 2)
        rec = w.read_vec("particles")
        again = None
        if os.environ.get("FAKE_RCCL_DEBUG"):
            again = [int(fake.fake_rccl_errors())]
            for kind in ("vec", "pos", "vec"):
                try:
                    ok = bool(bits_equal(w.read_vec("particles"), want)) if kind == "vec" else bool(
                        np.array_equal(w.read_positions(), want["position"][:, :3]))
                except Exception as e:
                    ok = repr(e)[:60]
                again += [ok, int(fake.fake_rccl_errors())]
        own = w.read()[1]                 # (rank-local: the ids this slab owns at the end)
        return rec, mid, w.stats(), w.counters(), own, again, int(fake.fake_rccl_errors())  # (while the communicators live)

    for world, graph, exact in ((2, False, False), (3, False, False), (4, False, False), (2, True, False), (4, True, False), (3, False, True)):
        # (the last case: WS_FLAG_EXACT_MESSAGES -- the size all-gathers on both communicators, sends and receives of
        # different sizes in one group, the packed far messages)
        # Captured cases travel with messages of FIXED size (what WS_FLAG_GRAPH chooses with peers): a change of the
        # message sizes would ask for a new capture, and hipGraphInstantiate / hipGraphExecDestroy in the middle of a run
        # wait for the whole device -- including the transport kernels of the other ranks of THIS process, which are
        # waiting for this rank (real ranks are processes).  The direct cases run with sizes that follow the fluid.
        before = calls()
        res = run_world(pos, params, world, program, graph, exact)
        after = calls()
        err = 0
        for r in res:
            err |= r[0][6]
        case = {"world": world, "graph": graph, "exact_messages": exact, "fake_errors": err,
                "communicators": [c for _, c in res],
                "sendrecv_ops": after[0] - before[0], "allgathers": after[1] - before[1], "new_communicators": after[2] - before[2],
                "alltoalls": after[3] - before[3],
                "graph_steps": [r[0][2]["graph_steps"] for r in res],
                "migrated": sum(r[0][3]["left"] for r in res),
                "mid_frame_positions_identical": [bool(np.array_equal(r[0][1], want_mid)) for r in res],
                "reads_repeated": [r[0][5] for r in res],
                "bit_identical_to_single_handle": [bool(bits_equal(r[0][0], want)) for r in res]}
        for k, r in enumerate(res):
            if not bits_equal(r[0][0], want):
                bad = np.flatnonzero(np.any(r[0][0]["position"].view(np.uint32) != want["position"].view(np.uint32), axis=1))
                owner_of = np.full(pos.shape[0], -1)
                for q, rq in enumerate(res):
                    owner_of[rq[0][4]] = q
                by_owner = np.bincount(owner_of[bad] + 1, minlength=world + 1).tolist()
                stale_mid = int(np.sum(np.all(r[0][0]["position"][bad, :3] == want_mid[bad], axis=1)))
                zero = int(np.sum(np.all(r[0][0]["position"][bad] == 0, axis=1)))
                case.setdefault("diagnostics", []).append(
                    {"rank": k, "positions_differ": int(bad.size), "differing_by_final_owner(-1,0,1..)": by_owner,
                     "owned_counts": [int(len(rq[0][4])) for rq in res], "equal_to_mid_frame_positions": stale_mid, "all_zero": zero, "first": bad[:6].tolist(), "last": bad[-3:].tolist(),
                     "fields": [f for f in want.dtype.names if not np.array_equal(r[0][0][f].view(np.uint32), want[f].view(np.uint32))]})
        out["cases"].append(case)
        print("case done: %r" % (case,), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
