#!/usr/bin/env python3
"""Developer: a two-slab step of C3x2 (8.4 M particles) on ONE GPU through the library's RCCL transport code and the tests'
stand-in for librccl (stream-ordered: no host rendezvous inside a transport call), to look at when the halo kernels run
beside the early K4 -- with the communication stream at the highest priority (the product) or at the step stream's
(WS_SLAB_COMM_PRIORITY=default, a hook of the developer build).  Run it under rocprofv3 --kernel-trace and feed the trace
to this script again:
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/slab_timeline.py run <steps>
    python3 tools/slab_timeline.py report <kernel_trace.csv> <steps>
Two slabs share the one GPU here, so every duration is inflated by the other slab's kernels; the comparison between the
two priorities is what the numbers are for."""
import csv
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(steps):
    import numpy as np

    import water_sandbox_amd as ws

    os.environ.setdefault("WS_RCCL_LIBRARY", ws.build.build_fake_rccl())
    DEV = ws.fluid.bind_library(ws.build.build_dev_library())
    world = 2
    block, size = ws.workloads.CONFIGS["c3"]
    block, size = (block[0] * 2, block[1], block[2]), (size[0] * 2, size[1], size[2])
    uid = ws.slab.NativeRcclTransport.unique_id(DEV)
    created = threading.Barrier(world, timeout=300)
    errors = []

    def body(r):
        try:
            pos, ids, n_global, params = ws.slab.make_dist_workload(ws, block, size, "cloud", r, world, seed=ws.workloads.cloud_seed("c3"))
            tr = ws.slab.NativeRcclTransport(uid, r, world, 0, library=DEV)
            w = ws.slab.SlabWorker(pos, ids, n_global, params, r, world, tr, lagged_messages=True, library=DEV)
            created.wait()
            w.run(steps)
            w.sync()
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
    print(json.dumps({"steps": steps, "errors": errors}))
    sys.exit(1 if errors else 0)


def report(path, steps):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "?")))
    rows.sort()
    per = {}
    for s, e, name, q in rows:
        per.setdefault(name, []).append((s, e, q))
    out = {}
    for name, v in per.items():
        if len(v) < steps:
            continue
        tail = v[-(len(v) * 2 // 3):]  # (the first third: creation, the first steps' full-capacity messages)
        d = [(e - s) / 1e3 for s, e, _ in tail]
        out[name] = {"launches": len(v), "mean_us": round(sum(d) / len(d), 1), "max_us": round(max(d), 1), "queues": sorted({q for _, _, q in v})}
    # from the end of a slab's k_reorder to the end of the halo-A unpack that follows on the SAME process (either slab:
    # the two run in lockstep) -- how long the early range has to cover
    reorder = [e for s, e, q in per.get("k_reorder<true>", [])]
    unpack = sorted(e for name, v in per.items() if name.startswith("k_halo_unpack") for s, e, q in v)
    gaps = []
    import bisect
    for e in reorder[len(reorder) // 3:]:
        k = bisect.bisect_right(unpack, e)
        if k < len(unpack):
            gaps.append((unpack[k] - e) / 1e3)
    span = (rows[-1][1] - rows[0][0]) / 1e6
    print(json.dumps({"comm_stream_priority": os.environ.get("WS_SLAB_COMM_PRIORITY", "highest"), "trace_span_ms": round(span, 1),
                      "reorder_end_to_next_unpack_end_us": {"mean": round(sum(gaps) / max(len(gaps), 1), 1), "max": round(max(gaps or [0]), 1)},
                      "kernels": {k: out[k] for k in sorted(out) if any(t in k for t in ("halo", "fake", "density", "force", "reorder", "migrate", "fill"))}}, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]))
    else:
        report(sys.argv[2], int(sys.argv[3]))
