"""Rank body of tests/test_gpu_rccl_two_gpus.py: one process per GPU, the library's own RCCL transport (csrc/ws_rccl.cpp)
with the REAL librccl -- needs two GPUs, so the test is skipped on this pool's one-GPU boxes.
    torchrun ... tests/dist_rccl_worker.py <out pattern> <steps> <graph 0|1>"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import water_sandbox_amd as ws  # noqa: E402


def main():
    out_path, steps, graph = sys.argv[1], int(sys.argv[2]), sys.argv[3] == "1"
    dist.init_process_group("gloo")  # bootstrap only: rank 0's unique id to every rank
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(dev)
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(65536, 1234, list(params.ext_min), list(params.ext_max))
    owner = ws.slab.assign(params, pos, world)
    sel = np.flatnonzero(owner == rank).astype(np.uint32)
    box = [ws.slab.NativeRcclTransport.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    tr = ws.slab.NativeRcclTransport(box[0], rank, world, dev)
    # graph: WS_FLAG_GRAPH | WS_FLAG_GRAPH_MULTIRANK -- a captured step with peers travels with fixed-capacity messages
    w = ws.slab.SlabWorker(pos[sel], sel, pos.shape[0], params, rank, world, tr, device=dev, graph=graph, graph_multirank=graph)
    w.run(steps // 2)
    mid = w.read_positions()              # the frame loop's collective read
    w.run(steps - steps // 2)
    rec = w.read_vec("particles")
    np.savez(out_path % rank, rec=rec, mid=mid, graph_steps=w.stats()["graph_steps"], communicators=tr.communicators())
    w.close()
    tr.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
