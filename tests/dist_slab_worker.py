"""Rank body of tests/test_gpu_slab_dist.py: 2 processes share cuda:0 and run the slab protocol through
TorchDistTransport on the gloo backend (two RCCL ranks cannot share one device; gloo moves the same
device tensors through the same torch.distributed calls)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import water_sandbox_amd as ws  # noqa: E402


def main():
    out_path = sys.argv[1]
    steps = int(sys.argv[2])
    overlap = len(sys.argv) < 4 or sys.argv[3] != "0"  # "0": WS_FLAG_NO_OVERLAP
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream(device=0))
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(32768, 77, list(params.ext_min), list(params.ext_max))
    owner = ws.slab.assign(params, pos, world)
    sel = np.flatnonzero(owner == rank).astype(np.uint32)
    tr = ws.slab.TorchDistTransport(rank, world, 0, data_group=None, ctrl_group=None)
    w = ws.slab.SlabWorker(pos[sel], sel, pos.shape[0], params, rank, world, tr, device=0,
                           stream=torch.cuda.current_stream().cuda_stream, overlap=overlap)
    w.run(steps)
    rec, ids = w.read()
    np.savez(out_path % rank, rec=rec, ids=ids)
    w.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
