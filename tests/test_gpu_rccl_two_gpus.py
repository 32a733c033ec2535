"""The library's RCCL transport with the real librccl between two GPUs: direct launches and (WS_FLAG_GRAPH | WS_FLAG_GRAPH_MULTIRANK:
fixed-capacity messages) captured hipGraphs against the single handle, bit for bit.  SKIPPED on a one-GPU box -- this pool's; there the same
transport code runs with peers through the tests' stand-in for librccl (tests/test_gpu_fake_rccl.py).  No multi-GPU run
of this library has happened yet: this is the first thing to run on a node that has two."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpus():
    try:
        import torch

        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs (real RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("graph", ["0", "1"])
def test_two_ranks_through_native_rccl_match_the_single_handle_bitwise(ws, tmp_path, graph):
    steps = 60
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    pattern = str(tmp_path / "rccl_%d.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29561", os.path.join(ROOT, "tests", "dist_rccl_worker.py"), pattern, str(steps), graph]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(65536, 1234, list(params.ext_min), list(params.ext_max))
    w = ws.FluidWorker(pos, params)
    w.run(steps // 2)
    want_mid = w.read_positions()
    w.run(steps - steps // 2)
    want = w.read_vec("particles")
    w.close()
    for r in range(2):
        d = np.load(pattern % r)
        assert int(d["communicators"]) == 2
        assert np.array_equal(d["mid"], want_mid)
        for f in want.dtype.names:
            assert np.array_equal(d["rec"][f].view(np.uint32), want[f].view(np.uint32)), (f, r)
        assert (int(d["graph_steps"]) >= 50) == (graph == "1")
