"""The torch.distributed transport of the slab protocol, two processes on the one test GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_process_slabs_match_single_handle_bitwise(ws, tmp_path):
    steps = 30
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    pattern = str(tmp_path / "slab_%d.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "tests", "dist_slab_worker.py"), pattern, str(steps)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(32768, 77, list(params.ext_min), list(params.ext_max))
    w = ws.FluidWorker(pos, params)
    w.run(steps)
    want = w.read_vec("particles")
    w.close()
    got = np.zeros_like(want)
    seen = np.zeros(len(want), np.int32)
    for r in range(2):
        d = np.load(pattern % r)
        got[d["ids"]] = d["rec"]
        np.add.at(seen, d["ids"], 1)
    assert np.all(seen == 1)
    for f in want.dtype.names:
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f
