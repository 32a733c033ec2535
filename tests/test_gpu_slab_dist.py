"""The torch.distributed transport of the slab protocol, two processes on the one test GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_two_process_slabs_match_single_handle_bitwise(ws, tmp_path, overlap):
    steps = 30
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    pattern = str(tmp_path / "slab_%d.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "tests", "dist_slab_worker.py"), pattern, str(steps), overlap]
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


def test_bench_multi_rank_path_rehearsal(tmp_path):
    """bench.py's --gpus N branch (launcher env, slab workload, transport, max-over-ranks timing, one JSON line
    from rank 0), rehearsed with two gloo ranks on the one GPU; RCCL itself needs one GPU per rank."""
    import json

    env = dict(os.environ, WS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--config", "c1", "--replicate"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["warmup"] == 2 and line["scaling"] == "weak"
    assert line["unit"] == "steps/s" and line["value"] > 0 and line["config"]["particles"] == 2 * 4096
    assert line["value"] == line["global_steps_per_s"]  # steps/s of the configuration run: one step advances every particle
    shares = line["config"]["particles"] / 4194304.0
    assert abs(line["c3_equivalent_steps_per_s"] - shares * line["value"]) < 1e-9 * line["value"]
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert line["settled"] is not None and line["settled"]["steps"] == 100 and line["settled"]["warmup"] == 400
    # the like-for-like yard-stick of a multi-GPU line (no one-GPU file is committed for this toy geometry: null, but there)
    assert {"same_config_one_gpu", "speedup_vs_one_gpu_same_config", "c3_equivalent_note"} <= set(line)
    assert line["same_config_one_gpu"] is None and line["speedup_vs_one_gpu_same_config"] is None
    assert "north_star" not in line and line["roofline"]["priced_against"].startswith("hbm")
    # the sizes of rank 0's messages when the last window ended: the (rounded-up) counts themselves
    m = line["messages_rank0"]
    assert 64 <= m["migration_records"] < line["stats"]["migration_capacity"] // 8 and m["far_records_per_destination"] >= 64
    assert m["sizing"].startswith("exact")
    assert abs(m["migration_MB_per_direction"] - m["migration_records"] * 32e-6) < 1e-9
