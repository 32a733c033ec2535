"""N > 1 on CPU: world_size-2 gloo run of the slab protocol model (tests/dist_cpu_slab_model.py) against
the single-domain oracle.  Floats differ only by summation order (each slab hashes into its own table)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_slab_model_matches_single_domain_oracle(oracle, ws, tmp_path):
    from util import oracle_from_params

    steps = 6
    pattern = str(tmp_path / "rank_%d.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", OMP_NUM_THREADS="2", WSO_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "tests", "dist_cpu_slab_model.py"), pattern, str(steps)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    params = ws.make_params(container_size=(8.0, 5.0, 5.0), gravity=(5.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(4096, 11, list(params.ext_min), list(params.ext_max))
    orc = oracle_from_params(oracle, pos, params)
    for _ in range(steps):
        orc.step(oracle.SORT_FAST)
    got = np.zeros_like(orc.particles)
    seen = np.zeros(len(pos), np.int32)
    counts = []
    for r in range(2):
        d = np.load(pattern % r)
        got[d["ids"]] = d["state"]
        np.add.at(seen, d["ids"], 1)
        counts.append(len(d["ids"]))
    assert np.all(seen == 1), "particle conservation across migration"
    first = np.bincount(ws.slab.assign(params, pos, 2), minlength=2)
    assert counts != list(first), "particles must have migrated in this test"
    for f, tol in (("position", 2e-5), ("velocity", 2e-3), ("density", 2e-3)):
        err = np.max(np.abs(got[f].astype(np.float64) - orc.particles[f].astype(np.float64)))
        scale = max(1.0, float(np.max(np.abs(orc.particles[f]))))
        assert err <= tol * scale, (f, err)


def test_two_rank_slab_model_capacity_overrun_fails_on_every_rank_alike(tmp_path):
    """A halo message too small for the boundary layer: the sender clamps and sets its sticky error bit, the bit
    reaches every rank with the per-step all-gather, and BOTH ranks stop at the same step, two steps later,
    outside any collective -- no rank is left waiting in one (csrc/ws_slab.inc, slab_consume_status)."""
    pattern = str(tmp_path / "rank_%d.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", OMP_NUM_THREADS="2", WSO_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29543", os.path.join(ROOT, "tests", "dist_cpu_slab_model.py"), pattern, "8", "16"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    failed = [int(np.load(pattern % r)["failed_at"]) for r in range(2)]
    assert failed[0] == failed[1] and 2 <= failed[0] <= 3, failed


def test_two_rank_global_reads_and_a_recut(oracle, ws, tmp_path):
    """The host's per-frame boundary on slabs, on CPU with two gloo ranks: every frame both ranks read the id-ordered
    positions of ALL particles (counts all-gather + records all-gather + scatter by id: csrc/ws_slab.inc slab_gather) and
    must see the same array; before step 3 the slabs are re-cut to equal particle counts with the library's own cut rule
    (ws_slab_balanced_cuts) from the gathered state.  The run still matches the single-domain oracle and conserves
    particles."""
    from util import oracle_from_params

    steps = 6
    pattern = str(tmp_path / "rank_%d.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29545", OMP_NUM_THREADS="2", WSO_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29545", os.path.join(ROOT, "tests", "dist_cpu_slab_model.py"), pattern, str(steps), "1024", "3"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    d = [np.load(pattern % r) for r in range(2)]
    assert np.array_equal(d[0]["frames"], d[1]["frames"]) and d[0]["frames"].shape[0] == steps   # every rank, every frame
    assert np.array_equal(d[0]["cuts"], d[1]["cuts"])                                             # every rank decides alike
    params = ws.make_params(container_size=(8.0, 5.0, 5.0), gravity=(5.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(4096, 11, list(params.ext_min), list(params.ext_max))
    nx = int(d[0]["cuts"][-1])
    assert list(d[0]["cuts"]) != [0, nx // 2, nx], "the tilted gravity should have moved the cut"
    owned = [int(x["owned_after_recut"]) for x in d]
    assert sum(owned) == 4096 and abs(owned[0] - owned[1]) < 4096 // 4, owned
    orc = oracle_from_params(oracle, pos, params)
    for k in range(steps):
        err = np.max(np.abs(d[0]["frames"][k].astype(np.float64) - orc.particles["position"][:, :3].astype(np.float64)))
        assert err <= 2e-5 * 8.0, (k, err)
        orc.step(oracle.SORT_FAST)
    seen = np.zeros(len(pos), np.int32)
    for x in d:
        np.add.at(seen, x["ids"], 1)
    assert np.all(seen == 1)
