"""BASELINE.json's configurations C2, C4 and C5 as -m gpu tests, through the C ABI.

C2 (262 144 particles) is small enough for the CPU oracle: one teacher-forced step at full size, integer
artefacts bit-exact, floats within the standard tolerance.  C4 (16 777 216) and C5 (67 108 864) are checked on
ONE GPU at full size through the size-independent properties of the path plus brute-force float64 sums for a
sample of particles; C4 is also cut into four x-slabs (loopback transport) and must reproduce the single handle
bit for bit.  (C1 and C3: tests/test_gpu_parity.py, tests/test_gpu_edge.py.)"""
import numpy as np
import pytest

from util import assert_particles_close, oracle_from_params, oracle_one_step, reorder_noise_tolerances

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------
# C2 against the oracle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ieee", [False, True], ids=["hw-rcp-sqrt", "ieee-division"])
@pytest.mark.parametrize("dist,presteps", [("cloud", 0), ("lattice", 0), ("cloud", 60)])
def test_c2_full_size_one_step_against_the_oracle(oracle, ws, dist, presteps, ieee):
    """One step of all 262 144 particles from identical state: hash keys, sorted key sequence and cell offsets
    bit-exact; every float field within 4x the oracle's own reorder noise.  `presteps` first lets the GPU run
    into the collapse so that the compared step has dense, uneven neighbourhoods (teacher forcing from there)."""
    O = oracle
    pos, params = ws.workloads.make_workload("c2", dist)
    w = ws.FluidWorker(pos, params, ieee_division=ieee)
    orc = oracle_from_params(O, pos, params)
    if presteps:
        w.run(presteps)
        state = w.read_vec("particles").astype(O.PARTICLE_DTYPE)
    else:
        state = orc.particles.copy()
    want = oracle_one_step(O, orc, state, mode=O.SORT_FAST)
    keys_want, sorted_want, off_want = orc.particle_cell_indicies.copy(), orc.sorted_keys().copy(), orc.cell_offsets.copy()
    rev = oracle_one_step(O, orc, state, reverse=True, mode=O.SORT_FAST)
    w.write_slice("particles", state)
    w.run()
    got = w.read_vec("particles")
    keys, perm, off = w.sort_view()
    w.close()
    assert np.array_equal(keys, keys_want)
    assert np.array_equal(np.sort(perm), np.arange(pos.shape[0], dtype=np.uint32))
    assert np.array_equal(keys[perm], sorted_want)
    assert np.array_equal(off, off_want)
    assert_particles_close(got, want, reorder_noise_tolerances(want, rev), "c2 %s +%d" % (dist, presteps),
                           "ieee-division" if ieee else "hw-rcp-sqrt")


# ------------------------------------------------------------------------------------------------
# C4 / C5 at full size on one GPU
# ------------------------------------------------------------------------------------------------
def _brute_force_densities(q32, h, pick):
    """(density, near density) of the particles `pick` as float64 sums over ALL particles (no cell grid)."""
    k2 = 15.0 / (2.0 * np.pi * h ** 5)
    k3 = 15.0 / (np.pi * h ** 6)
    qx = np.ascontiguousarray(q32[:, 0])
    out = []
    for i in pick:
        slab = np.flatnonzero(np.abs(qx - qx[i]) <= np.float32(h * 1.001))  # x prefilter on the contiguous f32 column
        q = q32[slab].astype(np.float64)
        dist = np.sqrt(((q - q32[i].astype(np.float64)) ** 2).sum(axis=1))
        dist = dist[dist <= h]
        out.append((((h - dist) ** 2).sum() * k2 + 1e-5, ((h - dist) ** 3).sum() * k3 + 1e-5, dist.size))
    return out


def _full_size_properties(ws, name, steps_before):
    pos, params = ws.workloads.make_workload(name, "cloud")
    n = pos.shape[0]
    w = ws.FluidWorker(pos, params)
    del pos
    w.run(steps_before)
    q = w.read_vec("particles")["predicted_position"][:, :3].copy()  # what the next step hashes and sums over
    w.run(1)
    keys, perm, off = w.sort_view()
    P = w.read_vec("particles")
    w.close()
    assert np.array_equal(np.sort(perm), np.arange(n, dtype=np.uint32))      # ids conserved
    sk = keys[perm]
    assert np.all(sk[1:] >= sk[:-1])                                          # sortedness
    heads = np.flatnonzero(np.r_[True, sk[1:] != sk[:-1]])
    first = np.full(n, 999999999, np.uint32)
    first[sk[heads]] = heads.astype(np.uint32)
    assert np.array_equal(off, first)                                         # offsets = first slot or INF
    del first, heads, sk, perm, off
    # keys = hash_cell(get_cell(predicted position the step started from)), recomputed on the host
    cell = np.floor(q / np.float32(params.smoothing_radius)).astype(np.int32).view(np.uint32)
    hk = (cell[:, 0] * np.uint32(15823) + cell[:, 1] * np.uint32(9737333) + cell[:, 2] * np.uint32(440817757)) % np.uint32(n)
    assert np.array_equal(keys, hk)
    del cell, hk, keys
    mn = np.float32(list(params.ext_min)[:3]); mx = np.float32(list(params.ext_max)[:3])
    assert np.all(P["position"][:, :3] >= mn) and np.all(P["position"][:, :3] <= mx)  # wall clamp
    assert np.array_equal(P["predicted_position"][:, :3], P["position"][:, :3] + P["velocity"][:, :3] * np.float32(0.02))
    assert np.all(np.isfinite(P["density"])) and np.all(P["density"][:, 0] > 152.0)   # the self term alone is 152.79
    assert np.array_equal(P["pressure"][:, 0], np.float32(22.0) * (P["density"][:, 0] - np.float32(10.0)))
    assert np.array_equal(P["pressure"][:, 1], np.float32(2.0) * P["density"][:, 1])
    for f in ("position", "velocity", "acceleration", "predicted_position"):
        assert not P[f][:, 3].any()
    # 64 particles (48 random + the 16 densest) against brute-force float64 sums over all n
    rng = np.random.default_rng(4 + n % 97)
    pick = np.r_[rng.integers(0, n, 48), np.argsort(-P["density"][:, 0])[:16]]
    for i, (rho, rho_near, cnt) in zip(pick, _brute_force_densities(q, float(params.smoothing_radius), pick)):
        assert abs(P["density"][i, 0] - rho) <= 2e-5 * rho + 1e-3, (i, P["density"][i, 0], rho, cnt)
        assert abs(P["density"][i, 1] - rho_near) <= 2e-5 * rho_near + 1e-3, (i, P["density"][i, 1], rho_near, cnt)


def test_full_size_properties_c4(ws):
    """BASELINE.json config 4 (16 777 216 particles, container 128 x 72 x 36) on one GPU."""
    _full_size_properties(ws, "c4", 24)


def test_full_size_properties_c5(ws):
    """BASELINE.json config 5 (67 108 864 particles, container 256 x 72 x 72) on one GPU (~27 GB of HBM)."""
    _full_size_properties(ws, "c5", 8)


@pytest.mark.parametrize("overlap", [False, True], ids=["no-overlap", "overlap"])
def test_c4_in_four_slabs_matches_the_single_handle(ws, overlap):
    """BASELINE.json config 4 in its 4-GPU decomposition (four x-slabs; loopback transport on the one test GPU):
    every field of every particle bit-identical to the single handle, with and without halo / compute overlap."""
    pos, params = ws.workloads.make_workload("c4", "cloud")
    steps = 6
    w = ws.FluidWorker(pos, params)
    w.run(steps)
    want = w.read_vec("particles")
    w.close()
    got, owned = ws.slab.run_loopback(pos, params, 4, steps, overlap=overlap)
    assert sum(owned) == pos.shape[0]
    for f in want.dtype.names:
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f


@pytest.mark.parametrize("steps", [3, 190])
def test_c5_in_eight_slabs_matches_the_single_handle(ws, steps):
    """BASELINE.json config 5 (67 108 864 particles) in its 8-GPU decomposition (eight x-slabs; loopback transport on the
    one test GPU, ~100 GB of HBM): the host's global read returns, on the first and the last rank, id-ordered positions
    bit-identical to the single handle's, and the slabs own all particles between them.  190 steps: into the rebound of
    the collapsed cloud (a million migrants per slab and step, the far route by the hundred thousand)."""
    pos, params = ws.workloads.make_workload("c5", "cloud")
    w = ws.FluidWorker(pos, params)
    w.run(steps)
    want = w.read_positions()
    w.close()

    def program(s, rank):
        s.run(steps)
        got = s.read_positions(want=(rank in (0, 7)))   # the other ranks only contribute to the collective read
        return None if got is None else bool(np.array_equal(got.view(np.uint32), want.view(np.uint32))), s.num_owned()

    res = ws.slab.run_loopback_program(pos, params, 8, program)
    assert sum(r[1] for r in res) == pos.shape[0]
    assert res[0][0] is True and res[7][0] is True


@pytest.mark.parametrize("lagged", [False, True])
def test_c4_in_four_slabs_into_the_rebound_matches_the_single_handle(ws, lagged):
    """BASELINE.json config 4 in its four slabs, stepped INTO the rebound of the collapsed cloud (200 steps): by then more
    than 300 000 particles leave a slab towards one neighbour per step and more than 100 000 cross several slabs at once
    (the far route), the migration fill runs as its multi-kernel form, and the messages have grown with the fluid from a
    few thousand records to hundreds of thousands -- with the DEFAULT capacities.  Positions on the first and the last
    rank bit-identical to the single handle's.  (Rounds 1-3 stepped C4 in slabs for six steps; run further it overran.)"""
    pos, params = ws.workloads.make_workload("c4", "cloud")
    steps = 200
    w = ws.FluidWorker(pos, params)
    w.run(steps)
    want = w.read_positions()
    w.close()

    def program(s, rank):
        s.run(steps)
        got = s.read_positions(want=(rank in (0, 3)))
        st = s.stats()
        return None if got is None else bool(np.array_equal(got.view(np.uint32), want.view(np.uint32))), s.num_owned(), st

    # (lagged: WS_FLAG_LAGGED_MESSAGES, the sizes bench.py --lagged-messages / --graph run with; default: exact sizes)
    res = ws.slab.run_loopback_program(pos, params, 4, program, lagged_messages=lagged)
    assert sum(r[1] for r in res) == pos.shape[0]
    assert res[0][0] is True and res[3][0] is True
    assert max(r[2]["migration_peak"] for r in res) > 100000 and max(r[2]["far_peak"] for r in res) > 10000, [r[2] for r in res]
    for r in res:
        assert r[2]["migration_peak"] < r[2]["migration_capacity"] and r[2]["far_peak"] < r[2]["far_capacity"]
        assert r[2]["halo_peak"] < r[2]["halo_capacity"]
