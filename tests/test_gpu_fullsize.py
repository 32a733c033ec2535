"""The BENCHMARKED configuration against the oracle at FULL size, in both regimes and both arithmetics.

One teacher-forced step of ALL particles (assets/simulation.wgsl:130-310 as restated by the oracle, fast sort
mode): per-particle hash keys, sorted key sequence and cell offsets bit-exact; every float field within the standard
tolerance (4 x the oracle's own reorder noise + 4 ulp).  Cases:

  C3 (4 194 304, the config bench.py reports) from the initial uniform cloud           -- the sparse regime
  C3 from the GPU's own state after 400 steps (the settled window of the bench line)  -- the dense regime
  C4 (16 777 216) from the initial cloud
  C2 (262 144) from the GPU's own state after 400 steps

The oracle's results for a case are computed once and shared by the two arithmetics.  Oracle cost at C3: ~1 s per
step in the sparse state, a few seconds in the dense one (16 OpenMP threads)."""
import numpy as np
import pytest

from util import assert_particles_close, oracle_from_params, oracle_one_step, reorder_noise_tolerances

pytestmark = pytest.mark.gpu

CASES = [("c3", "cloud", 0), ("c3", "cloud", 400), ("c4", "cloud", 0), ("c2", "cloud", 400), ("c3", "lattice", 0)]


@pytest.fixture(scope="module", params=CASES, ids=["%s-%s-step%d" % c for c in CASES])
def case(request, oracle, ws):
    """(name, pos, params, state, want, tol, keys, sorted keys, offsets) of one case, oracle side computed once."""
    name, dist, presteps = request.param
    O = oracle
    pos, params = ws.workloads.make_workload(name, dist)
    orc = oracle_from_params(O, pos, params)
    if presteps:
        w = ws.FluidWorker(pos, params)
        w.run(presteps)
        state = w.read_vec("particles").astype(O.PARTICLE_DTYPE)
        w.close()
    else:
        state = orc.particles.copy()
    want = oracle_one_step(O, orc, state, mode=O.SORT_FAST)
    ints = (orc.particle_cell_indicies.copy(), orc.sorted_keys().copy(), orc.cell_offsets.copy())
    rev = oracle_one_step(O, orc, state, reverse=True, mode=O.SORT_FAST)
    tol = reorder_noise_tolerances(want, rev)
    del rev, orc
    yield "%s %s +%d full size" % (name, dist, presteps), pos, params, state, want, tol, ints


@pytest.mark.parametrize("ieee", [False, True], ids=["hw-rcp-sqrt", "ieee-division"])
def test_one_teacher_forced_step_of_every_particle_against_the_oracle(ws, case, ieee):
    label, pos, params, state, want, tol, (keys_want, sorted_want, off_want) = case
    n = pos.shape[0]
    w = ws.FluidWorker(pos, params, ieee_division=ieee)
    w.write_slice("particles", state)
    w.run()
    got = w.read_vec("particles")
    keys, perm, off = w.sort_view()
    stats = w.stats()
    w.close()
    assert np.array_equal(keys, keys_want), "per-particle hash keys"
    assert np.array_equal(np.sort(perm), np.arange(n, dtype=np.uint32)), "permutation"
    assert np.array_equal(keys[perm], sorted_want), "sorted key sequence"
    assert np.array_equal(off, off_want), "cell offsets"
    del keys, perm, off
    assert_particles_close(got, want, tol, label, "ieee-division" if ieee else "hw-rcp-sqrt")
    # pressure = f(density) bit for bit (simulation.wgsl:192-193), on the product's own densities
    assert np.array_equal(got["pressure"][:, 0], np.float32(params.pressure_scalar) *
                          (got["density"][:, 0] - np.float32(params.target_density)))
    assert np.array_equal(got["pressure"][:, 1], np.float32(params.near_pressure_scalar) * got["density"][:, 1])
    assert stats["mask_overflow"] >= 0
