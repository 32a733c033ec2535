"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on identical
inputs.  Integer artefacts bit-exact; floats within 4x the oracle's own reorder noise, always
ONE step from identical state (teacher forcing), SURVEY.md 8(c)."""
import numpy as np
import pytest

from util import assert_particles_close, oracle_from_params, oracle_one_step, reorder_noise_tolerances

pytestmark = pytest.mark.gpu


def _check_sort_view(O, orc, worker):
    keys, perm, off = worker.sort_view()
    # per-particle hash keys: bit-exact
    assert np.array_equal(keys, orc.particle_cell_indicies)
    # permutation validity + sorted key sequence bit-exact
    assert np.array_equal(np.sort(perm), np.arange(orc.n, dtype=np.uint32))
    assert np.array_equal(keys[perm], orc.sorted_keys())
    # cell offsets (first slot per key or INF): bit-exact
    assert np.array_equal(off, orc.cell_offsets)


BOTH_ARITHMETICS = pytest.mark.parametrize("ieee", [False, True], ids=["hw-rcp-sqrt", "ieee-division"])


def _teacher_forced(O, ws, pos, params, steps, label, ieee=False):
    """ieee: WS_FLAG_IEEE_DIVISION (the oracle's correctly rounded sqrt / division in the pair terms) instead of the
    default 1-ULP hardware forms.  The tolerance is the same for both."""
    orc = oracle_from_params(O, pos, params)
    worker = ws.FluidWorker(pos, params, ieee_division=ieee)
    state = orc.particles.copy()
    for s in range(steps):
        want = oracle_one_step(O, orc, state)
        if s < 3:
            worker.write_slice("particles", state)
            worker.run()
            _check_sort_view(O, orc, worker)
        else:
            worker.write_slice("particles", state)
            worker.run()
        got = worker.read_vec("particles")
        rev = oracle_one_step(O, orc, state, reverse=True)
        tol = reorder_noise_tolerances(want, rev)
        assert_particles_close(got, want, tol, "%s step %d" % (label, s), "ieee-division" if ieee else "hw-rcp-sqrt")
        assert np.array_equal(worker.read_positions(), got["position"][:, :3])
        state = want
    worker.close()


@BOTH_ARITHMETICS
def test_lattice_4096_cube(oracle, ws, ieee):
    pos = ws.cube_fluid(16, 16, 16)
    _teacher_forced(oracle, ws, pos, ws.default_params(), 6, "cube16", ieee)


@BOTH_ARITHMETICS
def test_planar_c1_lattice(oracle, ws, ieee):
    pos, params = ws.workloads.make_workload("c1", "lattice")
    _teacher_forced(oracle, ws, pos, params, 4, "c1-lattice", ieee)


@BOTH_ARITHMETICS
def test_planar_c1_cloud(oracle, ws, ieee):
    pos, params = ws.workloads.make_workload("c1", "cloud")
    _teacher_forced(oracle, ws, pos, params, 4, "c1-cloud", ieee)


@BOTH_ARITHMETICS
def test_reference_default_65536(oracle, ws, ieee):
    pos, params = ws.workloads.make_workload("ref", "lattice")
    _teacher_forced(oracle, ws, pos, params, 3, "ref-65536", ieee)


def test_free_running_matches_first_steps(oracle, ws):
    """Free-running (no teacher forcing) GPU vs oracle for a few steps: reported drift must stay
    small early on; integer keys stay bit-exact as long as positions agree to the cell."""
    pos = ws.cube_fluid(16, 16, 16)
    params = ws.default_params()
    orc = oracle_from_params(oracle, pos, params)
    worker = ws.FluidWorker(pos, params)
    for s in range(5):
        orc.step(oracle.SORT_EXACT)
        worker.run()
    got = worker.read_vec("particles")
    err = np.max(np.abs(got["position"] - orc.particles["position"]))
    assert err < 1e-3
    worker.close()


def _run_variant(ws, variant, pos, params, steps, ieee=False, devlib=None):
    """`listed` = the product library as it ships; `simple` = the one-thread-per-particle kernels of the same sources,
    selected through the developer build's WS_VARIANT hook (the product library has no such switch)."""
    import os

    if variant == "listed":
        w = ws.FluidWorker(pos, params, ieee_division=ieee)
    else:
        os.environ["WS_VARIANT"] = variant
        try:
            w = ws.FluidWorker(pos, params, ieee_division=ieee, library=devlib)
        finally:
            os.environ.pop("WS_VARIANT", None)
    w.run(steps)
    out = w.read_vec("particles")
    w.close()
    return out


@BOTH_ARITHMETICS
@pytest.mark.parametrize("name,dist,steps", [("c2", "cloud", 12), ("c2", "lattice", 6), ("c1", "cloud", 8), ("c2", "cloud", 150)])
def test_listed_kernels_equal_simple_kernels_bitwise(ws, devlib, name, dist, steps, ieee):
    """The listed density/force kernels (planar radius sweep, LDS compaction, neighbour lists) visit neighbours in
    the same order with the same operations as the one-thread-per-particle kernels: every field must be
    bit-identical, also after many free-running steps (any divergence would be amplified, not hidden) -- with
    either arithmetic."""
    pos, params = ws.workloads.make_workload(name, dist)
    a = _run_variant(ws, "simple", pos, params, steps, ieee, devlib)
    b = _run_variant(ws, "listed", pos, params, steps, ieee)
    for f in a.dtype.names:
        assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), f


def test_the_two_arithmetics_differ_only_in_the_last_bits(ws):
    """Hardware rcp/sqrt against correctly rounded division after ONE step from the same state: relative
    difference of the accelerations at the level of a few ULP of their magnitude, positions almost always equal."""
    pos, params = ws.workloads.make_workload("c2", "cloud")
    a = _run_variant(ws, "listed", pos, params, 1, False)
    b = _run_variant(ws, "listed", pos, params, 1, True)
    assert np.array_equal(a["density"][:, 0].view(np.uint32) != b["density"][:, 0].view(np.uint32),
                          a["density"][:, 0] != b["density"][:, 0])
    scale = float(np.max(np.abs(b["acceleration"])))
    assert float(np.max(np.abs(a["acceleration"] - b["acceleration"]))) <= 64 * np.finfo(np.float32).eps * scale
    assert float(np.max(np.abs(a["density"] - b["density"]))) <= 16 * np.finfo(np.float32).eps * float(np.max(b["density"]))
