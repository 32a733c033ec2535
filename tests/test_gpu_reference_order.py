"""Reference-order mode (WS_FLAG_REFERENCE_ORDER): the reference's passes executed literally on the GPU, by the
TEST-ONLY build of the library (tests/libwsfluid_refcheck.so; the shipped libwsfluid.so does not contain them).
Against the oracle's exact mode EVERYTHING must be bit-identical, free-running: the permutation the
bitonic network leaves behind, hash keys, cell offsets and every float of the 80-byte records."""
import numpy as np
import pytest

import golden_util as G
from util import oracle_from_params

pytestmark = pytest.mark.gpu

FIELDS = ("position", "density", "pressure", "velocity", "acceleration", "predicted_position")


def _compare(w, orc, label):
    keys, perm, off = w.sort_view()
    assert np.array_equal(perm, orc.particle_indicies), label + ": particle_indicies"
    assert np.array_equal(keys, orc.particle_cell_indicies), label + ": particle_cell_indicies"
    assert np.array_equal(off, orc.cell_offsets), label + ": cell_offsets"
    got = w.read_vec("particles")
    for f in FIELDS:
        assert np.array_equal(got[f].view(np.uint32), orc.particles[f].view(np.uint32)), label + ": " + f


@pytest.mark.parametrize("name,dist", [("c1", "lattice"), ("c1", "cloud")])
def test_free_running_bit_identical_to_oracle(oracle, ws, refcheck, name, dist):
    pos, params = ws.workloads.make_workload(name, dist)
    orc = oracle_from_params(oracle, pos, params)
    w = ws.FluidWorker(pos, params, reference_order=True, library=refcheck)
    _compare(w, orc, "t=0")  # identity index buffers, zero fields
    for s in range(1, 13):
        orc.step(oracle.SORT_EXACT)
        w.run()
        _compare(w, orc, "%s-%s step %d" % (name, dist, s))
    w.close()


def test_cube_4096_with_param_change_and_reset(oracle, ws, refcheck):
    pos = ws.cube_fluid(16, 16, 16)
    params = ws.default_params()
    orc = oracle_from_params(oracle, pos, params)
    w = ws.FluidWorker(pos, params, reference_order=True, library=refcheck)
    for s in range(5):
        orc.step(oracle.SORT_EXACT); w.run()
    _compare(w, orc, "before change")
    p2 = ws.make_params(gravity=(1.0, -4.0, 0.5, 0.0), viscosity_strength=0.3, smoothing_radius=0.3)
    w.set_params(p2)
    orc2 = oracle_from_params(oracle, pos, p2)
    orc2.set_particles(orc.particles)
    orc2.particle_indicies[:] = orc.particle_indicies
    orc2.particle_cell_indicies[:] = orc.particle_cell_indicies
    orc2.cell_offsets[:] = orc.cell_offsets
    for s in range(4):
        orc2.step(oracle.SORT_EXACT); w.run()
    _compare(w, orc2, "after set_params")
    w.reset(pos)  # despawn_liquid: particles <- initial, index buffers <- identity
    orc3 = oracle_from_params(oracle, pos, p2)
    for s in range(3):
        orc3.step(oracle.SORT_EXACT); w.run()
    _compare(w, orc3, "after reset")
    w.close()


@pytest.mark.parametrize("name", G.CASES)
def test_reference_order_reproduces_golden_bitwise(ws, refcheck, name):
    g = G.load(name)
    pos = G.initial_positions(ws.workloads if int(g["gen_is_cloud"]) else ws, g)
    w = ws.FluidWorker(pos, ws.make_params(container_size=tuple(float(x) for x in g["container_size"])),
                       reference_order=True, library=refcheck)
    for s in range(1, 11):
        w.run()
        if s == 1:
            keys, perm, off = w.sort_view()
            assert np.array_equal(keys, g["keys_by_id_1"])
            assert np.array_equal(keys[perm], g["sorted_keys_1"])
            assert np.array_equal(off, g["cell_offsets_1"])
        if "position_%d" % s in g:
            got = w.read_vec("particles")
            assert np.array_equal(got["position"][:, :3].view(np.uint32), g["position_%d" % s].view(np.uint32)), s
            assert np.array_equal(got["velocity"][:, :3].view(np.uint32), g["velocity_%d" % s].view(np.uint32)), s
            if "density_%d" % s in g:
                assert np.array_equal(got["density"].view(np.uint32), g["density_%d" % s].view(np.uint32)), s
                assert np.array_equal(got["acceleration"][:, :3].view(np.uint32), g["acceleration_%d" % s].view(np.uint32)), s
    w.close()


def test_fast_path_agrees_with_reference_order_within_reorder_noise(oracle, ws, refcheck):
    """The production kernels against the literal GPU execution of the reference, one step from the same
    state: same integer artefacts, floats within the reorder-noise tolerance."""
    from util import assert_particles_close, oracle_one_step, reorder_noise_tolerances

    pos, params = ws.workloads.make_workload("ref", "lattice")  # the reference's own 65 536-particle default
    ref = ws.FluidWorker(pos, params, reference_order=True, library=refcheck)
    fast = ws.FluidWorker(pos, params)
    ref.run(5)
    state = ref.read_vec("particles")
    fast.write_slice("particles", state)
    ref.run(); fast.run()
    a, b = ref.read_vec("particles"), fast.read_vec("particles")
    orc = oracle_from_params(oracle, pos, params)
    fwd = oracle_one_step(oracle, orc, state.astype(oracle.PARTICLE_DTYPE))
    rev = oracle_one_step(oracle, orc, state.astype(oracle.PARTICLE_DTYPE), reverse=True)
    assert_particles_close(b, a, reorder_noise_tolerances(fwd, rev), "fast vs reference-order")
    ka, pa, oa = ref.sort_view(); kb, pb, ob = fast.sort_view()
    assert np.array_equal(ka, kb) and np.array_equal(ka[pa], kb[pb]) and np.array_equal(oa, ob)
    ref.close(); fast.close()


def test_the_product_library_refuses_reference_order(ws):
    """libwsfluid.so ships only the MI355X-native path: the validation mode is compiled into the test-only
    library alone, and asking the product for it fails loudly (WS_ERR_UNSUPPORTED), never silently falls back."""
    with pytest.raises(ws.WsError) as e:
        ws.FluidWorker(ws.cube_fluid(4, 4, 4), ws.default_params(), reference_order=True)
    assert e.value.status == 6  # WS_ERR_UNSUPPORTED
