"""Invariants of the oracle's passes (tier 2)."""
import numpy as np
import pytest

from util import FLOAT_FIELDS


def _stepped(oracle, pos, steps, mode, **kw):
    o = oracle.Oracle(pos, **kw)
    for _ in range(steps):
        o.step(mode)
    return o


@pytest.mark.parametrize("mode_name", ["exact", "fast"])
def test_sort_produces_valid_sorted_permutation_and_offsets(oracle, mode_name):
    mode = oracle.SORT_EXACT if mode_name == "exact" else oracle.SORT_FAST
    o = _stepped(oracle, oracle.cube_fluid(16, 16, 16, 0.1), 3, mode)
    n = o.n
    perm = o.particle_indicies
    assert np.array_equal(np.sort(perm), np.arange(n, dtype=np.uint32))  # still a permutation
    sk = o.sorted_keys()
    assert np.all(sk[1:] >= sk[:-1])                                      # ascending keys
    # cell_offsets[k] = first slot of key k, else INF (assets/bitonic_sort.wgsl:48-59)
    first = np.full(n, oracle.INF, np.uint32)
    heads = np.flatnonzero(np.r_[True, sk[1:] != sk[:-1]])
    first[sk[heads]] = heads.astype(np.uint32)
    assert np.array_equal(o.cell_offsets, first)


def test_fast_and_exact_sort_agree_on_integer_artefacts(oracle):
    pos = oracle.uniform_cloud(4096, 0x5EED0001, *oracle.get_ext((0, 0, 0), (16, 9, 9), 0.1))
    a = oracle.Oracle(pos); b = oracle.Oracle(pos)
    a.step(oracle.SORT_EXACT); b.step(oracle.SORT_FAST)
    assert np.array_equal(a.particle_cell_indicies, b.particle_cell_indicies)
    assert np.array_equal(a.sorted_keys(), b.sorted_keys())
    assert np.array_equal(a.cell_offsets, b.cell_offsets)
    # floats differ only by summation order inside a bucket
    for f in FLOAT_FIELDS:
        np.testing.assert_allclose(a.particles[f], b.particles[f], rtol=1e-4, atol=1e-4)


def test_bitonic_network_depends_on_incoming_permutation(oracle):
    """The reference never resets particle_indicies (src/fluid_compute.rs:306): the unstable
    network's output depends on the permutation left by the previous step."""
    pos = oracle.cube_fluid(16, 16, 16, 0.1)
    a = oracle.Oracle(pos)
    a.step(oracle.SORT_EXACT); a.step(oracle.SORT_EXACT)
    b = oracle.Oracle(pos)
    b.step(oracle.SORT_EXACT)
    b.particle_indicies[:] = np.arange(b.n, dtype=np.uint32)  # what the reference does NOT do
    b.step(oracle.SORT_EXACT)
    assert np.array_equal(a.sorted_keys(), b.sorted_keys())
    assert not np.array_equal(a.particle_indicies, b.particle_indicies)


def test_momentum_conservation_without_gravity(oracle):
    """The density-pressure pair term dir * shared * slope / (rho_i rho_j) and the viscosity term are
    antisymmetric in (i, j); the near-pressure term divides by near_density_j * density_i and is
    only approximately so (assets/simulation.wgsl:253-268).  So sum_i a_i is small next to
    sum_i |a_i|, not zero: a loose invariant."""
    pos = oracle.uniform_cloud(4096, 7, [-2, -2, -2, 0], [2, 2, 2, 0])
    o = oracle.Oracle(pos, gravity=[0, 0, 0, 0], ext_min=[-8, -8, -8, 0], ext_max=[8, 8, 8, 0])
    rng = np.random.default_rng(1)
    o.particles["velocity"][:, :3] = rng.standard_normal((o.n, 3)).astype(np.float32)
    o.hash_particles(); o.sort(oracle.SORT_EXACT); o.calculate_cell_offsets(); o.update_density()
    o.update_pressure_force()
    acc = o.particles["acceleration"][:, :3].astype(np.float64)
    scale = np.abs(acc).sum(0)
    assert np.all(np.abs(acc.sum(0)) < 0.05 * scale)


def test_reverse_order_changes_only_rounding(oracle):
    pos = oracle.uniform_cloud(4096, 3, *oracle.get_ext((0, 0, 0), (8, 4, 4), 0.1))
    mn, mx = oracle.get_ext((0, 0, 0), (8, 4, 4), 0.1)
    a = oracle.Oracle(pos, ext_min=mn, ext_max=mx); b = oracle.Oracle(pos, ext_min=mn, ext_max=mx)
    b.set_reverse_order(True)
    a.step(oracle.SORT_EXACT); b.step(oracle.SORT_EXACT)
    assert np.array_equal(a.particle_cell_indicies, b.particle_cell_indicies)
    np.testing.assert_allclose(a.particles["density"], b.particles["density"], rtol=1e-5)
    np.testing.assert_allclose(a.particles["position"], b.particles["position"], atol=1e-5)


def test_planar_config_stays_planar(oracle):
    # config 1: z pinned to 0 by the container
    pos = oracle.cube_fluid(64, 64, 1, 0.1)
    mn, mx = oracle.get_ext((0, 0, 0), (16, 18, 0.2), 0.1)
    o = oracle.Oracle(pos, ext_min=mn, ext_max=mx)
    for _ in range(5):
        o.step(oracle.SORT_EXACT)
    assert not np.any(o.particles["position"][:, 2])
    assert not np.any(o.particles["velocity"][:, 2])
