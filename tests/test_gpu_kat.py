"""Known-answer tests of SURVEY.md 8(c) on the PRODUCT path, bit for bit.

K1 (hash_particles) and K6 (integrate) have no reductions, so with contraction off their outputs can equal the
oracle's to the last bit (assets/simulation.wgsl:130-141, 271-310).  A particle without a neighbour inside the
smoothing radius has a one-term density sum and a zero force sum -- no summation order, no reorder noise -- so
EVERY field of such a particle must match the oracle bit for bit, with either arithmetic, also free-running
through wall hits."""
import numpy as np
import pytest

from util import assert_particles_bitwise, oracle_from_params

pytestmark = pytest.mark.gpu

BOTH_ARITHMETICS = pytest.mark.parametrize("ieee", [False, True], ids=["hw-rcp-sqrt", "ieee-division"])


def _arith(ieee):
    return "ieee-division" if ieee else "hw-rcp-sqrt"


@BOTH_ARITHMETICS
def test_kat7_single_free_particle_gravity_only(oracle, ws, ieee):
    """KAT 7: after one step v.y = -9.8/60, y = y0 + v.y/60, pred = pos + v/50 -- and the product's record equals the
    oracle's bit for bit (N = 1: all 27 stencil offsets alias onto bucket 0; the multiplicity path reproduces the 27-fold
    self density)."""
    pos = np.float32([[1.0, 2.0, 3.0]])
    params = ws.default_params()
    orc = oracle_from_params(oracle, pos, params)
    w = ws.FluidWorker(pos, params, ieee_division=ieee)
    for step in range(3):
        orc.step(oracle.SORT_EXACT)
        w.run()
        got = w.read_vec("particles")
        assert_particles_bitwise(got, orc.particles, "KAT7 free particle step %d" % step, _arith(ieee))
        keys, perm, off = w.sort_view()
        assert np.array_equal(keys, orc.particle_cell_indicies) and np.array_equal(off, orc.cell_offsets)
    dt = np.float32(1.0) / np.float32(60.0)
    w2 = ws.FluidWorker(pos, params, ieee_division=ieee)
    w2.run()
    p = w2.read_vec("particles")[0]
    vy = np.float32(0) + (np.float32(-9.8) + np.float32(0)) * dt
    y = np.float32(2.0) + vy * dt
    assert p["velocity"][1] == vy and p["position"][1] == y and p["predicted_position"][1] == y + vy * np.float32(0.02)
    assert p["position"][0] == 1.0 and p["position"][2] == 3.0
    w.close()
    w2.close()


@BOTH_ARITHMETICS
def test_kat8_wall_hit_reflects_and_damps(oracle, ws, ieee):
    """KAT 8: a particle that would cross ext_min.y ends with y = -4.4 exactly and v.y multiplied by -0.95
    (assets/simulation.wgsl:292-298) -- on the GPU, bit for bit."""
    pos = np.float32([[0.0, -4.399, 0.0]])
    params = ws.default_params()
    orc = oracle_from_params(oracle, pos, params)
    state = orc.particles.copy()
    state["velocity"][0, 1] = -5.0
    state["predicted_position"][0, 1] = np.float32(-4.399) + np.float32(-5.0) * np.float32(0.02)
    orc.set_particles(state)
    orc.step(oracle.SORT_EXACT)
    w = ws.FluidWorker(pos, params, ieee_division=ieee)
    w.write_slice("particles", state)
    w.run()
    got = w.read_vec("particles")
    assert_particles_bitwise(got, orc.particles, "KAT8 wall hit", _arith(ieee))
    v_before = np.float32(-5.0) + np.float32(-9.8) * (np.float32(1) / np.float32(60))
    assert got["position"][0, 1] == np.float32(-4.4)
    assert got["velocity"][0, 1] == v_before * (np.float32(-1.0) * np.float32(0.95)) and got["velocity"][0, 1] > 0
    w.close()


def _sparse_cloud(ws, n_side, spacing, seed):
    """n_side^3 particles on a jittered lattice whose cells are `spacing` wide, with random velocities: nobody has a
    neighbour within the smoothing radius for the first steps.  Container = the lattice's extent + padding."""
    size = n_side * spacing + 0.2
    params = ws.make_params(container_size=(size, size, size))
    rng = np.random.default_rng(seed)
    g = (np.arange(n_side, dtype=np.float32) + np.float32(0.5)) * np.float32(spacing) - np.float32(n_side * spacing / 2)
    pos = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    pos = (pos + rng.uniform(-0.15, 0.15, pos.shape)).astype(np.float32)
    vel = rng.uniform(-2.0, 2.0, pos.shape).astype(np.float32)
    vel[:, 1] -= 2.0  # towards the floor: the bottom layer hits it within the run
    return pos, vel, params


@BOTH_ARITHMETICS
def test_cloud_without_neighbours_equals_the_oracle_bit_for_bit_through_wall_hits(oracle, ws, ieee):
    """4 096 particles far apart (listed kernels, power-of-two N without stencil aliasing), free-running next to the
    oracle: as long as no particle has a neighbour (checked on the oracle: density == the self term), position,
    velocity, predicted position, density, pressure and acceleration are equal to the oracle's bit for bit, and so
    are the three index buffers' comparable artefacts.  The run goes through wall reflections."""
    pos, vel, params = _sparse_cloud(ws, 16, 1.5, 2024)
    orc = oracle_from_params(oracle, pos, params)
    state = orc.particles.copy()
    state["velocity"][:, :3] = vel
    state["predicted_position"][:, :3] = state["position"][:, :3] + state["velocity"][:, :3] * np.float32(0.02)
    orc.set_particles(state)
    w = ws.FluidWorker(pos, params, ieee_division=ieee)
    w.write_slice("particles", state)
    self_density = None
    wall_hits = 0
    steps_compared = 0
    mn, mx = np.float32(list(params.ext_min)[:3]), np.float32(list(params.ext_max)[:3])
    for step in range(40):
        orc.step(oracle.SORT_FAST)
        if self_density is None:
            self_density = orc.particles["density"][0].copy()
        if not (np.all(orc.particles["density"] == self_density) and not orc.particles["acceleration"].any()):
            break  # somebody met a neighbour: from here on sums have an order
        w.run()
        got = w.read_vec("particles")
        assert_particles_bitwise(got, orc.particles, "no-neighbour cloud step %d" % step, _arith(ieee))
        keys, perm, off = w.sort_view()
        assert np.array_equal(keys, orc.particle_cell_indicies)
        assert np.array_equal(keys[perm], orc.sorted_keys()) and np.array_equal(off, orc.cell_offsets)
        wall_hits += int(np.sum((got["position"][:, :3] == mn) | (got["position"][:, :3] == mx)))
        steps_compared += 1
    w.close()
    assert steps_compared >= 12, "the cloud met neighbours after %d steps: make it sparser" % steps_compared
    assert wall_hits >= 100, "the run must go through wall reflections (%d seen)" % wall_hits


def test_k6_integrate_is_bit_exact_for_particles_without_neighbours_inside_a_dense_run(oracle, ws):
    """The same statement inside a REAL workload: in one teacher-forced C2 step every particle the oracle finds
    without a neighbour (density == the self term, zero acceleration) comes out of the product with position, velocity
    and predicted position equal to the oracle's bit for bit (K6 has no reduction: simulation.wgsl:271-310)."""
    from util import oracle_one_step

    pos, params = ws.workloads.make_workload("c2", "cloud")
    orc = oracle_from_params(oracle, pos, params)
    state = orc.particles.copy()
    want = oracle_one_step(oracle, orc, state, mode=oracle.SORT_FAST)
    w = ws.FluidWorker(pos, params)
    w.write_slice("particles", state)
    w.run()
    got = w.read_vec("particles")
    w.close()
    k = ws.get_smoothing_kernel(params)
    h = np.float32(params.smoothing_radius)
    self_rho = h * h * np.float32(k.pow2) + np.float32(1e-5)
    lonely = np.flatnonzero((want["density"][:, 0] == self_rho) & ~want["acceleration"].any(axis=1))
    assert lonely.size > 1000, lonely.size
    for f in ("position", "velocity", "predicted_position", "density", "pressure", "acceleration"):
        assert np.array_equal(got[f][lonely].view(np.uint32), want[f][lonely].view(np.uint32)), f
