"""Known-answer tests that pin the CPU oracle.  The reference ships no tests or
golden vectors, so every value here is derived from the reference's SOURCE TEXT (cited), not from
running it: parity of the oracle itself is "unpinned by the reference"."""
import struct

import numpy as np
import pytest


def f32_hex(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


def test_kernel_constants_at_default_radius(oracle):
    # src/fluid_compute.rs:55-63 with h = 0.25 (powi exact): values + bit patterns, SURVEY.md a4
    k = oracle.smoothing_kernel(oracle.default_props())
    assert f32_hex(k.pow2) == 0x4518C9EB
    assert f32_hex(k.pow2_der) == 0x4598C9EB
    assert f32_hex(k.pow3) == 0x4698C9EB
    assert f32_hex(k.pow3_der) == 0x47652EE0
    assert f32_hex(k.spikey_pow3) == 0x48C88904
    assert k.pow2 == pytest.approx(15.0 / (2 * np.pi * 0.25**5), rel=1e-6)
    assert k.spikey_pow3 == pytest.approx(315.0 / (64 * np.pi * 0.25**9), rel=1e-6)


def test_default_props(oracle):
    p = oracle.default_props()  # src/fluid_compute.rs:20-27
    assert p.delta_time == np.float32(1.0) / np.float32(60.0)
    assert (p.collision_damping, p.smoothing_radius, p.target_density) == (np.float32(0.95), 0.25, 10.0)
    assert (p.pressure_scalar, p.near_pressure_scalar, p.viscosity_strength) == (22.0, 2.0, np.float32(0.1))


@pytest.mark.parametrize(
    "n,cell,want",
    [  # assets/simulation.wgsl:125-128 with the primes of :38-40; u32 wrap, then % N
        (65536, (0, 0, 0), 0), (65536, (1, 0, 0), 15823), (65536, (0, 1, 0), 38005), (65536, (0, 0, 1), 22621),
        (65536, (-1, -1, -1), 54623), (4194304, (0, 1, 0), 1348725), (4194304, (0, 0, 1), 415837),
        (4194304, (-1, -1, -1), 2413919),
    ],
)
def test_hash_cell(oracle, n, cell, want):
    assert oracle.hash_cell(cell, n) == want
    x, y, z = (c & 0xFFFFFFFF for c in cell)
    assert want == ((x * 15823 + y * 9737333 + z * 440817757) & 0xFFFFFFFF) % n


def test_get_cell_floor_semantics(oracle):
    # assets/simulation.wgsl:121-123: floor(pos / h), negatives round down
    assert list(oracle.get_cell((0.0, 0.24, 0.25), 0.25)) == [0, 0, 1]
    assert list(oracle.get_cell((-0.01, -0.25, -0.26), 0.25)) == [-1, -1, -2]


@pytest.mark.parametrize("n,count", [(4096, 78), (65536, 136), (262144, 171), (4194304, 253), (1 << 24, 300), (1 << 26, 351)])
def test_bit_sorter_stage_counts(oracle, n, count):
    # src/fluid_compute.rs:251-273: k(k+1)/2 stages, k = log2(next pow2)
    assert len(oracle.bit_sorter_stages(n)) == count


def test_bit_sorter_stage_order(oracle):
    assert oracle.bit_sorter_stages(8) == [(1, 2), (2, 4), (1, 4), (4, 8), (2, 8), (1, 8)]
    assert len(oracle.bit_sorter_stages(5)) == 6  # checked_next_power_of_two(5) = 8


def test_cube_fluid(oracle):
    # src/helpers.rs:3-20 with the reference's (64,32,32,0.1): id = (i*32 + j)*32 + k, centred
    p = oracle.cube_fluid(64, 32, 32, 0.1)
    assert p.shape == (65536, 3)
    np.testing.assert_allclose(p[0], (-6.3, -3.1, -3.1), atol=1e-6)
    np.testing.assert_allclose(p[-1], (6.3, 3.1, 3.1), atol=1e-6)
    i, j, k = 5, 7, 9
    d = np.float32(0.1) * np.float32(2)
    off = np.float32(0.1) - np.float32([64, 32, 32]) * np.float32(0.1)
    want = np.float32([i, j, k]) * d + off  # (i as f32) * diam, then + offset: same op order
    assert np.array_equal(p[(i * 32 + j) * 32 + k], want)


def test_get_ext_default_container(oracle):
    # src/fluid_container.rs:8,42-50 with padding = PARTICLE_RADIUS
    mn, mx = oracle.get_ext((0, 0, 0), (16, 9, 9), 0.1)
    assert np.array_equal(mn, np.float32([-8 + 0.1, -4.5 + 0.1, -4.5 + 0.1, 0]))
    assert np.array_equal(mx, np.float32([8 - 0.1, 4.5 - 0.1, 4.5 - 0.1, 0]))
    # the planar config-1 container pins z to exactly 0
    mn, mx = oracle.get_ext((0, 0, 0), (16, 18, 0.2), 0.1)
    assert mn[2] == 0.0 and mx[2] == 0.0


def test_initial_lattice_density_and_symmetry(oracle):
    """t = 0, interior lattice particle (spacing 0.2, h = 0.25): neighbours within h are self + 6
    face neighbours at 0.2 (next shell 0.2828 > h) -> density = W(0) + 6 W(0.2) + 1e-5."""
    pos = oracle.cube_fluid(16, 16, 16, 0.1)
    o = oracle.Oracle(pos)
    o.hash_particles(); o.sort(oracle.SORT_EXACT); o.calculate_cell_offsets(); o.update_density()
    o.update_pressure_force()
    k = o.st.kernel
    h = 0.25
    rho = h * h * k.pow2 + 6 * (h - 0.2) ** 2 * k.pow2
    rho_n = h**3 * k.pow3 + 6 * (h - 0.2) ** 3 * k.pow3
    i = (8 * 16 + 8) * 16 + 8
    assert o.particles["density"][i, 0] == pytest.approx(rho, rel=2e-6)      # ~189.458
    assert o.particles["density"][i, 1] == pytest.approx(rho_n, rel=2e-6)    # ~320.245
    assert o.particles["pressure"][i, 0] == pytest.approx(22 * (rho - 10), rel=2e-6)  # ~3948.08
    assert o.particles["pressure"][i, 1] == pytest.approx(2 * rho_n, rel=2e-6)        # ~640.49
    # acceleration ~ 0 by symmetry for interior particles (heavy cancellation: absolute bound)
    assert np.max(np.abs(o.particles["acceleration"][i])) < 1e-3


def test_single_free_particle_gravity_only(oracle):
    # assets/simulation.wgsl:279-309: v += g dt; x += v dt; pred = x + v / 50
    o = oracle.Oracle(np.float32([[1.0, 2.0, 3.0]]))
    o.step(oracle.SORT_EXACT)
    p = o.particles[0]
    dt = np.float32(1.0) / np.float32(60.0)
    vy = np.float32(0) + (np.float32(-9.8) + np.float32(0)) * dt
    y = np.float32(2.0) + vy * dt
    assert p["velocity"][1] == vy and p["position"][1] == y
    assert p["predicted_position"][1] == y + vy * np.float32(0.02)
    assert p["position"][0] == 1.0 and p["position"][2] == 3.0
    # N = 1: all 27 stencil cells hash to bucket 0 (% num_particles), so the reference's walk
    # meets the particle 27 times -- the hashed table's aliasing is part of its semantics
    assert p["density"][0] == pytest.approx(27 * 0.25**2 * o.st.kernel.pow2 + 1e-5, rel=1e-6)


def test_wall_hit_reflects_and_damps(oracle):
    # assets/simulation.wgsl:292-298: y < ext_min.y -> v.y *= -0.95, y = ext_min.y
    o = oracle.Oracle(np.float32([[0.0, -4.399, 0.0]]))
    o.particles["velocity"][0, 1] = -5.0
    o.step(oracle.SORT_EXACT)
    p = o.particles[0]
    v_before = np.float32(-5.0) + np.float32(-9.8) * (np.float32(1) / np.float32(60))
    assert p["position"][1] == np.float32(-4.4)
    assert p["velocity"][1] == v_before * (np.float32(-1.0) * np.float32(0.95))
    assert p["velocity"][1] > 0


def test_w_components_stay_zero(oracle):
    o = oracle.Oracle(oracle.cube_fluid(8, 8, 8, 0.1))
    for _ in range(3):
        o.step(oracle.SORT_EXACT)
    for f in ("position", "velocity", "acceleration", "predicted_position"):
        assert not np.any(o.particles[f][:, 3])
