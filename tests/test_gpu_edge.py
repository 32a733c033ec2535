"""GPU edge cases and size-independent properties, through the C ABI."""
import numpy as np
import pytest

from util import assert_particles_close, oracle_from_params, oracle_one_step, reorder_noise_tolerances

pytestmark = pytest.mark.gpu


def _one_step_parity(oracle, ws, pos, params, mode=None, state=None, label=""):
    orc = oracle_from_params(oracle, pos, params)
    st = orc.particles.copy() if state is None else state
    want = oracle_one_step(oracle, orc, st, mode=mode)
    keys_want, sorted_want, off_want = orc.particle_cell_indicies.copy(), orc.sorted_keys().copy(), orc.cell_offsets.copy()
    rev = oracle_one_step(oracle, orc, st, reverse=True, mode=mode)
    w = ws.FluidWorker(pos, params)
    w.write_slice("particles", st)
    w.run()
    got = w.read_vec("particles")
    keys, perm, off = w.sort_view()
    assert np.array_equal(keys, keys_want)
    assert np.array_equal(keys[perm], sorted_want)
    assert np.array_equal(off, off_want)
    assert_particles_close(got, want, reorder_noise_tolerances(want, rev), label)
    w.close()
    return got


@pytest.mark.parametrize("n", [1, 2, 8, 64, 512])
def test_tiny_n_hash_aliasing(oracle, ws, n):
    """For tiny N several cells of one 27-stencil share a bucket of the reference's N-entry table and
    the reference counts those neighbours more than once; the HIP path reproduces the multiplicity."""
    side = {1: (1, 1, 1), 2: (2, 1, 1), 8: (2, 2, 2), 64: (4, 4, 4), 512: (8, 8, 8)}[n]
    pos = ws.cube_fluid(*side)
    got = _one_step_parity(oracle, ws, pos, ws.make_params(container_size=(4.0, 4.0, 4.0)), label="n=%d" % n)
    if n == 1:
        k = ws.get_smoothing_kernel(ws.default_params())
        assert got["density"][0, 0] == pytest.approx(27 * 0.0625 * k.pow2 + 1e-5, rel=1e-6)


@pytest.mark.parametrize("n", [1000, 3000, 15823, 31646])
def test_non_power_of_two_n(oracle, ws, n):
    """The reference's bitonic sort needs a power of two (src/fluid_compute.rs:15 FIXME); the counting
    sort here does not.  Oracle in fast-sort mode defines the expected values.  15 823 is the x prime of
    hash_cell (and 31 646 twice it): there the x-neighbour cells of every cell share one bucket of the
    reference's table and the reference counts such neighbours three times; the HIP path reproduces it
    (per-pair multiplicity for a non-power-of-two N whose stencil can alias)."""
    mn, mx = ws.get_ext((0, 0, 0), (6.0, 4.0, 4.0), 0.1)
    pos = ws.workloads.uniform_cloud(n, 99, mn, mx)
    _one_step_parity(oracle, ws, pos, ws.make_params(container_size=(6.0, 4.0, 4.0)), mode=oracle.SORT_FAST,
                     label="n=%d" % n)


@pytest.mark.parametrize("size", [(2.7, 2.7, 2.7), (2.7, 2.2, 1.7)])
def test_grids_whose_cell_count_is_not_a_multiple_of_four(oracle, ws, size):
    """The cell scan moves 16 B per lane; the last cells of such a grid go through its scalar tail."""
    params = ws.make_params(container_size=size)
    pos = ws.workloads.uniform_cloud(4096, 5, list(params.ext_min), list(params.ext_max))
    w = ws.FluidWorker(pos, params)
    dims = w.grid_dims()
    w.close()
    assert (dims[0] * dims[1] * dims[2]) % 4 != 0, dims
    _one_step_parity(oracle, ws, pos, params, label="grid %s" % (dims,))


def test_particles_far_outside_the_grid_are_clamped_not_lost(oracle, ws):
    """Predicted positions may leave the container (position is clamped, position + v/50 is not): cells
    outside the padded dense grid are clamped; the exact distance test keeps the neighbour set."""
    params = ws.make_params(container_size=(4.0, 4.0, 4.0))
    pos = ws.cube_fluid(8, 8, 8)
    orc = oracle_from_params(oracle, pos, params)
    st = orc.particles.copy()
    rng = np.random.default_rng(5)
    fast = rng.choice(512, 64, replace=False)
    st["velocity"][fast, :3] = rng.uniform(-400, 400, (64, 3)).astype(np.float32)
    st["predicted_position"][:, :3] = st["position"][:, :3] + st["velocity"][:, :3] * np.float32(0.02)
    # a tight clump outside the grid: they must still see each other
    st["predicted_position"][fast[:8], :3] = np.float32([30.0, 30.0, 30.0]) + rng.uniform(0, 0.1, (8, 3)).astype(np.float32)
    _one_step_parity(oracle, ws, pos, params, state=st, label="escaped")


def test_set_params_gravity_and_radius_change(oracle, ws):
    """update() re-uploads fluid_props / smoothing_kernel / gravity every frame (fluid_compute.rs:479-481);
    a new smoothing radius changes the cell size and forces a re-grid."""
    pos = ws.cube_fluid(16, 8, 8)
    params = ws.make_params(container_size=(6.0, 4.0, 4.0))
    w = ws.FluidWorker(pos, params)
    w.run(3)
    state = w.read_vec("particles")
    p2 = ws.make_params(container_size=(6.0, 4.0, 4.0), gravity=(0.0, 0.0, 0.0, 0.0), smoothing_radius=0.35,
                        viscosity_strength=0.2)
    w.set_params(p2)
    w.run()
    got = w.read_vec("particles")
    orc = oracle_from_params(oracle, pos, p2)
    want = oracle_one_step(oracle, orc, state.astype(oracle.PARTICLE_DTYPE))
    rev = oracle_one_step(oracle, orc, state.astype(oracle.PARTICLE_DTYPE), reverse=True)
    assert_particles_close(got, want, reorder_noise_tolerances(want, rev), "after set_params")
    keys, perm, off = w.sort_view()
    assert np.array_equal(keys, orc.particle_cell_indicies)
    w.close()


def test_record_view_reports_the_last_steps_acceleration_whatever_happens_after_it(ws):
    """The step does not store accelerations; ws_read_particles computes them on demand from the state the last step
    left behind.  They must be the ones that step USED: reading twice gives the same bits, and a parameter change
    (viscosity, pressure; a new radius that throws the grid away) between the step and the read does not leak in."""
    pos = ws.workloads.uniform_cloud(20000, 3, [-2.9, -1.9, -1.9], [2.9, 1.9, 1.9])
    params = ws.make_params(container_size=(6.0, 4.0, 4.0))

    def after(change):
        w = ws.FluidWorker(pos, params)
        w.run(5)
        if change is not None:
            w.set_params(change)
        out = w.read_vec("particles")
        again = w.read_vec("particles")
        w.close()
        for f in out.dtype.names:
            assert np.array_equal(out[f].view(np.uint32), again[f].view(np.uint32)), f
        return out

    want = after(None)
    assert np.abs(want["acceleration"][:, :3]).max() > 0
    for change in (ws.make_params(container_size=(6.0, 4.0, 4.0), viscosity_strength=0.4, pressure_scalar=50.0),
                   ws.make_params(container_size=(6.0, 4.0, 4.0), smoothing_radius=0.35),
                   ws.make_params(container_size=(6.0, 4.0, 4.0), gravity=(1.0, 2.0, 3.0, 0.0))):
        got = after(change)
        for f in ("acceleration", "position", "velocity", "predicted_position"):
            assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f


def test_reset_restores_initial_state_and_identity_views(ws):
    pos = ws.cube_fluid(8, 8, 8)
    w = ws.FluidWorker(pos, ws.make_params(container_size=(4.0, 4.0, 4.0)))
    keys, perm, off = w.sort_view()  # before the first step: all identity (fluid_compute.rs:306-308)
    ident = np.arange(512, dtype=np.uint32)
    assert np.array_equal(keys, ident) and np.array_equal(perm, ident) and np.array_equal(off, ident)
    first = None
    for _ in range(2):
        w.run(4)
        out = w.read_vec("particles")
        if first is None:
            first = out
        else:
            for f in out.dtype.names:
                assert np.array_equal(out[f].view(np.uint32), first[f].view(np.uint32)), f  # deterministic
        w.reset(pos)
        assert w.steps_done() == 0
        p0 = w.read_vec("particles")
        assert np.array_equal(p0["position"][:, :3], pos) and np.array_equal(p0["predicted_position"][:, :3], pos)
        assert not p0["velocity"].any() and not p0["density"].any() and not p0["acceleration"].any()
    w.close()


def test_step_is_asynchronous_and_ready_polls(ws):
    """`ws_step` only enqueues (AppComputeWorker::run) and `ws_ready` polls without blocking (::ready,
    src/fluid_compute.rs:474): with 40 C3 steps queued (>= 20 ms of GPU work, enqueued in ~2 ms) the handle must
    report busy, the enqueue must return long before the work is done, and after ws_sync it must report ready."""
    import time

    pos, params = ws.workloads.make_workload("c3", "cloud")
    w = ws.FluidWorker(pos, params)
    w.run(2)
    w.sync()
    assert w.ready()
    t0 = time.perf_counter()
    w.run(40)
    t_enqueued = time.perf_counter() - t0
    busy = not w.ready()
    w.sync()
    t_done = time.perf_counter() - t0
    assert busy, "ws_ready reported 1 with 40 steps of 4 194 304 particles still queued"
    assert t_enqueued < 0.5 * t_done, "ws_step blocked: enqueue %.1f ms of %.1f ms" % (t_enqueued * 1e3, t_done * 1e3)
    assert w.ready()
    assert w.steps_done() == 42
    w.close()


def test_full_size_properties_c3(ws):
    """BASELINE.json config 3 (4 194 304 particles): properties that need no oracle at this size."""
    pos, params = ws.workloads.make_workload("c3", "cloud")
    n = pos.shape[0]
    w = ws.FluidWorker(pos, params)
    w.run(3)
    keys, perm, off = w.sort_view()
    assert np.array_equal(np.sort(perm), np.arange(n, dtype=np.uint32))      # ids conserved
    sk = keys[perm]
    assert np.all(sk[1:] >= sk[:-1])                                          # sortedness
    heads = np.flatnonzero(np.r_[True, sk[1:] != sk[:-1]])
    first = np.full(n, 999999999, np.uint32)
    first[sk[heads]] = heads.astype(np.uint32)
    assert np.array_equal(off, first)                                         # offsets = first slot or INF
    P = w.read_vec("particles")
    mn = np.float32(list(params.ext_min)[:3]); mx = np.float32(list(params.ext_max)[:3])
    assert np.all(P["position"][:, :3] >= mn) and np.all(P["position"][:, :3] <= mx)  # wall clamp
    assert np.array_equal(P["predicted_position"][:, :3], P["position"][:, :3] + P["velocity"][:, :3] * np.float32(0.02))
    assert np.all(np.isfinite(P["density"])) and np.all(P["density"][:, 0] > 152.0)   # self term alone is 152.79
    assert np.array_equal(P["pressure"][:, 0], np.float32(22.0) * (P["density"][:, 0] - np.float32(10.0)))
    for f in ("position", "velocity", "acceleration", "predicted_position"):
        assert not P[f][:, 3].any()
    # keys are the hash of the cell of the predicted position the step started from: recompute on the host
    w2 = ws.FluidWorker(pos, params)
    w2.run(2)
    Q = w2.read_vec("particles")["predicted_position"][:, :3]
    cell = np.floor(Q / np.float32(params.smoothing_radius)).astype(np.int64).astype(np.uint32) if False else \
        np.floor(Q / np.float32(params.smoothing_radius)).astype(np.int32).view(np.uint32)
    h = (cell[:, 0] * np.uint32(15823) + cell[:, 1] * np.uint32(9737333) + cell[:, 2] * np.uint32(440817757)) % np.uint32(n)
    assert np.array_equal(keys, h)
    w.close(); w2.close()


def test_full_size_density_and_force_spot_check_c3(ws):
    """BASELINE.json config 3: the densities (and, for half of them, the accelerations) of 256 particles against
    brute-force sums over ALL 4 194 304 particles (float64 numpy, no cell grid, no oracle): finds a lost or
    double-counted neighbour at full size."""
    pos, params = ws.workloads.make_workload("c3", "cloud")
    w = ws.FluidWorker(pos, params)
    w.run(40)                      # into the collapse: the floor layer is already dense
    before = w.read_vec("particles")
    w.run(1)
    after = w.read_vec("particles")
    w.close()
    q = before["predicted_position"][:, :3].astype(np.float64)   # the step's densities are sums over these
    h = float(params.smoothing_radius)
    k2 = 15.0 / (2.0 * np.pi * h ** 5)
    k3 = 15.0 / (np.pi * h ** 6)
    rng = np.random.default_rng(11)
    pick = np.r_[rng.integers(0, q.shape[0], 192), np.argsort(-after["density"][:, 0])[:64]]  # random + the densest
    for i in pick:
        box = np.flatnonzero(np.all(np.abs(q - q[i]) <= h, axis=1))
        dist = np.sqrt(((q[box] - q[i]) ** 2).sum(axis=1))
        dist = dist[dist <= h]
        rho = ((h - dist) ** 2).sum() * k2 + 1e-5
        rho_near = ((h - dist) ** 3).sum() * k3 + 1e-5
        assert abs(after["density"][i, 0] - rho) <= 2e-5 * rho + 1e-3, (i, after["density"][i, 0], rho, dist.size)
        assert abs(after["density"][i, 1] - rho_near) <= 2e-5 * rho_near + 1e-3, (i, after["density"][i, 1], rho_near)
    # and their accelerations (simulation.wgsl:197-269 restated in float64 over the brute-force neighbour set,
    # with the step's own densities as inputs); the bound is relative to the sum of the term magnitudes
    rho_all = after["density"].astype(np.float64)
    vel = before["velocity"][:, :3].astype(np.float64)
    ps, nps, tgt, visc = (float(params.pressure_scalar), float(params.near_pressure_scalar), float(params.target_density),
                          float(params.viscosity_strength))
    k2d, k3d, ksp = 15.0 / (np.pi * h ** 5), 45.0 / (np.pi * h ** 6), 315.0 / (64.0 * np.pi * h ** 9)
    for i in pick[::2]:
        box = np.flatnonzero(np.all(np.abs(q - q[i]) <= h, axis=1))
        box = box[box != i]
        off = q[box] - q[i]
        dist = np.sqrt((off ** 2).sum(axis=1))
        keep = dist <= h
        box, off, dist = box[keep], off[keep], dist[keep]
        direction = np.where(dist[:, None] > 0, off / np.maximum(dist, 1e-300)[:, None], np.array([0.0, 1.0, 0.0]))
        p_i, pn_i = ps * (rho_all[i, 0] - tgt), nps * rho_all[i, 1]
        p_j, pn_j = ps * (rho_all[box, 0] - tgt), nps * rho_all[box, 1]
        press = direction * ((p_i + p_j) / 2 * (dist - h) * k2d / rho_all[box, 0])[:, None]
        near = direction * ((pn_i + pn_j) / 2 * (dist - h) ** 2 * k3d / rho_all[box, 1])[:, None]
        visco = (vel[box] - vel[i]) * ((h * h - dist * dist) ** 3 * ksp)[:, None]
        acc = (press + near).sum(axis=0) / rho_all[i, 0] + visco.sum(axis=0) * visc
        scale = (np.abs(press) + np.abs(near)).sum(axis=0) / rho_all[i, 0] + np.abs(visco).sum(axis=0) * visc
        err = np.abs(after["acceleration"][i, :3] - acc)
        assert np.all(err <= 3e-5 * scale + 1e-4), (i, after["acceleration"][i, :3], acc, scale, box.size)


def test_full_size_two_slabs_match_the_single_handle_c3(ws):
    """BASELINE.json config 3 cut into two x-slabs (loopback transport, one GPU): bit-identical to the single handle."""
    pos, params = ws.workloads.make_workload("c3", "cloud")
    w = ws.FluidWorker(pos, params)
    w.run(4)
    want = w.read_vec("particles")
    w.close()
    got, owned = ws.slab.run_loopback(pos, params, 2, 4)
    assert sum(owned) == pos.shape[0]
    for f in want.dtype.names:
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f


def test_asynchronous_readback_captures_the_state_at_begin(ws):
    """update() of frame f draws the step submitted in frame f-1 while frame f's step already runs: `begin` must
    capture the positions as of the steps enqueued before it, whatever is enqueued afterwards."""
    pos, params = ws.workloads.make_workload("c2", "cloud")
    w = ws.FluidWorker(pos, params)
    buf = np.empty((pos.shape[0], 3), np.float32)
    w.pin_host_buffer(buf)
    try:
        w.run(3)
        want = w.read_positions()
        w.read_positions_begin(buf)
        with pytest.raises(ws.WsError):
            w.read_positions_begin(buf)          # one in flight
        w.run(5)                                 # overlaps with the copy
        w.read_positions_end()
        assert np.array_equal(buf, want)
        assert not np.array_equal(w.read_positions(), want)
        with pytest.raises(ws.WsError):
            w.read_positions_end()               # nothing in flight
    finally:
        w.unpin_host_buffer(buf)
    w.close()


def test_asynchronous_readback_into_the_librarys_own_buffers(ws):
    """ws_read_positions_begin(h, NULL): the copy goes into one of two page-locked buffers the library owns; the view
    of frame k stays untouched while frame k + 1's copy is in flight (what update() scatters into its Transforms)."""
    pos, params = ws.workloads.make_workload("c2", "cloud")
    w = ws.FluidWorker(pos, params)
    with pytest.raises(ws.WsError):
        w.read_positions_view()                  # nothing read yet
    frames = []
    for f in range(4):
        w.run(2)
        want = w.read_positions()
        w.read_positions_begin_owned()
        w.run(3)                                 # overlaps with the copy
        if frames:                               # the previous frame's view is still intact while this copy runs
            assert np.array_equal(frames[-1][0], frames[-1][1])
        w.read_positions_end()
        view = w.read_positions_view()
        assert np.array_equal(view, want)
        frames.append((view, want))
    assert frames[-1][0].ctypes.data != frames[-2][0].ctypes.data  # two buffers, alternately
    assert frames[-1][0].ctypes.data == frames[-3][0].ctypes.data
    # a caller buffer in between does not disturb the owned pair; the runtime's copy path gives the same bytes
    buf = np.empty((pos.shape[0], 3), np.float32)
    want = w.read_positions()
    w.read_positions_begin(buf)                  # pageable: hipMemcpyAsync
    w.read_positions_end()
    assert np.array_equal(buf, want)
    assert np.array_equal(w.read_positions_view(), frames[-1][1])
    w.close()


def test_readback_into_an_explicitly_pinned_buffer(ws):
    """update() reads into the same host buffer every frame; the host may page-lock it (PCIe-rate copy).
    The data must not depend on the copy path."""
    pos = ws.cube_fluid(32, 16, 16)
    w = ws.FluidWorker(pos, ws.make_params(container_size=(10.0, 6.0, 6.0)))
    buf = np.empty((pos.shape[0], 3), np.float32)
    w.pin_host_buffer(buf)
    try:
        for step in range(4):
            w.run()
            w.read_positions_into(buf)
            assert np.array_equal(buf, w.read_vec("particles")["position"][:, :3])
    finally:
        w.unpin_host_buffer(buf)
    v = w.read_vec("particles")["velocity"][:, :3]
    assert np.array_equal(w.read_speeds(), np.sqrt(v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2]))
    other = np.empty_like(buf)
    w.read_positions_into(other)  # a pageable buffer still works
    assert np.array_equal(other, buf)
    w.close()
