"""The host's whole per-frame boundary on slab (multi-GPU) handles, on ONE GPU through the loopback transport.

The reference's host needs, every frame, id-ordered positions of ALL particles (update(), src/fluid_compute.rs:478-485),
pushes parameters (:479-481; the HUD can change the smoothing radius, src/hud.rs:135-138, which changes the cell grid)
and resets on Space (despawn_liquid, :505-525).  On slab handles these are collective calls over global, id-ordered
arrays; the frame loop below -- written once against the worker interface -- must give bit-identical results on one
handle and on 2 / 3 / 4 slabs, through a reset and two radius changes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frame_loop(ws, pos, params, frames, events):
    """update -> despawn -> run, in the order of src/schedule.rs:24-36.  events: {frame: ("reset",) | ("params", p)}."""

    def program(w, rank=0):
        seen = {}
        current = params
        for f in range(frames):
            ev = events.get(f)
            if ev and ev[0] == "reset":            # Update / DespawnEntities: despawn_liquid
                w.reset(pos)
            if ev and ev[0] == "params":           # Update / UserInput: the HUD edits FluidStaticProps / Gravity
                current = ev[1]
            xyz = w.read_positions()               # Update / EntityUpdates: update() reads the positions ...
            w.set_params(current)                  # ... and pushes fluid_props / smoothing_kernel / gravity
            if f % 5 == 0 or f == frames - 1:
                seen[f] = xyz
            w.run(1)                               # PostUpdate / Pass
        return seen, w.read_vec("particles"), w.sort_view(), w.read_speeds()

    return program


def _events(ws, size):
    return {
        6: ("params", ws.make_params(container_size=size, gravity=(6.0, -9.8, 0.0, 0.0), smoothing_radius=0.35)),
        14: ("reset",),
        20: ("params", ws.make_params(container_size=size, gravity=(-4.0, -9.8, 1.0, 0.0), smoothing_radius=0.15,
                                      viscosity_strength=0.2)),
    }


@pytest.mark.parametrize("world", [2, 3, 4])
def test_frame_loop_with_reset_and_radius_changes_is_bit_identical_on_slabs(ws, world):
    size = (16.0, 9.0, 9.0)
    params = ws.make_params(container_size=size, gravity=(6.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(40000, 808, list(params.ext_min), list(params.ext_max))  # not a power of two
    frames = 30
    program = _frame_loop(ws, pos, params, frames, _events(ws, size))
    single = ws.FluidWorker(pos, params)
    want_seen, want_rec, want_view, want_speed = program(single)
    single.close()
    results = ws.slab.run_loopback_program(pos, params, world, program)
    for r, (seen, rec, view, speed) in enumerate(results):   # every rank gets the whole, id-ordered arrays
        for f in want_seen:
            assert np.array_equal(seen[f].view(np.uint32), want_seen[f].view(np.uint32)), "positions of frame %d, rank %d" % (f, r)
        for name in want_rec.dtype.names:
            assert np.array_equal(rec[name].view(np.uint32), want_rec[name].view(np.uint32)), (name, r)
        for a, b in zip(view, want_view):                    # particle_cell_indicies / particle_indicies / cell_offsets
            assert np.array_equal(a, b)
        assert np.array_equal(speed.view(np.uint32), want_speed.view(np.uint32))


def test_slab_handles_take_and_return_global_records(ws):
    """ws_write_particles / ws_read_particles on slab handles: every rank is handed ALL records (id order) and keeps
    what its cuts own -- by PREDICTED position, which is what the step bins by --, and every rank reads ALL records
    back.  Round trip without a step is the identity on position / velocity / predicted position; stepping from the
    written state equals the single handle."""
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(32768, 31, list(params.ext_min), list(params.ext_max))
    w = ws.FluidWorker(pos, params)
    w.run(10)
    state = w.read_vec("particles")
    w.write_slice("particles", state)
    w.run(5)
    want = w.read_vec("particles")
    w.close()

    def program(s, rank):
        s.write_particles(state)
        back = s.read_vec("particles")
        owned = s.num_owned()
        s.run(5)
        return back, s.read_vec("particles"), owned

    results = ws.slab.run_loopback_program(pos, params, 3, program)
    assert sum(r[2] for r in results) == pos.shape[0]
    for back, got, _ in results:
        for f in ("position", "velocity", "predicted_position"):
            assert np.array_equal(back[f].view(np.uint32), state[f].view(np.uint32)), f
        assert not back["density"].any() and not back["acceleration"].any()   # no step on this set yet
        for f in want.dtype.names:
            assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f


def test_asynchronous_global_readback_on_slabs(ws):
    """ws_read_positions_begin / _end on slab handles: the positions as of the steps enqueued before `begin`,
    whatever is enqueued afterwards."""
    params = ws.make_params(container_size=(16.0, 9.0, 9.0))
    pos = ws.workloads.uniform_cloud(20000, 4, list(params.ext_min), list(params.ext_max))

    def program(s, rank):
        s.run(3)
        want = s.read_positions()
        buf = np.empty((pos.shape[0], 3), np.float32)
        s.read_positions_begin(buf)
        s.run(4)
        s.read_positions_end()
        later = s.read_positions()
        s.read_positions_begin_owned()       # the same into the library's own page-locked double buffer
        s.run(2)
        s.read_positions_end()
        return want, buf, later, s.read_positions_view().copy()

    w = ws.FluidWorker(pos, params)
    w.run(3)
    single3 = w.read_positions()
    w.run(4)
    single7 = w.read_positions()
    w.close()
    for want, buf, later, owned in ws.slab.run_loopback_program(pos, params, 2, program):
        assert np.array_equal(want, single3) and np.array_equal(buf, single3) and np.array_equal(later, single7)
        assert np.array_equal(owned, single7)


def test_capacity_overrun_with_a_sync_after_every_step_still_fails_every_rank_at_the_same_step(ws):
    """The frame pattern is step-then-read: a rank that learns of ITS overrun in ws_sync (one or two steps before the
    others can) must keep stepping -- and issuing the step's collectives -- until ws_step itself fails, on every rank at
    the same step.  (ws_sync used to latch the local bit and the rank then left its peers alone in the next collective.)"""
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(65536, 1234, list(params.ext_min), list(params.ext_max))
    world = 3
    errs = ws.slab.run_loopback(pos, params, world, 80, collect_errors=True, sync_every_step=True,
                                capacity=65536 // world + 256)
    step_errs = {k: v for k, v in errs.items() if not isinstance(k, tuple)}
    assert sorted(step_errs) == list(range(world)), "every rank must fail in ws_step: %r" % (errs,)
    assert len({k for k, _ in step_errs.values()}) == 1, "all ranks must fail at the same step: %r" % (errs,)
    early = [k for k in errs if isinstance(k, tuple)]
    assert early, "some rank should have heard of the overrun from ws_sync before ws_step failed"
    for k, e in step_errs.values():
        assert e.status == 3 and "capacity" in str(e), e


@pytest.mark.parametrize("world", [2, 3])
def test_particles_that_overshoot_the_padding_with_halo_overlap_on(ws, world):
    """Predicted positions that leave the two cells of padding are clamped into the grid's border rows, whose
    linearised stencil wraps into the neighbouring LAYER -- for the first / last early layer of a slab a ghost layer,
    whose cell starts the halo stream is rewriting while the early range computes.  Early launches cut their runs to
    the owned range, so K4 and K5 number their candidates alike whatever the ghost starts hold: bit-identical to the
    single handle."""
    params = ws.make_params(container_size=(16.0, 9.0, 9.0))
    pos = ws.workloads.uniform_cloud(65536, 77, list(params.ext_min), list(params.ext_max))
    orc_like = np.zeros(pos.shape[0], ws.PARTICLE_DTYPE)
    orc_like["position"][:, :3] = pos
    rng = np.random.default_rng(3)
    fast = rng.choice(pos.shape[0], 6000, replace=False)   # v / 50 far beyond the 0.5 units of padding, every direction in y, z
    orc_like["velocity"][fast, 1:3] = rng.choice([-300.0, 300.0], (fast.size, 2)).astype(np.float32)
    orc_like["predicted_position"][:, :3] = orc_like["position"][:, :3] + orc_like["velocity"][:, :3] * np.float32(0.02)
    w = ws.FluidWorker(pos, params)
    w.write_slice("particles", orc_like)
    w.run(6)
    want = w.read_vec("particles")
    w.close()
    got, owned = ws.slab.run_loopback(pos, params, world, 6, state=orc_like)
    assert sum(owned) == pos.shape[0]
    for f in want.dtype.names:
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f


def test_rebalance_moves_the_cuts_where_static_cuts_overrun(ws):
    """Equal-layer cuts are balanced only while the fluid is spread evenly along x.  With gravity tilted towards +x the
    last of three slabs fills up until its capacity overruns (ws_step fails on every rank); the same run with a
    collective ws_slab_rebalance every 8 steps keeps every slab near n / 3, never overruns, and still reproduces the single
    handle bit for bit."""
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(65536, 1234, list(params.ext_min), list(params.ext_max))
    world, steps, cap = 3, 96, 65536 // 3 + 6000
    errs = ws.slab.run_loopback(pos, params, world, steps, collect_errors=True, capacity=cap)
    assert sorted(errs) == list(range(world)), "static cuts were expected to overrun this capacity: %r" % (errs,)
    w = ws.FluidWorker(pos, params)
    w.run(steps)
    want = w.read_vec("particles")
    w.close()

    def program(s, rank):
        owned = []
        for k in range(steps // 8):
            s.rebalance()          # (the record view reports densities / accelerations of the last STEP: re-cut first)
            owned.append(s.num_owned())
            s.run(8)
        return s.read_vec("particles"), owned

    results = ws.slab.run_loopback_program(pos, params, world, program, capacity=cap)
    for r, (got, owned) in enumerate(results):
        assert max(owned) <= cap
        for f in want.dtype.names:
            assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), (f, r)
    final = [results[r][1][-1] for r in range(world)]
    assert sum(final) == pos.shape[0]
    assert max(final) - min(final) < 65536 // 3 // 2, "not balanced: %r" % (final,)


def test_a_rank_local_overrun_with_a_collective_read_after_every_frame_fails_every_rank_at_the_same_call(ws, monkeypatch):
    """A halo overrun is set by k_halo_pack on the rank it happens on and reaches the others only with the NEXT step's
    all-gather.  A host that reads positions after every step (update(), src/fluid_compute.rs:478) enters the collective
    read in between: the rank that knows must not leave it alone (its peers would wait in the all-gathers for ever) --
    the bits travel in the read's count words and every rank refuses the view at the same call."""
    monkeypatch.setenv("WS_LOOPBACK_TIMEOUT", "30")  # a rank left alone breaks the barrier after 30 s instead of 300
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(65536, 1234, list(params.ext_min), list(params.ext_max))
    world = 3

    def program(s, rank):
        for k in range(40):
            try:
                s.run(1)
            except ws.WsError as e:
                return ("step", k, e.status, str(e))
            try:
                s.read_positions()
            except ws.WsError as e:
                return ("read", k, e.status, str(e))
        return None

    res = ws.slab.run_loopback_program(pos, params, world, program, ghost_capacity=64)  # a boundary layer holds ~1000
    assert all(r is not None for r in res), res
    assert len({(r[0], r[1]) for r in res}) == 1, "every rank must fail at the same call: %r" % (res,)
    assert res[0][0] == "read", res          # the read right behind the step that overran, not a step later
    for r in res:
        assert r[2] == 3 and "halo" in r[3], r  # WS_ERR_OUT_OF_MEMORY, the halo message


def test_a_regrid_that_would_overrun_a_slab_is_refused_by_every_rank_before_anything_changes(ws):
    """h = 0.25 -> 1.0 on three slabs of a uniform cloud: equal-LAYER cuts of the coarse grid (20 layers, 16 of them
    inside the container) would give the middle slab 7 / 16 of the fluid, more than its capacity of 0.40 n.  Every rank
    knows every rank's new share (the gathered state) and capacity (the count words): all refuse alike, nothing has been
    touched, the run goes on bit-identically.  After a re-cut by particle count the same re-grid is accepted -- the
    equal-count rule is re-applied on the new grid -- and still reproduces the single handle; the migration counters
    are cumulative over all of it."""
    size = (16.0, 9.0, 9.0)
    params = ws.make_params(container_size=size, gravity=(2.0, -9.8, 0.0, 0.0))
    coarse = ws.make_params(container_size=size, gravity=(2.0, -9.8, 0.0, 0.0), smoothing_radius=1.0)
    pos = ws.workloads.uniform_cloud(32768, 4321, list(params.ext_min), list(params.ext_max))
    world, cap = 3, int(0.40 * 32768)
    w = ws.FluidWorker(pos, params)
    w.run(12)
    want_a = w.read_vec("particles")
    w.set_params(coarse)
    w.run(6)
    want_b = w.read_vec("particles")
    w.close()

    def program(s, rank):
        s.run(6)
        refused = None
        try:
            s.set_params(coarse)
        except ws.WsError as e:
            refused = (e.status, str(e))
        s.run(6)                                  # ... with the old radius: the handle is untouched
        a = s.read_vec("particles")
        before = s.counters()
        s.rebalance()
        s.set_params(coarse)                      # accepted now: equal counts on the new grid
        owned = s.num_owned()
        after = s.counters()
        s.run(6)
        return refused, a, s.read_vec("particles"), owned, before, after

    res = ws.slab.run_loopback_program(pos, params, world, program, capacity=cap)
    for r, (refused, a, b, owned, before, after) in enumerate(res):
        assert refused is not None and refused[0] == 3 and "would own" in refused[1] and "nothing was changed" in refused[1], refused
        assert owned <= cap
        assert after["left"] >= before["left"] and after["arrived"] >= before["arrived"]  # carried over the loads
        for f in want_a.dtype.names:
            assert np.array_equal(a[f].view(np.uint32), want_a[f].view(np.uint32)), (f, r)
        for f in ("position", "velocity", "predicted_position"):
            assert np.array_equal(b[f].view(np.uint32), want_b[f].view(np.uint32)), (f, r)
    assert len({res[r][0] for r in range(world)}) == 1  # the same refusal everywhere
    assert sum(res[r][3] for r in range(world)) == pos.shape[0]
    assert sum(res[r][5]["left"] for r in range(world)) > 0
