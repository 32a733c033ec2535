"""Multi-GPU slab decomposition, exercised on ONE GPU: `world` slab handles in one process (one host
thread each) exchange halos and migrants through the loopback transport.  Because the in-cell order
is canonical (by particle id) and every slab uses the global y/z cell layout, the merged result must
equal the single-handle run BIT FOR BIT -- including after particles have migrated between slabs."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _single(ws, pos, params, steps, ieee=False):
    w = ws.FluidWorker(pos, params, ieee_division=ieee)
    w.run(steps)
    out = w.read_vec("particles")
    w.close()
    return out


@pytest.mark.parametrize("world", [2, 3, 4])
def test_slabs_reproduce_single_gpu_bitwise(ws, world):
    # gravity tilted along +x so that fluid crosses slab boundaries (migration) within a few steps
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(65536, 1234, list(params.ext_min), list(params.ext_max))
    steps = 40
    want = _single(ws, pos, params, steps)
    got, owned = ws.slab.run_loopback(pos, params, world, steps)
    assert sum(owned) == pos.shape[0]
    first = np.bincount(ws.slab.assign(params, pos, world), minlength=world)
    assert list(first) != owned, "the test must actually migrate particles"
    for f in want.dtype.names:
        bad = np.any(got[f].view(np.uint32) != want[f].view(np.uint32), axis=1)
        assert not bad.any(), "%s: %d particles differ (first ids %s), max |diff| %.3e, x of the first %s" % (
            f, int(bad.sum()), np.flatnonzero(bad)[:8], float(np.max(np.abs(got[f] - want[f]))),
            want["position"][np.flatnonzero(bad)[:8], 0])


def test_particles_that_cross_several_slabs_in_one_step_take_the_far_route(ws):
    """Eight thin slabs (8-9 cell layers = ~2.1 units each) and a gravity of 3000 along +x: from the third step on the
    fluid moves more than a slab's width per step (0.83, 1.67, 2.5, 3.3 units), so leavers skip their neighbour and
    travel in the "far" message addressed to their destination rank (one per destination, exchanged all-to-all).  In
    the fifth step the whole fluid comes back off the far wall at once and keeps sloshing across several slabs per step.
    The merged result still equals the single handle bit for bit, and the counters show the route was used."""
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(3000.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(4096, 77, list(params.ext_min), list(params.ext_max))
    steps, world = 12, 8
    want = _single(ws, pos, params, steps)
    counters = {}
    got, owned = ws.slab.run_loopback(pos, params, world, steps, counters=counters)
    assert sum(owned) == pos.shape[0]
    far = sum(c["far"] for c in counters.values())
    left = sum(c["left"] for c in counters.values())
    arrived = sum(c["arrived"] for c in counters.values())
    assert far > 1000, "the far route was not exercised: %r" % (counters,)
    assert left == arrived and left > 0, counters
    assert [counters[r]["owned"] for r in range(world)] == owned
    for f in want.dtype.names:
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f


def test_parameter_change_on_slab_handles_matches_the_single_handle(ws):
    """ws_set_params on slab handles in the middle of a run (gravity turned, viscosity doubled -- what the reference's
    update() pushes every frame): the slabs keep their layer ranges, neighbours and migration targets, and the merged
    result equals the single handle given the same change at the same step."""
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))
    later = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(-8.0, -9.8, 2.0, 0.0), viscosity_strength=0.2)
    pos = ws.workloads.uniform_cloud(32768, 21, list(params.ext_min), list(params.ext_max))
    w = ws.FluidWorker(pos, params)
    w.run(12)
    w.set_params(later)
    w.run(18)
    want = w.read_vec("particles")
    w.close()
    got, owned = ws.slab.run_loopback(pos, params, 3, 30, change_params=(12, later))
    assert sum(owned) == pos.shape[0]
    for f in want.dtype.names:
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f
    # (a new smoothing radius re-grids slab handles too: tests/test_gpu_slab_frame.py)


def test_slab_assign_matches_cell_cuts(ws):
    params = ws.make_params(container_size=(16.0, 9.0, 9.0))
    pos = ws.workloads.uniform_cloud(20000, 5, list(params.ext_min), list(params.ext_max))
    for world in (1, 2, 5, 8):
        r = ws.slab.assign(params, pos, world)
        assert r.min() == 0 and r.max() == world - 1
        # monotone in the x cell index
        cx = np.floor(pos[:, 0] / np.float32(0.25)).astype(np.int64)
        order = np.argsort(cx, kind="stable")
        assert np.all(np.diff(r[order].astype(np.int64)) >= 0)


def test_single_slab_world_of_one_equals_plain_handle(ws):
    params = ws.make_params(container_size=(8.0, 6.0, 6.0))
    pos = ws.cube_fluid(16, 16, 8)
    want = _single(ws, pos, params, 12)
    got, _ = ws.slab.run_loopback(pos, params, 1, 12)
    for f in want.dtype.names:
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f


def test_slabs_reproduce_single_gpu_bitwise_with_ieee_division(ws):
    """The same with WS_FLAG_IEEE_DIVISION on every handle (the flag must reach the slab kernels)."""
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(32768, 4321, list(params.ext_min), list(params.ext_max))
    want = _single(ws, pos, params, 25, ieee=True)
    got, _ = ws.slab.run_loopback(pos, params, 3, 25, ieee_division=True)
    other = _single(ws, pos, params, 25, ieee=False)
    assert not np.array_equal(other["acceleration"].view(np.uint32), want["acceleration"].view(np.uint32))
    for f in want.dtype.names:
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f


@pytest.mark.parametrize("world", [2, 3, 4])
def test_slabs_with_halo_overlap_give_the_same_bits(ws, world):
    """The default: halos on a second stream, K4 / K5 split into an early range and the late boundary layers."""
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(65536, 1234, list(params.ext_min), list(params.ext_max))
    want = _single(ws, pos, params, 40)
    got, owned = ws.slab.run_loopback(pos, params, world, 40, overlap=True)
    assert sum(owned) == pos.shape[0]
    for f in want.dtype.names:
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f


def test_slabs_without_halo_overlap_give_the_same_bits(ws):
    """WS_FLAG_NO_OVERLAP keeps halos and kernels on one stream (no early / late split); same result."""
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(32768, 99, list(params.ext_min), list(params.ext_max))
    want = _single(ws, pos, params, 25)
    got, _ = ws.slab.run_loopback(pos, params, 3, 25, overlap=False)
    for f in want.dtype.names:
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f


def test_native_rccl_transport_single_rank(ws):
    """The library's own RCCL transport (communicator of one rank: the collectives run, there are no neighbours):
    a slab handle driven through it reproduces the plain handle bit for bit."""
    params = ws.make_params(container_size=(16.0, 9.0, 9.0))
    pos = ws.workloads.uniform_cloud(32768, 7, list(params.ext_min), list(params.ext_max))
    want = _single(ws, pos, params, 12)
    tr = ws.slab.NativeRcclTransport(ws.slab.NativeRcclTransport.unique_id(), 0, 1, 0)
    ids = np.arange(pos.shape[0], dtype=np.uint32)
    w = ws.slab.SlabWorker(pos, ids, pos.shape[0], params, 0, 1, tr)
    w.run(12)
    rec, got_ids = w.read()
    w.close()
    tr.close()
    got = np.zeros_like(want)
    got[got_ids] = rec
    for f in want.dtype.names:
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f


def test_native_rccl_collectives_execute_on_a_communicator_of_one():
    """A slab step with one rank makes no transport call, so the test above never reaches librccl's collectives.  Here
    the transport's entry points are called directly on a one-rank communicator of the REAL librccl: ncclAllToAll (RCCL's
    own symbol, bound by name at run time) and ncclAllGather run and deliver -- in a process of its own that loads the
    library first and never imports torch, so that it runs on the system's HIP runtime and librccl as a C++ host's would
    (torch bundles its own copies under the same sonames: whichever is loaded first serves the whole process), once with
    the transport as the process's very first HIP user.  What a one-GPU box can say about the binding; with peers:
    tests/test_gpu_fake_rccl.py (stand-in) and tests/test_gpu_rccl_two_gpus.py (needs two GPUs)."""
    import subprocess
    import sys

    for extra in ([], ["rccl_first"]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_one_rank_probe_notorch.py"), "1"] + extra,
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        assert "alltoall rc 0 equal True" in out.stdout and "allgather rc 0 equal True" in out.stdout, out.stdout[-2000:]


@pytest.mark.parametrize("what", ["ghost_capacity", "capacity"])
def test_capacity_overrun_fails_every_rank_alike_and_hangs_nobody(ws, what):
    """A slab whose halo messages (ghost_capacity) or owned range (capacity) are too small: the kernels clamp and set
    a sticky error bit, the bit travels to every rank with the next step's all-gather, and ws_step fails with
    WS_ERR_OUT_OF_MEMORY on EVERY rank at the same step, before any collective of that step -- nobody is left
    waiting inside one."""
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(65536, 1234, list(params.ext_min), list(params.ext_max))
    world = 3
    if what == "ghost_capacity":
        kw = dict(ghost_capacity=64)           # a boundary layer holds ~1000 particles
    else:
        kw = dict(capacity=65536 // world + 256)  # the tilted gravity piles the fluid up in the last slab
    errs = ws.slab.run_loopback(pos, params, world, 80, collect_errors=True, **kw)
    assert sorted(errs) == list(range(world)), "every rank must fail: %r" % (errs,)
    steps = {k for k, _ in errs.values()}
    assert len(steps) == 1, "all ranks must fail at the same step: %r" % (errs,)
    for k, e in errs.values():
        assert e.status == 3 and "capacity" in str(e), e  # WS_ERR_OUT_OF_MEMORY


def test_slab_step_only_enqueues(ws):
    """`ws_step` on a slab handle never waits for the step it enqueues (counts stay on the device, messages have fixed
    sizes); the only thing it may wait for is the status table of the step enqueued TWO calls earlier (a bounded
    run-ahead: it is what lets every rank act on the same table).  So after a burst of calls the handle is still
    busy -- at least the last two steps are still queued or running -- and the calls return before the work is done."""
    import time

    pos, params = ws.workloads.make_workload("c3", "cloud")
    tr = ws.slab.NativeRcclTransport(ws.slab.NativeRcclTransport.unique_id(), 0, 1, 0)
    ids = np.arange(pos.shape[0], dtype=np.uint32)
    w = ws.slab.SlabWorker(pos, ids, pos.shape[0], params, 0, 1, tr)
    w.run(3)
    w.sync()
    ready = ws.fluid.C.c_int(0)
    assert w._L.ws_ready(w._h, ws.fluid.C.byref(ready)) == 0 and ready.value == 1
    t0 = time.perf_counter()
    w.run(12)
    t_enqueued = time.perf_counter() - t0
    assert w._L.ws_ready(w._h, ws.fluid.C.byref(ready)) == 0
    busy = ready.value == 0
    w.sync()
    t_done = time.perf_counter() - t0
    assert busy, "ws_ready reported 1 right after 12 slab steps of 4 194 304 particles were enqueued"
    assert t_done - t_enqueued > 0.5e-3, "the last steps had already run when ws_step returned (%.2f / %.2f ms)" % (
        t_enqueued * 1e3, t_done * 1e3)
    assert w.num_owned() == pos.shape[0]
    w.close()
    tr.close()


@pytest.mark.parametrize("fixed", ["0", "1"])
def test_message_sizes_follow_the_fluid(ws, fixed):
    """WS_FLAG_LAGGED_MESSAGES.  The buffers of a slab's messages have fixed capacities (sized for the violent phase of a collapsing cloud); what
    TRAVELS per step is a prefix sized from what every rank reported a few steps ago -- the same all-gathered table on
    every rank, hence the same size at both ends of every exchange.  WS_FLAG_FIXED_MESSAGES keeps the capacities.
    Either way the result is the single handle's, bit for bit."""
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(65536, 1234, list(params.ext_min), list(params.ext_max))
    steps, world = 40, 3
    want = _single(ws, pos, params, steps)

    def program(s, rank):
        s.run(steps)
        return s.read_vec("particles"), s.stats()

    res = ws.slab.run_loopback_program(pos, params, world, program, lagged_messages=fixed == "0", fixed_messages=fixed == "1")
    for rec, st in res:
        for f in want.dtype.names:
            assert np.array_equal(rec[f].view(np.uint32), want[f].view(np.uint32)), f
        assert 0 < st["migration_peak"] <= st["migration_now"] <= st["migration_capacity"]
        assert 0 < st["halo_peak"] <= st["halo_now"] <= st["halo_capacity"]
        if fixed == "1":
            assert (st["migration_now"], st["halo_now"], st["far_now"]) == (st["migration_capacity"], st["halo_capacity"], st["far_capacity"])
        else:
            assert st["migration_now"] < st["migration_capacity"] // 4 and st["halo_now"] < st["halo_capacity"]
            assert st["far_now"] < st["far_capacity"] // 16
    assert len({(st["migration_now"], st["halo_now"], st["far_now"]) for _, st in res}) == 1  # the same on every rank


def test_four_slabs_through_the_collapse_and_rebound_of_c2_match_the_single_handle(ws):
    """The benchmark trajectory, not a few quiet steps: C2 (262 144 particles) in four slabs through the collapse of the
    cloud and its rebound (220 steps) -- migration by the ten thousand per step, the multi-kernel migration fill, the LAGGED
    message sizes following the fluid up and down (WS_FLAG_LAGGED_MESSAGES; exact sizes: tests/test_gpu_slab_exact.py) -- is the single handle's result bit for bit, with the DEFAULT capacities
    (rounds 1-3 never stepped a multi-slab run this far: every multi-GPU benchmark configuration overran there)."""
    pos, params = ws.workloads.make_workload("c2", "cloud")
    steps, world = 220, 4
    want = _single(ws, pos, params, steps)

    def program(s, rank):
        sizes = []
        for k in range(steps // 20):
            s.run(20)
            sizes.append(s.stats()["migration_now"])
        return s.read_vec("particles"), s.stats(), sizes, s.counters()

    res = ws.slab.run_loopback_program(pos, params, world, program, lagged_messages=True)
    for rec, st, sizes, c in res:
        for f in want.dtype.names:
            assert np.array_equal(rec[f].view(np.uint32), want[f].view(np.uint32)), f
        assert st["migration_peak"] <= st["migration_capacity"] and st["halo_peak"] <= st["halo_capacity"]
        assert all(4096 <= x <= st["migration_capacity"] // 4 for x in sizes), sizes  # sized by the fluid, not the capacity
    assert sum(c["left"] for _, _, _, c in res) > 10000


def test_slabs_created_with_different_flags_fail_alike_instead_of_hanging(ws):
    """Every choice that fixes the step's collective sequence (message sizing, halo overlap, capture) must be the same on
    every rank: ws_slab_create all-gathers a mode word and fails on ALL ranks when they differ (a rank that
    differs used to issue its halos on the other communicator and the run hung inside the transport)."""
    import threading

    params = ws.make_params(container_size=(16.0, 9.0, 9.0))
    pos = ws.workloads.uniform_cloud(16384, 11, list(params.ext_min), list(params.ext_max))
    owner = ws.slab.assign(params, pos, 2)
    hub = ws.slab.LoopbackHub(2)
    seen = [None, None]

    def body(r):
        sel = np.flatnonzero(owner == r).astype(np.uint32)
        try:
            w = ws.slab.SlabWorker(pos[sel], sel, pos.shape[0], params, r, 2, hub.transport(r), overlap=(r == 0))
            w.close()
            seen[r] = "created"
        except ws.WsError as e:
            seen[r] = (e.status, str(e))

    ts = [threading.Thread(target=body, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for r in range(2):
        assert seen[r] != "created" and seen[r][0] == 1 and "different flags" in seen[r][1], seen  # WS_ERR_INVALID_ARG on both
