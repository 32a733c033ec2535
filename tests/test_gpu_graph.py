"""WS_FLAG_GRAPH: the step replayed from a captured hipGraph (the reference builds its pass graph once and replays it
every frame, src/fluid_compute.rs:309-363,:396).  Same bits as direct launches, on single and slab handles, through
parameter pushes, a parameter change, a radius change and a reset."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _program(ws, pos, params, later, radius):
    def run(w):
        w.run(6)
        for _ in range(5):
            w.set_params(params)      # update() pushes the same parameters every frame: the capture must survive that
            w.run(1)
        w.set_params(later)           # a HUD edit: gravity / viscosity
        w.run(6)
        w.set_params(radius)          # ... and the smoothing radius: a re-grid
        w.run(6)
        a = w.read_vec("particles")
        w.reset(pos)                  # despawn_liquid
        w.set_params(params)
        w.run(8)
        return a, w.read_vec("particles"), w.stats()
    return run


def test_single_handle_graph_replay_equals_direct_launches(ws):
    size = (16.0, 9.0, 9.0)
    params = ws.make_params(container_size=size, gravity=(6.0, -9.8, 0.0, 0.0))
    later = ws.make_params(container_size=size, gravity=(-8.0, -9.8, 2.0, 0.0), viscosity_strength=0.2)
    radius = ws.make_params(container_size=size, gravity=(-8.0, -9.8, 2.0, 0.0), viscosity_strength=0.2, smoothing_radius=0.35)
    pos = ws.workloads.uniform_cloud(65536, 9, list(params.ext_min), list(params.ext_max))
    run = _program(ws, pos, params, later, radius)
    d = ws.FluidWorker(pos, params)
    want = run(d)
    d.close()
    g = ws.FluidWorker(pos, params, graph=True)
    got = run(g)
    g.close()
    assert want[2]["graph_steps"] == 0
    assert got[2]["graph_steps"] >= 20, got[2]
    for a, b in ((want[0], got[0]), (want[1], got[1])):
        for f in a.dtype.names:
            assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), f


def test_slab_step_graph_replay_through_the_native_rccl_transport(ws):
    """One rank through the library's own RCCL transport (the multi-rank launch sequence minus the neighbours): the
    captured slab step -- migration bookkeeping, sort, K4, K5, the status ring copy on its side stream -- reproduces
    the plain single handle bit for bit, and the steps really are replays."""
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(65536, 7, list(params.ext_min), list(params.ext_max))
    w = ws.FluidWorker(pos, params)
    w.run(40)
    want = w.read_vec("particles")
    w.close()
    tr = ws.slab.NativeRcclTransport(ws.slab.NativeRcclTransport.unique_id(), 0, 1, 0)
    assert tr.communicators() == 2
    ids = np.arange(pos.shape[0], dtype=np.uint32)
    s = ws.slab.SlabWorker(pos, ids, pos.shape[0], params, 0, 1, tr, graph=True)
    s.run(40)
    got = s.read_vec("particles")
    st = s.stats()
    s.close()
    tr.close()
    assert st["graph_steps"] >= 38, st
    for f in want.dtype.names:
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f


def test_a_host_transport_is_never_captured(ws):
    """A host-supplied transport (here the tests' loopback, which synchronises the stream and meets the other slabs on a
    host barrier inside its callbacks) cannot be recorded into a graph: such handles ignore the flag and launch
    directly -- same results, nobody hangs."""
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(32768, 5, list(params.ext_min), list(params.ext_max))
    w = ws.FluidWorker(pos, params)
    w.run(12)
    want = w.read_vec("particles")
    w.close()

    def program(s, rank):
        s.run(12)
        return s.read_vec("particles"), s.stats()

    for got, st in ws.slab.run_loopback_program(pos, params, 2, program, graph=True):
        assert st["graph_steps"] == 0
        for f in want.dtype.names:
            assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f


def test_captured_slab_step_survives_loads_re_grids_and_re_cuts(ws):
    """One rank through the RCCL transport with WS_FLAG_GRAPH, through everything that invalidates a capture: a reset, a
    state upload, a smoothing-radius change (arrays re-allocated), a re-cut, a parameter change -- every time the next
    steady-state step is captured again and the results equal the plain single handle given the same calls."""
    size = (16.0, 9.0, 9.0)
    params = ws.make_params(container_size=size, gravity=(6.0, -9.8, 0.0, 0.0))
    radius = ws.make_params(container_size=size, gravity=(6.0, -9.8, 0.0, 0.0), smoothing_radius=0.35)
    later = ws.make_params(container_size=size, gravity=(-6.0, -9.8, 1.0, 0.0), smoothing_radius=0.35, viscosity_strength=0.3)
    pos = ws.workloads.uniform_cloud(50000, 17, list(params.ext_min), list(params.ext_max))

    def program(w, slab):
        w.run(6)
        state = w.read_vec("particles")
        w.reset(pos)
        w.run(5)
        (w.write_particles if slab else lambda s: w.write_slice("particles", s))(state)
        w.run(5)
        w.set_params(radius)
        w.run(5)
        if slab:
            w.rebalance()
        w.run(5)
        w.set_params(later)
        w.run(5)
        return w.read_vec("particles"), w.read_positions(), w.stats()

    d = ws.FluidWorker(pos, params)
    want, want_pos, _ = program(d, False)
    d.close()
    tr = ws.slab.NativeRcclTransport(ws.slab.NativeRcclTransport.unique_id(), 0, 1, 0)
    ids = np.arange(pos.shape[0], dtype=np.uint32)
    s = ws.slab.SlabWorker(pos, ids, pos.shape[0], params, 0, 1, tr, graph=True)
    got, got_pos, st = program(s, True)
    s.close()
    tr.close()
    assert st["graph_steps"] >= 20, st
    assert np.array_equal(got_pos.view(np.uint32), want_pos.view(np.uint32))
    for f in want.dtype.names:
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f
