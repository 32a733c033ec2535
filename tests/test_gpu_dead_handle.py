"""A re-grid that fails AFTER the old cell tables were given up leaves the handle dead (include/wsfluid.h, ws_set_params):
every later entry point that would touch device arrays must return the reason as an error code -- never launch over
half-built tables (only ws_step checked).  The allocation failure is forced through the developer build's
WS_FAIL_REGRID hook (tests/libwsfluid_dev.so); the product library has no such switch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _every_entry_point_refuses(ws, w, pos, state, params):
    calls = {
        "ws_step": lambda: w.run(1),
        "ws_read_positions": w.read_positions,
        "ws_read_particles": w.read_vec,
        "ws_read_speeds": w.read_speeds,
        "ws_read_sort_view": w.sort_view,
        "ws_reset": lambda: w.reset(pos),
        "ws_write_particles": lambda: (w.write_slice("particles", state) if hasattr(w, "write_slice") else w.write_particles(state)),
        "ws_set_params": lambda: w.set_params(params),
    }
    for name, call in calls.items():
        with pytest.raises(ws.WsError) as e:
            call()
        assert "unusable" in str(e.value), (name, str(e.value))
        assert e.value.status == 4, (name, e.value.status)  # WS_ERR_HIP


def test_a_dead_single_handle_refuses_every_call_that_touches_device_arrays(ws, devlib, monkeypatch):
    pos, params = ws.workloads.make_workload("c1", "cloud")
    w = ws.FluidWorker(pos, params, library=devlib)
    w.run(3)
    state = w.read_vec("particles")
    smaller = ws.make_params(container_size=ws.workloads.CONFIGS["c1"][1], smoothing_radius=np.float32(0.15))
    monkeypatch.setenv("WS_FAIL_REGRID", "1")
    with pytest.raises(ws.WsError):
        w.set_params(smaller)
    monkeypatch.delenv("WS_FAIL_REGRID")
    _every_entry_point_refuses(ws, w, pos, state, params)
    # the asynchronous readback pair too
    buf = np.empty((w.n, 3), np.float32)
    with pytest.raises(ws.WsError):
        w.read_positions_begin(buf)
    w.close()  # ws_destroy still frees what is left


def test_dead_slabs_refuse_alike_on_every_rank(ws, devlib, monkeypatch):
    """Two loopback slabs: the re-grid fails on BOTH (the verdict of a re-grid is made common by an all-gather), and from
    then on every collective call returns the error on every rank instead of entering the collective."""
    params = ws.make_params(container_size=(16.0, 9.0, 9.0))
    pos = ws.workloads.uniform_cloud(32768, 5, list(params.ext_min), list(params.ext_max))
    smaller = ws.make_params(container_size=(16.0, 9.0, 9.0), smoothing_radius=np.float32(0.15))
    import os

    def program(s, rank):
        s.run(3)
        state = s.read_vec("particles")
        if rank == 1:
            os.environ["WS_FAIL_REGRID"] = "1"   # (one process: both ranks see it; what matters is that they fail ALIKE)
        try:
            s.set_params(smaller)
            failed = False
        except ws.WsError:
            failed = True
        assert failed
        os.environ.pop("WS_FAIL_REGRID", None)
        _every_entry_point_refuses(ws, s, pos, state, params)
        return True

    assert ws.slab.run_loopback_program(pos, params, 2, program, library=devlib) == [True, True]
