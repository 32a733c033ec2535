"""Smoothing radii far below the default: no cliff.

The HUD changes the smoothing radius by 0.1 per key press (src/hud.rs:135-138) and update() re-uploads it every frame
(src/fluid_compute.rs:479-480); the reference's N-bucket hashed table (assets/simulation.wgsl:125-128) costs the same for
every radius.  A dense grid of reference-sized cells does not (C3 at h = 0.05: 6.6e8 cells), so beyond a cell budget the
library merges grid cells along z, then y, then x (csrc/ws_api.cpp derive_dev).  Cell edges stay >= h: the neighbour
SET is untouched, and every result still matches the oracle."""
import numpy as np
import pytest

from util import assert_particles_close, oracle_from_params, oracle_one_step, reorder_noise_tolerances

pytestmark = pytest.mark.gpu


def _one_step_vs_oracle(oracle, ws, pos, params, label, state=None, ieee=False, library=None):
    O = oracle
    orc = oracle_from_params(O, pos, params)
    st = orc.particles.copy() if state is None else state.astype(O.PARTICLE_DTYPE)
    want = oracle_one_step(O, orc, st, mode=O.SORT_FAST)
    ints = (orc.particle_cell_indicies.copy(), orc.sorted_keys().copy(), orc.cell_offsets.copy())
    rev = oracle_one_step(O, orc, st, reverse=True, mode=O.SORT_FAST)
    w = ws.FluidWorker(pos, params, ieee_division=ieee, library=library)
    w.write_slice("particles", st)
    w.run()
    got = w.read_vec("particles")
    keys, perm, off = w.sort_view()
    stats, dims = w.stats(), w.grid_dims()
    w.close()
    assert np.array_equal(keys, ints[0]) and np.array_equal(keys[perm], ints[1]) and np.array_equal(off, ints[2])
    assert_particles_close(got, want, reorder_noise_tolerances(want, rev), label, "ieee-division" if ieee else "hw-rcp-sqrt")
    return stats, dims


@pytest.mark.parametrize("h,merged", [(0.05, True), (0.15, False)])
def test_c3_at_the_radii_two_key_presses_reach(oracle, ws, h, merged):
    """BASELINE.json config 3 at h = 0.05 and h = 0.15 (one and two presses of the radius key from the default 0.25):
    creates, steps, and one teacher-forced step of all 4 194 304 particles matches the oracle."""
    pos, _ = ws.workloads.make_workload("c3", "cloud")
    params = ws.make_params(container_size=ws.workloads.CONFIGS["c3"][1], smoothing_radius=np.float32(h))
    stats, dims = _one_step_vs_oracle(oracle, ws, pos, params, "c3 cloud h=%.2f full size" % h)
    assert (stats["cells_merged"] != (1, 1, 1)) == merged, (stats, dims)
    assert dims[0] * dims[1] * dims[2] <= 1 << 29


def test_radius_key_presses_during_a_run_never_fail_and_stay_on_the_oracle(oracle, ws):
    """0.25 -> 0.15 -> 0.05 (f32 arithmetic of `smoothing_radius -= 0.1`, hud.rs:137) on a running C3 handle: every
    ws_set_params succeeds (the old 2^29-cell cap made the second one fail), and a teacher-forced step from the state
    reached matches the oracle at the final radius."""
    pos, params = ws.workloads.make_workload("c3", "cloud")
    size = ws.workloads.CONFIGS["c3"][1]
    w = ws.FluidWorker(pos, params)
    h = np.float32(0.25)
    for _ in range(2):
        w.run(15)
        h = np.float32(h - np.float32(0.1))
        w.set_params(ws.make_params(container_size=size, smoothing_radius=h))
    w.run(15)
    state = w.read_vec("particles")
    assert w.stats()["cells_merged"] != (1, 1, 1)
    w.close()
    assert np.isfinite(state["position"]).all() and np.isfinite(state["density"]).all()
    _one_step_vs_oracle(oracle, ws, pos, ws.make_params(container_size=size, smoothing_radius=h),
                        "c3 after two radius presses (h=%.8f)" % h, state=state)


@pytest.mark.parametrize("budget,axes", [("50000", "z"), ("2000", "zy"), ("64", "zyx")])
@pytest.mark.parametrize("ieee", [False, True], ids=["hw-rcp-sqrt", "ieee-division"])
def test_cells_merged_along_z_then_y_then_x(oracle, ws, devlib, monkeypatch, budget, axes, ieee):
    """WS_CELL_BUDGET (an environment hook of the developer build, tests/libwsfluid_dev.so) forces the merge on a small domain so that every stage of it is exercised:
    z only, z and y, all three axes -- against the oracle, several teacher-forced steps into a collapse."""
    monkeypatch.setenv("WS_CELL_BUDGET", budget)
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(4.0, -9.8, 2.0, 0.0))
    pos = ws.workloads.uniform_cloud(65536, 12, list(params.ext_min), list(params.ext_max))
    w = ws.FluidWorker(pos, params, library=devlib)
    w.run(25)
    state = w.read_vec("particles")
    merged = w.stats()["cells_merged"]
    w.close()
    assert [m > 1 for m in merged] == ["x" in axes, "y" in axes, "z" in axes], merged
    _one_step_vs_oracle(oracle, ws, pos, params, "merged cells %s" % (merged,), state=state, ieee=ieee, library=devlib)


@pytest.mark.parametrize("budget", ["2000", "64"])
def test_slabs_on_a_merged_grid_reproduce_the_single_handle(ws, devlib, monkeypatch, budget):
    """The slab cuts are expressed in grid layers along x -- merged ones too: bit-identical to the single handle."""
    monkeypatch.setenv("WS_CELL_BUDGET", budget)
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(65536, 1234, list(params.ext_min), list(params.ext_max))
    w = ws.FluidWorker(pos, params, library=devlib)
    w.run(30)
    want = w.read_vec("particles")
    w.close()
    got, owned = ws.slab.run_loopback(pos, params, 3, 30, library=devlib)
    assert sum(owned) == pos.shape[0]
    for f in want.dtype.names:
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f
