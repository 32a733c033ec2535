"""Exact message sizes (the default of a slab handle; WS_FLAG_EXACT_MESSAGES asks for them explicitly): every message of a slab step at exactly its sender's count (ws_step waits for four words per rank
twice per step) instead of a size derived from the demand of a few steps earlier.  The lagged sizes never wait but FAIL the
run when a demand outgrows them within four steps -- a shock front reaching a slab face broadside makes the number of
particles that change owner grow tenfold in one step (tools/slab_series.py); nothing below the buffers' capacities can
overrun in this mode, and the messages are as small as they can be."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _single(ws, pos, params, steps):
    w = ws.FluidWorker(pos, params)
    w.run(steps)
    out = w.read_vec("particles")
    w.close()
    return out


@pytest.mark.parametrize("world", [2, 3, 5])
def test_exact_messages_reproduce_the_single_handle_bitwise(ws, world):
    params = ws.make_params(container_size=(16.0, 9.0, 9.0), gravity=(6.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(65536, 1234, list(params.ext_min), list(params.ext_max))
    steps = 40
    want = _single(ws, pos, params, steps)
    counters = {}
    got, owned = ws.slab.run_loopback(pos, params, world, steps, counters=counters, exact_messages=True)
    assert sum(owned) == pos.shape[0] and sum(c["left"] for c in counters.values()) > 0
    for f in want.dtype.names:
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f


def test_a_shock_front_that_overruns_the_lagged_sizes_passes_with_exact_ones(ws):
    """A dense cloud in a short box with a gravity of 60 along x (case 41 of tools/slab_fuzz.py 100 7): the leavers of one
    slab face go 8 905, 17 764, 35 103, 162 450 in consecutive steps.  The lagged sizing -- four times the largest
    demand of the last eight tables, four steps old -- is overrun and every rank fails at the same step, cleanly; with
    exact sizes the run is the single handle's, bit for bit."""
    params = ws.make_params(container_size=(8.0, 5.0, 9.0), gravity=(60.0, -9.8, 0.0, 0.0))
    pos = ws.workloads.uniform_cloud(600000, 99, list(params.ext_min), list(params.ext_max))
    world, steps = 4, 30
    errs = ws.slab.run_loopback(pos, params, world, steps, collect_errors=True, lagged_messages=True)
    assert sorted(errs) == list(range(world)), "the lagged sizes were expected to be overrun on every rank: %r" % (errs,)
    assert len({k for k, _ in errs.values()}) == 1 and all(e.status == 3 for _, e in errs.values()), errs
    want = _single(ws, pos, params, steps)
    got, owned = ws.slab.run_loopback(pos, params, world, steps, exact_messages=True)
    assert sum(owned) == pos.shape[0]
    for f in want.dtype.names:
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f


def test_exact_messages_carry_what_there_is_and_no_more(ws):
    """C2 in four slabs into the rebound of the collapsed cloud: the sizes reported after every 20 steps are the rounded-up
    counts of the step before -- far below the capacities, and below the default rule's floors while the fluid is calm."""
    pos, params = ws.workloads.make_workload("c2", "cloud")
    steps = 200
    want = _single(ws, pos, params, steps)

    def program(s, rank):
        sizes = []
        for _ in range(steps // 20):
            s.run(20)
            st = s.stats()
            sizes.append((st["migration_now"], st["halo_now"], st["far_now"]))
        return s.read()[0:2], s.stats(), sizes

    res = ws.slab.run_loopback_program(pos, params, 4, program, exact_messages=True)
    got = np.zeros_like(want)
    for (rec, ids), st, sizes in res:
        got[ids] = rec
        assert all(m <= st["migration_capacity"] // 8 and m % 64 == 0 for m, _, _ in sizes), sizes
        assert min(m for m, _, _ in sizes) < 4096, sizes  # (the default rule never goes below 4 096 records)
        assert st["migration_peak"] < st["migration_capacity"] and st["halo_peak"] < st["halo_capacity"]
    for f in want.dtype.names:
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f


def test_randomised_slab_runs_are_identical_with_exact_messages():
    """tools/slab_fuzz.py: random container, gravity (up to 3 000 along x), particle count, slab count (2..8) and length per
    case; with exact sizes every case must equal the single handle bit for bit (with the default sizes about a quarter
    of such cases are overrun -- profiles/r04/slab/slab_fuzz_*.log)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "slab_fuzz.py"), "30", "11", "exact"], capture_output=True,
                         text=True, timeout=900)
    last = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert out.returncode == 0 and last["cases"] == 30 and last["bad"] == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert last["capacity_overruns"] <= 2, out.stdout[-3000:]  # (a buffer's capacity itself: a third of the particles in one cell layer)
