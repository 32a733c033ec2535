"""The cost-guided tile schedule of K4 / K5 (csrc/ws_kernels.hip "tile schedule"): which handles take it, and
that it changes WHEN a tile runs and on which XCD, never what it computes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_which_handles_take_the_schedule(ws):
    """2^18 <= n < 2^20 particles (C2), not the reference's 65 536, not C1, not a captured step."""
    taken = {}
    for cfg, graph in (("c1", False), ("ref", False), ("c2", False), ("c2", True)):
        pos, params = ws.workloads.make_workload(cfg, "cloud")
        w = ws.FluidWorker(pos, params, graph=graph)
        w.run(3)
        taken[(cfg, graph)] = w.stats()["tile_schedule"]
        w.close()
    assert taken == {("c1", False): False, ("ref", False): False, ("c2", False): True, ("c2", True): False}, taken


@pytest.mark.parametrize("ieee", [False, True], ids=["hw-rcp-sqrt", "ieee-division"])
def test_the_schedule_changes_no_bit(ws, devlib, ieee):
    """C2 through the collapse (costs, cuts and tile order change every step) with the schedule forced off and on
    (the developer build's WS_TILE_SCHEDULE hook), and the product library as it ships: every field bit-identical."""
    pos, params = ws.workloads.make_workload("c2", "cloud")
    out = {}
    for name, lib, env in (("product", None, None), ("off", devlib, "0"), ("on", devlib, "1")):
        if env is not None:
            os.environ["WS_TILE_SCHEDULE"] = env
        try:
            w = ws.FluidWorker(pos, params, ieee_division=ieee, library=lib)
        finally:
            os.environ.pop("WS_TILE_SCHEDULE", None)
        w.run(140)
        out[name] = (w.read_vec("particles"), w.stats()["tile_schedule"])
        w.close()
    assert (out["product"][1], out["off"][1], out["on"][1]) == (True, False, True)
    for name in ("off", "on"):
        for f in out["product"][0].dtype.names:
            assert np.array_equal(out[name][0][f].view(np.uint32), out["product"][0][f].view(np.uint32)), (name, f)


def test_the_schedule_at_a_size_it_is_not_used_at_is_still_right(ws, devlib):
    """Forced on at C1 and at 1.5 M particles (outside the window the product uses it in): same bits."""
    for n, steps in ((4096, 30), (1536000, 12)):
        params = ws.make_params(container_size=(16.0, 9.0, 9.0))
        pos = ws.workloads.uniform_cloud(n, 21, list(params.ext_min), list(params.ext_max))
        res = []
        for env in ("0", "1"):
            os.environ["WS_TILE_SCHEDULE"] = env
            try:
                w = ws.FluidWorker(pos, params, library=devlib)
            finally:
                os.environ.pop("WS_TILE_SCHEDULE", None)
            w.run(steps)
            res.append(w.read_vec("particles"))
            w.close()
        for f in res[0].dtype.names:
            assert np.array_equal(res[0][f].view(np.uint32), res[1][f].view(np.uint32)), (n, f)
