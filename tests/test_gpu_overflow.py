"""A particle with more candidates than K4's accept mask holds (> 2048): its wave must take the full sweep in the
force kernel and still match the one-thread-per-particle kernels bit for bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(ws, pos, params, variant, ieee, devlib=None):
    # `simple` = the developer build's WS_VARIANT hook; `listed` = the product library as it ships
    if variant == "listed":
        w = ws.FluidWorker(pos, params, ieee_division=ieee)
    else:
        os.environ["WS_VARIANT"] = variant
        try:
            w = ws.FluidWorker(pos, params, ieee_division=ieee, library=devlib)
        finally:
            os.environ.pop("WS_VARIANT", None)
    w.run(3)
    out, stats = w.read_vec("particles"), w.stats()
    w.close()
    return out, stats


@pytest.mark.parametrize("ieee", [False, True], ids=["hw-rcp-sqrt", "ieee-division"])
def test_mask_overflow_waves_take_the_full_sweep(ws, devlib, ieee):
    params = ws.make_params(container_size=(6.0, 6.0, 6.0))
    rng = np.random.default_rng(3)
    # a dense clump: ~5000 particles inside one 27-cell neighbourhood, plus a sparse background
    clump = rng.uniform(-0.3, 0.3, (5000, 3)).astype(np.float32)
    back = ws.workloads.uniform_cloud(3192, 9, list(params.ext_min), list(params.ext_max))
    pos = np.concatenate([clump, back])  # 8192 particles
    want, _ = _run(ws, pos, params, "simple", ieee, devlib)
    got, stats = _run(ws, pos, params, "listed", ieee)
    assert stats["mask_overflow"] > 1000
    for f in want.dtype.names:
        assert np.array_equal(want[f].view(np.uint32), got[f].view(np.uint32)), f
