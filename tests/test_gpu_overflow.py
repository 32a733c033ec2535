"""A particle with more candidates than K4's accept mask holds (> 1024): its workgroup must take the
full-test K5 kernel and still match the other variants bit for bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_mask_overflow_workgroups_take_the_full_kernel(ws):
    params = ws.make_params(container_size=(6.0, 6.0, 6.0))
    rng = np.random.default_rng(3)
    # a dense clump: ~3000 particles inside one 27-cell neighbourhood, plus a sparse background
    clump = rng.uniform(-0.3, 0.3, (3000, 3)).astype(np.float32)
    back = ws.workloads.uniform_cloud(5192, 9, list(params.ext_min), list(params.ext_max))
    pos = np.concatenate([clump, back])  # 8192 particles
    outs = {}
    for variant in ("simple", "listed"):
        os.environ["WS_VARIANT"] = variant
        try:
            w = ws.FluidWorker(pos, params)
        finally:
            os.environ.pop("WS_VARIANT", None)
        w.run(3)
        outs[variant] = w.read_vec("particles")
        w.close()
    for f in outs["simple"].dtype.names:
        assert np.array_equal(outs["simple"][f].view(np.uint32), outs["listed"][f].view(np.uint32)), f
