"""bench.py's host-side bookkeeping that needs no GPU: which geometry an N-GPU run takes and which committed one-GPU
line it is compared with (multi-GPU speed-up is reported like for like -- the SAME configuration on one
GPU -- not against C3)."""
import argparse
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _args(**kw):
    base = dict(config=None, replicate=False, copies=0)
    base.update(kw)
    return argparse.Namespace(**base)


def test_geometry_by_gpu_count():
    b = _bench()
    assert b.dist_geometry(_args(), 1)[0] == "c3"
    assert b.dist_geometry(_args(), 4)[0] == "c4"   # BASELINE.json config 4
    assert b.dist_geometry(_args(), 8)[0] == "c5"   # BASELINE.json config 5
    name, block, size = b.dist_geometry(_args(), 2)  # no 2-GPU config in BASELINE.json: C3 twice along x
    assert name == "c3x2" and block == (512, 128, 128) and size == (128.0, 36.0, 36.0)
    # the same geometry on ONE GPU (the like-for-like run of the N = 2 line)
    assert b.dist_geometry(_args(config="c3", copies=2), 1) == (name, block, size)


def test_same_config_one_gpu_reads_the_committed_line():
    b = _bench()
    for cfg in ("c4", "c5"):
        got = b.same_config_one_gpu(cfg, "cloud")
        assert got is not None and got["ms_per_step"] > 0 and got["settled_ms_per_step"] > got["ms_per_step"]
        src = got["source"].split(" ")[0]
        with open(os.path.join(ROOT, src)) as f:
            line = json.load(f)
        assert line["n_gpus"] == 1 and line["config"]["particles"] == got["particles"]
        assert line["config"]["workload"].upper().startswith(cfg.upper())
    assert b.same_config_one_gpu("c1x2", "cloud") is None
    assert b.same_config_one_gpu("c4", "lattice") is None
