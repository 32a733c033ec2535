"""The library's RCCL transport (csrc/ws_rccl.cpp) with PEERS, on a one-GPU box: a child process that loads the DEVELOPER
build of the library (tests/libwsfluid_dev.so -- the product library honours no environment variable) and in which
WS_RCCL_LIBRARY points at the tests' stand-in for librccl (tests/fake_rccl/fake_rccl.hip: the eleven nccl* symbols, ranks =
host threads, stream-ordered device-side handshakes, capturable).  What runs is the product's own transport code --
ncclCommInitRank, ncclCommSplit, the grouped ncclSend / ncclRecv with `rank - 1` / `rank + 1`, ncclAllGather, one
communicator per stream -- under the slab step, directly and (WS_FLAG_GRAPH | WS_FLAG_GRAPH_MULTIRANK, fixed-capacity messages) inside captured hipGraphs.
It does not replace a run on two GPUs: the wire, RCCL's own kernels and its proxy threads are not here."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_transport_with_peers_direct_and_captured(ws):
    lib = ws.build.build_fake_rccl()
    env = dict(os.environ, WS_RCCL_LIBRARY=lib, HSA_ENABLE_IPC_MODE_LEGACY="0",
               GPU_MAX_HW_QUEUES="24")  # every stream of every rank on a hardware queue of its own: a rank's wait kernel
    #                                     must never sit in front of the kernel it waits for
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fake_rccl_worker.py")], env=env, capture_output=True,
                         text=True, timeout=900)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "fake_rccl_result.log"), "w") as f:
        f.write(out.stdout + "\n---- stderr ----\n" + out.stderr)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["library"] == lib
    assert [(c["world"], c["graph"], c["exact_messages"]) for c in res["cases"]] == [
        (2, False, False), (3, False, False), (4, False, False), (2, True, False), (4, True, False), (3, False, True)]
    for c in res["cases"]:
        W = c["world"]
        assert c["fake_errors"] == 0, c                       # (sampled by every rank while its communicators live) no wait timed
        #                                                       out, every message had the size its receiver expected
        assert c["communicators"] == [2] * W, c               # ncclCommSplit: one communicator per stream
        assert c["new_communicators"] == 2 * W, c
        # the step really went through the transport (host-side calls into the stand-in: a captured step makes them once
        # per capture, the replays run the recorded kernels)
        # (all-to-all: the far messages and the status words, once per step; all-gather: the collective reads of the program)
        assert c["sendrecv_ops"] > (10 if c["graph"] else 100) and c["alltoalls"] > (3 if c["graph"] else 50) and c["allgathers"] >= 4, c
        if c["exact_messages"]:
            assert c["allgathers"] > 100, c                   # two size all-gathers per step
        assert c["migrated"] > 0, c
        assert all(c["mid_frame_positions_identical"]), c
        assert all(c["bit_identical_to_single_handle"]), c
        if c["graph"]:
            assert min(c["graph_steps"]) >= 50, c             # replays of captured steps with transport calls inside
        else:
            assert c["graph_steps"] == [0] * W, c


def test_thin_and_thick_slabs_agree_on_the_halo_stream(tmp_path):
    """tests/fake_rccl_thin_worker.py: eight slabs of two or three cell layers each; the halos must
    travel on the same communicator on every rank (decided from the thinnest slab of the world -- a rank deciding from its
    own thickness, as until round 4, would exchange its halos on the other communicator of the RCCL transport)."""
    lib = os.path.join(ROOT, "tests", "libfakerccl.so")
    env = dict(os.environ, WS_RCCL_LIBRARY=lib, HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="24")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fake_rccl_thin_worker.py")], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["errors"] == [] and res["fake_errors"] == [0] * 8, res
    assert min(res["owned_layers"]) < 3 <= max(res["owned_layers"]), res  # the scenario: thin slabs beside a thick one
    assert res["identical"] == [True] * 8 and sum(res["owned"]) == 20000, res


def test_the_product_never_defaults_to_the_stand_in():
    """WS_RCCL_LIBRARY is a hook of the DEVELOPER build alone (csrc/ws_devhooks.h): no product source names the tests'
    library (build.py only knows how to compile it for the tests, as it does for the reference-order library), and the
    product binary does not even contain the variable's name."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "water-sandbox_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".inc", ".h", ".hpp")) and f != "build.py":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "libfakerccl" not in text, os.path.join(dirpath, f)
    rccl = open(os.path.join(ROOT, "water-sandbox_amd", "csrc", "ws_rccl.cpp")).read()
    assert 'WS_DEV_ENV("WS_RCCL_LIBRARY")' in rccl and "getenv" not in rccl and '"librccl.so.1"' in rccl
    product = open(os.path.join(ROOT, "water-sandbox_amd", "libwsfluid.so"), "rb").read()
    assert b"WS_RCCL_LIBRARY" not in product and b"libfakerccl" not in product
