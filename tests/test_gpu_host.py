"""The compiled C++ host mirror (water-sandbox_amd/host) driving the library in the reference's frame order."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _exe():
    exe = os.path.join(ROOT, "water-sandbox_amd", "host", "frame_loop")
    if not os.path.exists(exe):
        import runpy

        runpy.run_path(os.path.join(ROOT, "water-sandbox_amd", "host", "build_host.py"), run_name="__main__")
    return exe


@pytest.mark.parametrize("world", [2, 3, 4])
def test_cpp_frame_loop_on_slabs_is_bit_identical_to_the_single_handle(world):
    """host/frame_loop.cpp --slabs W: the reference's frame order (update -> despawn -> run, src/fluid_compute.rs:468-525)
    from C++ on one handle and on W x-slabs (one host thread each, the library's in-process transport), with two
    smoothing-radius changes, a reset and a gravity change on the way: every frame's id-ordered positions equal."""
    out = subprocess.run([_exe(), "--slabs", str(world), "50"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bit-identical to the single handle on every rank" in out.stdout


def test_frame_loop_runs_reset_and_param_change():
    exe = _exe()
    out = subprocess.run([exe, "60"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "60 frames, 65536 particles" in out.stdout
    lines = [l for l in out.stdout.splitlines() if l.startswith("frame")]
    ys = [float(l.split("=")[1]) for l in lines]
    assert ys[1] < ys[0]          # falling under gravity
    assert ys[3] > ys[2]          # frame 30 reset put the particle back up
