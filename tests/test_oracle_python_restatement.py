"""The C oracle (oracle/ws_oracle.c) against a second, independent restatement of the reference path written in
Python with numpy binary32 scalars straight from the WGSL / Rust text (oracle/pyref.py): the host generators and the
six passes -- hash, the bitonic network stage by stage, atomicMin cell offsets, density, force, integrate -- must agree
BIT FOR BIT on every field and every index buffer, free-running.  Neither restatement is the reference (parity stays
unpinned: the reference has no vectors and cannot run here); two readings in two languages with two code shapes agreeing
to the last bit is what can be had, next to the third one on the GPU (tests/test_gpu_reference_order.py)."""
import numpy as np
import pytest

from oracle import pyref


def _compare(o, r, what):
    rec = r.records(o.particles.dtype)
    for f in o.particles.dtype.names:
        assert np.array_equal(o.particles[f].view(np.uint32), rec[f].view(np.uint32)), "%s: %s" % (what, f)
    assert np.array_equal(o.particle_indicies, np.array(r.particle_indicies, np.uint32)), "%s: particle_indicies" % what
    assert np.array_equal(o.particle_cell_indicies, np.array(r.particle_cell_indicies, np.uint32)), "%s: keys" % what
    assert np.array_equal(o.cell_offsets, np.array(r.cell_offsets, np.uint32)), "%s: cell_offsets" % what


def test_host_generators_and_constants_agree(oracle):
    for h in (0.25, 0.35, 0.15, 0.05):
        p = oracle.default_props()
        p.smoothing_radius = h
        q = pyref.Props()
        q.smoothing_radius = np.float32(h)
        a, b = oracle.smoothing_kernel(p), q.smoothing_kernel()
        for f in ("pow2", "pow2_der", "pow3", "pow3_der", "spikey_pow3"):
            assert np.float32(getattr(a, f)).view(np.uint32) == b[f].view(np.uint32), (h, f)
    assert np.array_equal(oracle.cube_fluid(4, 3, 5, 0.1), pyref.cube_fluid(4, 3, 5, 0.1))
    assert np.array_equal(oracle.cube_fluid(8, 4, 4, 0.07), pyref.cube_fluid(8, 4, 4, 0.07))
    for size in ((16, 9, 9), (16, 18, 0.2), (5.5, 3.25, 1.0)):
        a, b = oracle.get_ext((0.5, -1, 2), size, 0.1), pyref.get_ext((0.5, -1, 2), size, 0.1)
        assert np.array_equal(a[0], np.float32(b[0])) and np.array_equal(a[1], np.float32(b[1]))
    for n in (1, 2, 5, 64, 4096, 65536):
        assert oracle.bit_sorter_stages(n) == pyref.bit_sorter_stages(n)


@pytest.mark.parametrize("case", ["lattice-64", "clump-128", "planar-64"])
def test_six_passes_free_running_bit_for_bit(oracle, case):
    rng = np.random.default_rng(42)
    gravity = (0.0, -9.8, 0.0, 0.0)
    vel = None
    if case == "lattice-64":                    # spacing 0.2 < h: every particle has neighbours; acceleration cancels by symmetry
        pos = pyref.cube_fluid(4, 4, 4, 0.1)
        mn, mx = pyref.get_ext((0.0, 0.0, 0.0), (1.6, 1.2, 1.6), 0.1)
    elif case == "clump-128":                   # a dense random clump with random velocities in a box it bounces around in
        mn, mx = pyref.get_ext((0.0, 0.0, 0.0), (1.4, 1.0, 1.2), 0.1)
        pos = rng.uniform(-0.45, 0.45, (128, 3)).astype(np.float32)
        pos[5] = pos[4]                         # two coincident particles: the (0, 1, 0) direction branch
        vel = rng.uniform(-6.0, 6.0, (128, 3)).astype(np.float32)
        gravity = (3.0, -9.8, -1.5, 0.0)
    else:                                       # the planar C1 construction: z pinned to 0 by a 0.2-thick container
        mn, mx = pyref.get_ext((0.0, 0.0, 0.0), (2.0, 2.2, 0.2), 0.1)
        pos = pyref.cube_fluid(8, 8, 1, 0.1)
    o = oracle.Oracle(pos, ext_min=[float(v) for v in mn], ext_max=[float(v) for v in mx], gravity=list(gravity))
    r = pyref.PyRef(pos, ext_min=mn, ext_max=mx, gravity=gravity)
    if vel is not None:
        o.particles["velocity"][:, :3] = vel
        o.particles["predicted_position"][:, :3] = o.particles["position"][:, :3] + vel * np.float32(0.02)
        for i in range(len(pos)):
            r.velocity[i][:3] = [np.float32(v) for v in vel[i]]
            r.predicted[i] = [np.float32(r.position[i][c] + np.float32(r.velocity[i][c] * pyref.LOOKAHEAD_FACTOR)) for c in range(4)]
    _compare(o, r, case + " t=0")
    hits, rho_max, acc_max = 0, 0.0, 0.0
    for step in range(4):
        # pass by pass on the first step (a mismatch names its pass), then whole steps
        if step == 0:
            o.hash_particles(); r.hash_particles(); _compare(o, r, case + " hash")
            for block, dim in pyref.bit_sorter_stages(len(pos)):
                o.bitonic_stage(block, dim); r.bitonic_sort(block, dim)
            _compare(o, r, case + " sort")
            o.calculate_cell_offsets(); r.calculate_cell_offsets(); _compare(o, r, case + " offsets")
            o.update_density(); r.update_density(); _compare(o, r, case + " density")
            o.update_pressure_force(); r.update_pressure_force(); _compare(o, r, case + " force")
            o.integrate(); r.integrate(); _compare(o, r, case + " integrate")
        else:
            o.step(oracle.SORT_EXACT); r.step(); _compare(o, r, "%s step %d" % (case, step))
        rho_max = max(rho_max, float(o.particles["density"][:, 0].max()))
        acc_max = max(acc_max, float(np.abs(o.particles["acceleration"]).max()))
        hits += int(np.sum((o.particles["position"][:, :3] == np.float32(mn[:3])) | (o.particles["position"][:, :3] == np.float32(mx[:3]))))
    assert acc_max > 0 and rho_max > 160.0   # real neighbour work (the self term alone is 152.8)
    if case == "clump-128":
        assert hits > 0, "the clump case should go through wall reflections"
