"""Loading the golden fixtures of tests/golden/ (see make_golden.py there)."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["cube_8x16x8", "planar_32x32", "cloud_1024", "c1_lattice_4096", "c1_cloud_4096"]
SMALL = CASES[:3]


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def initial_positions(gen, g):
    """Regenerate the fixture's initial positions with generator module `gen` (oracle or product)."""
    args = [int(x) for x in g["gen_args"]]
    if int(g["gen_is_cloud"]):
        return gen.uniform_cloud(args[0], args[1], g["ext_min"], g["ext_max"])
    return gen.cube_fluid(*args)


def state_at(g, step, dtype):
    """80-byte particle records of the fixture at `step` (position, velocity, predicted)."""
    pos = g["position_%d" % step]
    vel = g["velocity_%d" % step]
    out = np.zeros(pos.shape[0], dtype)
    out["position"][:, :3] = pos
    out["velocity"][:, :3] = vel
    # assets/simulation.wgsl:309, two f32 operations (numpy does not contract them)
    out["predicted_position"][:, :3] = pos + vel * np.float32(0.02)
    if "density_%d" % step in g:
        out["density"] = g["density_%d" % step]
        out["acceleration"][:, :3] = g["acceleration_%d" % step]
    return out
