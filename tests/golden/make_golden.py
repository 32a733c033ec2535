#!/usr/bin/env python3
"""Generate the golden fixtures in this directory with the CPU oracle (exact sort mode).

    python tests/golden/make_golden.py

The reference ships no golden vectors and cannot be run here, so these are outputs
of the oracle -- the committed restatement of the reference WGSL -- not of the reference itself.
They freeze the oracle (any change in its arithmetic shows up as a diff) and give the GPU tests
fixed expected values.  Each .npz holds the inputs (params, initial positions are regenerated from
the recorded generator arguments) and, for free-running exact-mode steps 1, 2, 9, 10: position,
velocity, density, acceleration; plus the three index buffers after step 1.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

STEPS = (1, 2, 9, 10)

CASES = {
    # name: (generator, args, container size)
    "cube_8x16x8": ("cube", (8, 16, 8), (4.0, 5.0, 4.0)),
    "planar_32x32": ("cube", (32, 32, 1), (8.0, 9.0, 0.2)),
    "cloud_1024": ("cloud", (1024, 0x5EED00AA), (5.0, 3.0, 3.0)),
    # config 1 of BASELINE.md (4 096 particles): integer artefacts + final positions only
    "c1_lattice_4096": ("cube", (64, 64, 1), (16.0, 18.0, 0.2)),
    "c1_cloud_4096": ("cloud", (4096, 0x5EED0001), (16.0, 18.0, 0.2)),
}


def positions(gen, args, mn, mx):
    if gen == "cube":
        return O.cube_fluid(*args, 0.1)
    return O.uniform_cloud(args[0], args[1], mn, mx)


def main():
    for name, (gen, args, size) in CASES.items():
        mn, mx = O.get_ext((0, 0, 0), size, 0.1)
        pos = positions(gen, args, mn, mx)
        orc = O.Oracle(pos, ext_min=mn, ext_max=mx)
        out = {"container_size": np.float32(size), "ext_min": mn, "ext_max": mx, "gen_args": np.int64(args),
               "gen_is_cloud": np.int64(gen == "cloud"), "n": np.int64(pos.shape[0])}
        small = pos.shape[0] <= 1024
        for s in range(1, max(STEPS) + 1):
            orc.step(O.SORT_EXACT)
            if s == 1:
                out["keys_by_id_1"] = orc.particle_cell_indicies.copy()
                out["sorted_keys_1"] = orc.sorted_keys().copy()
                out["cell_offsets_1"] = orc.cell_offsets.copy()
            if s in STEPS and (small or s == max(STEPS)):
                P = orc.particles
                out["position_%d" % s] = P["position"][:, :3].copy()
                out["velocity_%d" % s] = P["velocity"][:, :3].copy()
                if small:
                    out["density_%d" % s] = P["density"].copy()
                    out["acceleration_%d" % s] = P["acceleration"][:, :3].copy()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-18s n=%5d  %7.1f KB" % (name, pos.shape[0], os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
