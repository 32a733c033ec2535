"""The oracle reproduces the committed golden fixtures bit for bit (CPU only)."""
import numpy as np
import pytest

import golden_util as G


@pytest.mark.parametrize("name", G.CASES)
def test_oracle_reproduces_golden(oracle, name):
    g = G.load(name)
    pos = G.initial_positions(oracle, g)
    assert pos.shape[0] == int(g["n"])
    orc = oracle.Oracle(pos, ext_min=g["ext_min"], ext_max=g["ext_max"])
    for s in range(1, 11):
        orc.step(oracle.SORT_EXACT)
        if s == 1:
            assert np.array_equal(orc.particle_cell_indicies, g["keys_by_id_1"])
            assert np.array_equal(orc.sorted_keys(), g["sorted_keys_1"])
            assert np.array_equal(orc.cell_offsets, g["cell_offsets_1"])
        if "position_%d" % s in g:
            for f, key in (("position", "position_%d"), ("velocity", "velocity_%d")):
                assert np.array_equal(orc.particles[f][:, :3].view(np.uint32), g[key % s].view(np.uint32)), (s, f)
        if "density_%d" % s in g:
            assert np.array_equal(orc.particles["density"].view(np.uint32), g["density_%d" % s].view(np.uint32))
            assert np.array_equal(orc.particles["acceleration"][:, :3].view(np.uint32),
                                  g["acceleration_%d" % s].view(np.uint32))


def test_fixture_predicted_position_reconstruction(oracle):
    """state_at() rebuilds predicted_position with the same two f32 operations as integrate."""
    g = G.load("cube_8x16x8")
    orc = oracle.Oracle(G.initial_positions(oracle, g), ext_min=g["ext_min"], ext_max=g["ext_max"])
    orc.step(oracle.SORT_EXACT)
    st = G.state_at(g, 1, oracle.PARTICLE_DTYPE)
    assert np.array_equal(st["predicted_position"].view(np.uint32), orc.particles["predicted_position"].view(np.uint32))
