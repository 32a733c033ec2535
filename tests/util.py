"""Shared helpers of the parity tests: oracle <-> product parameter conversion and the
tolerance policy of SURVEY.md 8(c)."""
import numpy as np

FLOAT_FIELDS = ("position", "density", "pressure", "velocity", "acceleration", "predicted_position")


def oracle_from_params(O, positions, params):
    """Build an oracle with the same uniforms as a product ws_params."""
    p = O.Props()
    for f in ("delta_time", "collision_damping", "smoothing_radius", "target_density", "pressure_scalar",
              "near_pressure_scalar", "viscosity_strength"):
        setattr(p, f, getattr(params, f))
    return O.Oracle(positions, props=p, ext_min=list(params.ext_min), ext_max=list(params.ext_max),
                    gravity=list(params.gravity))


def oracle_one_step(O, orc, state, reverse=False, mode=None):
    """Run ONE oracle step from `state` (teacher forcing) and return the resulting particles.
    The permutation is reset to the identity so the result depends on `state` only."""
    orc.set_particles(state)
    orc.particle_indicies[:] = np.arange(orc.n, dtype=np.uint32)
    orc.set_reverse_order(reverse)
    orc.step(O.SORT_EXACT if mode is None else mode)
    orc.set_reverse_order(False)
    return orc.particles.copy()


def reorder_noise_tolerances(a, b, scale=4.0):
    """Tolerance per field = scale x the oracle's own reorder noise (L-inf between the oracle run
    with neighbours visited as written and in reversed order), floored at 4 ulp of the field's
    magnitude so a noise-free case does not demand bit equality of a different summation order."""
    tol = {}
    for f in FLOAT_FIELDS:
        noise = float(np.max(np.abs(a[f].astype(np.float64) - b[f].astype(np.float64))))
        mag = float(np.max(np.abs(a[f])))
        tol[f] = scale * noise + 4.0 * np.finfo(np.float32).eps * max(mag, 1e-30)
    return tol


def assert_particles_close(got, want, tol, what=""):
    for f in FLOAT_FIELDS:
        err = float(np.max(np.abs(got[f].astype(np.float64) - want[f].astype(np.float64))))
        assert err <= tol[f], "%s field %s: L-inf error %.3e > tolerance %.3e" % (what, f, err, tol[f])
    # .w components stay exactly 0 (SURVEY.md 8c KAT 10)
    for f in ("position", "velocity", "acceleration", "predicted_position"):
        assert not np.any(got[f][:, 3]), "%s: %s.w != 0" % (what, f)
