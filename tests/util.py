"""Shared helpers of the parity tests: oracle <-> product parameter conversion and the
tolerance policy of SURVEY.md 8(c).

Every float comparison against the oracle goes through assert_particles_close, which also records how much of
each tolerance the comparison used (error / tolerance per field); conftest.py writes the records of a session to
gpurun_out/parity_report.json (committed per round under profiles/rNN/parity_report.json)."""
import numpy as np

FLOAT_FIELDS = ("position", "density", "pressure", "velocity", "acceleration", "predicted_position")

# one record per (comparison, field): filled by assert_particles_close / record_bitwise, dumped by conftest.py
PARITY_REPORT = []


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


def _linf(a, b):
    """max |a - b| in float64 without materialising float64 copies of multi-GB fields."""
    worst = 0.0
    flat_a, flat_b = a.reshape(len(a), -1), b.reshape(len(b), -1)
    for s in range(0, len(a), 1 << 22):
        d = flat_a[s:s + (1 << 22)].astype(np.float64) - flat_b[s:s + (1 << 22)].astype(np.float64)
        if d.size:
            worst = max(worst, float(np.max(np.abs(d))))
    return worst


def reorder_noise_tolerances(a, b, scale=4.0):
    """Tolerance per field = scale x the oracle's own reorder noise (L-inf between the oracle run
    with neighbours visited as written and in reversed order), floored at 4 ulp of the field's
    magnitude so a noise-free case does not demand bit equality of a different summation order."""
    tol = {}
    noise_of = {}
    for f in FLOAT_FIELDS:
        noise = _linf(a[f], b[f])
        mag = float(np.max(np.abs(a[f])))
        tol[f] = float(scale * noise + 4.0 * float(np.finfo(np.float32).eps) * max(mag, 1e-30))
        noise_of[f] = noise
    tol["_noise"] = noise_of
    return tol


def assert_particles_close(got, want, tol, what="", arithmetic=None):
    noise = tol.get("_noise", {})
    failures = []
    for f in FLOAT_FIELDS:
        err = _linf(got[f], want[f])
        PARITY_REPORT.append({"case": what, "arithmetic": arithmetic, "field": f, "n": int(len(got)), "linf_error": err,
                              "noise_unit": noise.get(f), "tolerance": float(tol[f]),
                              "error_over_tolerance": err / tol[f] if tol[f] > 0 else (0.0 if err == 0 else float("inf"))})
        if not err <= tol[f]:
            failures.append("%s field %s: L-inf error %.3e > tolerance %.3e" % (what, f, err, tol[f]))
    assert not failures, "; ".join(failures)
    # .w components stay exactly 0 (SURVEY.md 8c KAT 10)
    for f in ("position", "velocity", "acceleration", "predicted_position"):
        assert not np.any(got[f][:, 3]), "%s: %s.w != 0" % (what, f)


def assert_particles_bitwise(got, want, what="", arithmetic=None, fields=None):
    """Every named field equal BIT FOR BIT (K1 / K6 outputs of particles without neighbours, integer artefacts)."""
    for f in fields or want.dtype.names:
        same = np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32))
        PARITY_REPORT.append({"case": what, "arithmetic": arithmetic, "field": f, "n": int(len(got)),
                              "linf_error": 0.0 if same else _linf(got[f], want[f]), "noise_unit": 0.0, "tolerance": 0.0,
                              "error_over_tolerance": 0.0 if same else float("inf"), "bitwise": True})
        assert same, "%s: field %s differs from the oracle (must be bit-exact)" % (what, f)
