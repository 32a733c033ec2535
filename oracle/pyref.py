"""A SECOND CPU restatement of the reference path, in Python with numpy binary32 scalars, for small N.

TEST INFRASTRUCTURE ONLY (never imported by the product; only tests/ use it).  PARITY UNPINNED BY THE REFERENCE: the
reference holds no tests or vectors and cannot run here; this file is one more independent READING of its source,
written from the WGSL / Rust text -- not from oracle/ws_oracle.c -- so that the C restatement is checked bit for bit by
a restatement in another language with another code shape (tests/test_oracle_python_restatement.py).  Two gcc/x86-64
and hipcc/gfx950 restatements agreeing (the oracle and the test-only reference-order build) is one pin; this is a third.

Every function cites the reference lines it follows (paths relative to /root/reference).  Semantics fixed where WGSL
leaves them open, the same choices the C oracle states: IEEE binary32 for every operation (numpy float32 scalar
arithmetic rounds each operation once), no contraction, evaluation order as written, correctly rounded sqrt and
division, distance() summed x, y, z, w, vector operations component by component, u32 arithmetic wrapping.
Pure-Python loops: use for N <= a few hundred."""
import numpy as np

F = np.float32
U32 = 0xFFFFFFFF
INF = 999999999                                   # assets/simulation.wgsl:36
P1, P2, P3 = 15823, 9737333, 440817757            # assets/simulation.wgsl:38-40
LOOKAHEAD_FACTOR = F(1.0) / F(50.0)               # assets/simulation.wgsl:3
DENSITY_PADDING = F(0.00001)                      # assets/simulation.wgsl:4
OFFSET_TABLE = [(x, y, z) for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1)]  # assets/simulation.wgsl:6-34


def powi(a, n):
    """Rust f32::powi -> llvm.powi.f32 -> compiler-rt __powisf2: square-and-multiply, every product rounded."""
    a, r = F(a), F(1.0)
    while True:
        if n & 1:
            r = F(r * a)
        n //= 2
        if n == 0:
            return r
        a = F(a * a)


class Props:
    """FluidStaticProps::default, src/fluid_compute.rs:20-27,:67-79."""

    def __init__(self):
        self.delta_time = F(1.0) / F(60.0)
        self.collision_damping = F(0.95)
        self.smoothing_radius = F(0.25)
        self.target_density = F(10.0)
        self.pressure_scalar = F(22.0)
        self.near_pressure_scalar = F(2.0)
        self.viscosity_strength = F(0.1)

    def smoothing_kernel(self):
        """get_smoothing_kernel, src/fluid_compute.rs:55-63 (PI = std::f32::consts::PI)."""
        pi, h = F(np.pi), self.smoothing_radius
        return {"pow2": F(F(15.0) / F(F(F(2.0) * pi) * powi(h, 5))),
                "pow2_der": F(F(15.0) / F(pi * powi(h, 5))),
                "pow3": F(F(15.0) / F(pi * powi(h, 6))),
                "pow3_der": F(F(45.0) / F(pi * powi(h, 6))),
                "spikey_pow3": F(F(315.0) / F(F(F(64.0) * pi) * powi(h, 9)))}


def cube_fluid(ni, nj, nk, r):
    """helpers::cube_fluid, src/helpers.rs:3-20."""
    r = F(r)
    off = [F(r - F(F(n) * r)) for n in (ni, nj, nk)]
    diam = F(r * F(2.0))
    return np.array([[F(F(F(i) * diam) + off[0]), F(F(F(j) * diam) + off[1]), F(F(F(k) * diam) + off[2])]
                     for i in range(ni) for j in range(nj) for k in range(nk)], np.float32)


def get_ext(position, size, padding):
    """FluidContainer::get_ext, src/fluid_container.rs:42-50."""
    mn = [F(F(F(p) - F(F(s) / F(2.0))) + F(padding)) for p, s in zip(position, size)] + [F(0)]
    mx = [F(F(F(p) + F(F(s) / F(2.0))) - F(padding)) for p, s in zip(position, size)] + [F(0)]
    return mn, mx


def bit_sorter_stages(n):
    """get_bit_sorter_stages, src/fluid_compute.rs:251-273."""
    p = 1
    while p < n:
        p <<= 1
    stages, dim = [], 2
    while dim <= p:
        block = dim >> 1
        while block > 0:
            stages.append((block, dim))
            block >>= 1
        dim <<= 1
    return stages


class PyRef:
    """The reference's buffer set and its six passes (src/fluid_compute.rs:299-363)."""

    def __init__(self, positions, props=None, ext_min=None, ext_max=None, gravity=(0.0, -9.8, 0.0, 0.0)):
        self.n = len(positions)
        self.props = props or Props()
        self.kernel = self.props.smoothing_kernel()
        if ext_min is None:
            ext_min, ext_max = get_ext((0.0, 0.0, 0.0), (16.0, 9.0, 9.0), 0.1)   # src/fluid_container.rs:8-9, fluid_compute.rs:302
        self.ext_min, self.ext_max = [F(v) for v in ext_min], [F(v) for v in ext_max]
        self.gravity = [F(v) for v in gravity]                                     # src/gravity.rs:29-33
        z4 = lambda: [F(0)] * 4
        # FluidParticle::make_vec_from_positions, src/fluid_compute.rs:118-130
        self.position = [[F(p[0]), F(p[1]), F(p[2]), F(0)] for p in positions]
        self.predicted = [list(p) for p in self.position]
        self.velocity = [z4() for _ in range(self.n)]
        self.acceleration = [z4() for _ in range(self.n)]
        self.density = [[F(0), F(0)] for _ in range(self.n)]
        self.pressure = [[F(0), F(0)] for _ in range(self.n)]
        self.particle_indicies = list(range(self.n))        # src/fluid_compute.rs:243-249,:306-308
        self.particle_cell_indicies = list(range(self.n))
        self.cell_offsets = list(range(self.n))

    # ---- assets/simulation.wgsl:93-128 ----
    def sk(self, dst):
        v = F(self.props.smoothing_radius - dst)
        return F(F(v * v) * self.kernel["pow2"])

    def sk_near(self, dst):
        v = F(self.props.smoothing_radius - dst)
        return F(F(F(v * v) * v) * self.kernel["pow3"])

    def sk_der(self, dst):
        return F(F(dst - self.props.smoothing_radius) * self.kernel["pow2_der"])

    def sk_der_near(self, dst):
        v = F(dst - self.props.smoothing_radius)
        return F(F(v * v) * self.kernel["pow3_der"])

    def sk_visc(self, dst):
        h = self.props.smoothing_radius
        v = F(F(h * h) - F(dst * dst))
        return F(F(F(v * v) * v) * self.kernel["spikey_pow3"])

    def get_cell(self, p):
        return [int(np.floor(F(p[c] / self.props.smoothing_radius))) for c in range(3)]

    def hash_cell(self, cell):
        x, y, z = [c & U32 for c in cell]                                  # vec3<u32>(cell_index): reinterpretation
        return ((x * P1 + y * P2 + z * P3) & U32) % self.n

    @staticmethod
    def distance(a, b):
        d = [F(a[c] - b[c]) for c in range(4)]
        s = F(d[0] * d[0])
        for c in (1, 2, 3):
            s = F(s + F(d[c] * d[c]))
        return F(np.sqrt(s))

    # ---- the passes, in the order src/fluid_compute.rs:309-363 adds them ----
    def hash_particles(self):                                              # assets/simulation.wgsl:130-141
        for index in range(self.n):
            self.cell_offsets[index] = INF
            pid = self.particle_indicies[index]
            self.particle_cell_indicies[pid] = self.hash_cell(self.get_cell(self.predicted[pid]))

    def bitonic_sort(self, block, dim):                                    # assets/bitonic_sort.wgsl:22-46
        idx, keys = self.particle_indicies, self.particle_cell_indicies
        for i in range(self.n):
            j = i ^ block
            if j < i or i >= self.n:
                continue
            sign = -1 if (i & dim) != 0 else 1
            key_i, key_j = idx[i], idx[j]
            diff = (keys[key_i] - keys[key_j]) & U32                       # u32 subtraction wraps ...
            diff = diff - (1 << 32) if diff >= (1 << 31) else diff         # ... i32(..)
            if diff * sign > 0:
                idx[i], idx[j] = key_j, key_i

    def calculate_cell_offsets(self):                                      # assets/bitonic_sort.wgsl:48-59
        for index in range(self.n):
            cell = self.particle_cell_indicies[self.particle_indicies[index]]
            self.cell_offsets[cell] = min(self.cell_offsets[cell], index)

    def _neighbours(self, origin):
        """The 27-bucket walk both neighbour passes share (assets/simulation.wgsl:162-173, :219-230): yields neighbour ids."""
        cell = self.get_cell(origin)
        for off in OFFSET_TABLE:
            hash_index = self.hash_cell([cell[c] + off[c] for c in range(3)])
            it = self.cell_offsets[hash_index]
            while it < self.n:
                nb = self.particle_indicies[it]
                if self.particle_cell_indicies[nb] != hash_index:
                    break
                it += 1
                yield nb

    def update_density(self):                                              # assets/simulation.wgsl:143-195
        h = self.props.smoothing_radius
        for index in range(self.n):
            pid = self.particle_indicies[index]
            origin = self.predicted[pid]
            density, near = F(0), F(0)
            for nb in self._neighbours(origin):
                dst = self.distance(self.predicted[nb], origin)
                if dst > h:
                    continue
                density = F(density + self.sk(dst))
                near = F(near + self.sk_near(dst))
            density = F(density + DENSITY_PADDING)
            near = F(near + DENSITY_PADDING)
            self.density[pid] = [density, near]
            self.pressure[pid] = [F(self.props.pressure_scalar * F(density - self.props.target_density)),
                                  F(self.props.near_pressure_scalar * near)]

    def update_pressure_force(self):                                       # assets/simulation.wgsl:197-269
        h = self.props.smoothing_radius
        for index in range(self.n):
            pid = self.particle_indicies[index]
            origin, velocity = self.predicted[pid], self.velocity[pid]
            pressure, near_pressure = self.pressure[pid]
            pf, vf = [F(0)] * 3, [F(0)] * 3
            for nb in self._neighbours(origin):
                if pid == nb:
                    continue
                npred = self.predicted[nb]
                dst = self.distance(npred, origin)
                if dst > h:
                    continue
                d = [F(npred[c] - origin[c]) for c in range(3)]
                d = [F(d[c] / dst) for c in range(3)] if dst > F(0) else [F(0), F(1), F(0)]
                slope = self.sk_der(dst)
                shared = F(F(pressure + self.pressure[nb][0]) / F(2.0))
                slope_near = self.sk_der_near(dst)
                shared_near = F(F(near_pressure + self.pressure[nb][1]) / F(2.0))
                pf = [F(pf[c] + F(F(F(d[c] * shared) * slope) / self.density[nb][0])) for c in range(3)]
                pf = [F(pf[c] + F(F(F(d[c] * shared_near) * slope_near) / self.density[nb][1])) for c in range(3)]
                visc = self.sk_visc(dst)
                vf = [F(vf[c] + F(F(self.velocity[nb][c] - velocity[c]) * visc)) for c in range(3)]
            rho = self.density[pid][0]
            self.acceleration[pid] = [F(F(pf[c] / rho) + F(vf[c] * self.props.viscosity_strength)) for c in range(3)] + [F(0)]

    def integrate(self):                                                   # assets/simulation.wgsl:271-310
        dt, damp = self.props.delta_time, F(F(-1.0) * self.props.collision_damping)
        for i in range(self.n):
            v = [F(self.velocity[i][c] + F(F(self.gravity[c] + self.acceleration[i][c]) * dt)) for c in range(4)]
            x = [F(self.position[i][c] + F(v[c] * dt)) for c in range(4)]
            for c in range(3):
                if x[c] < self.ext_min[c]:
                    v[c] = F(v[c] * damp)
                    x[c] = self.ext_min[c]
                elif x[c] > self.ext_max[c]:
                    v[c] = F(v[c] * damp)
                    x[c] = self.ext_max[c]
            self.velocity[i], self.position[i] = v, x
            self.predicted[i] = [F(x[c] + F(v[c] * LOOKAHEAD_FACTOR)) for c in range(4)]

    def step(self):
        self.hash_particles()
        for block, dim in bit_sorter_stages(self.n):
            self.bitonic_sort(block, dim)
        self.calculate_cell_offsets()
        self.update_density()
        self.update_pressure_force()
        self.integrate()

    def records(self, dtype):
        out = np.zeros(self.n, dtype)
        out["position"] = np.array(self.position, np.float32)
        out["density"] = np.array(self.density, np.float32)
        out["pressure"] = np.array(self.pressure, np.float32)
        out["velocity"] = np.array(self.velocity, np.float32)
        out["acceleration"] = np.array(self.acceleration, np.float32)
        out["predicted_position"] = np.array(self.predicted, np.float32)
        return out
