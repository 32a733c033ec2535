"""ctypes binding of the CPU parity oracle (oracle/ws_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.  PARITY UNPINNED BY THE
REFERENCE (no reference tests / golden vectors exist; see ws_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libwsoracle.so")

INF = 999999999

PARTICLE_DTYPE = np.dtype(
    [
        ("position", np.float32, 4),
        ("density", np.float32, 2),
        ("pressure", np.float32, 2),
        ("velocity", np.float32, 4),
        ("acceleration", np.float32, 4),
        ("predicted_position", np.float32, 4),
    ]
)
assert PARTICLE_DTYPE.itemsize == 80


class Props(C.Structure):
    _fields_ = [
        ("delta_time", C.c_float),
        ("collision_damping", C.c_float),
        ("smoothing_radius", C.c_float),
        ("target_density", C.c_float),
        ("pressure_scalar", C.c_float),
        ("near_pressure_scalar", C.c_float),
        ("viscosity_strength", C.c_float),
    ]


class Kernel(C.Structure):
    _fields_ = [
        ("pow2", C.c_float),
        ("pow2_der", C.c_float),
        ("pow3", C.c_float),
        ("pow3_der", C.c_float),
        ("spikey_pow3", C.c_float),
    ]


class BitSorter(C.Structure):
    _fields_ = [("block", C.c_uint32), ("dim", C.c_uint32)]


class _State(C.Structure):
    _fields_ = [
        ("num_particles", C.c_uint32),
        ("props", Props),
        ("kernel", Kernel),
        ("ext_min", C.c_float * 4),
        ("ext_max", C.c_float * 4),
        ("gravity", C.c_float * 4),
        ("particles", C.c_void_p),
        ("particle_indicies", C.c_void_p),
        ("particle_cell_indicies", C.c_void_p),
        ("cell_offsets", C.c_void_p),
        ("reverse_order", C.c_int),
    ]


def build(force=False):
    """Compile libwsoracle.so with the committed Makefile (gcc, -ffp-contract=off)."""
    src = os.path.join(_HERE, "ws_oracle.c")
    hdr = os.path.join(_HERE, "ws_oracle.h")
    if (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libwsoracle.so"])
    return _LIB_PATH


_lib = None


def cpu_share():
    """(cores this process may actually use, why): the scheduler affinity, cut to the cgroup's CPU quota when one is set
    (a GPU box shows every host core -- 256 -- but grants a share of them: cpu.max = "1600000 100000" = 16 CPUs; more
    OpenMP threads than that only time-slice: measured on such a box at C3, s/step with 8 / 16 / 32 / 64 / 128 / 256
    threads: 1.14 / 0.71 / 0.74 / 0.84 / 1.30 / 3.10)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    why = "scheduler affinity: %d" % avail
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            q = max(1, int(int(quota) / int(period)))
            if q < avail:
                why = "cgroup cpu.max = %s/%s grants %d of the %d visible cores" % (quota, period, q, avail)
                avail = q
    except (OSError, ValueError):
        pass
    return max(1, avail), why


def default_threads():
    """OpenMP threads the oracle uses: $WSO_THREADS, else every core this process may use (cpu_share)."""
    env = os.environ.get("WSO_THREADS")
    if env:
        return max(1, int(env))
    return cpu_share()[0]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.wso_default_props.argtypes = [C.POINTER(Props)]
        L.wso_smoothing_kernel.argtypes = [C.POINTER(Props), C.POINTER(Kernel)]
        L.wso_cube_fluid.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p]
        L.wso_get_ext.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        L.wso_default_gravity.argtypes = [C.c_void_p]
        L.wso_bit_sorter_stages.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32]
        L.wso_bit_sorter_stages.restype = C.c_uint32
        L.wso_make_particles.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.wso_identity.argtypes = [C.c_uint32, C.c_void_p]
        L.wso_uniform_cloud.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wso_get_cell.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
        L.wso_hash_cell.argtypes = [C.c_void_p, C.c_uint32]
        L.wso_hash_cell.restype = C.c_uint32
        for name in (
            "wso_hash_particles",
            "wso_sort_exact",
            "wso_sort_fast",
            "wso_calculate_cell_offsets",
            "wso_update_density",
            "wso_update_pressure_force",
            "wso_integrate",
        ):
            getattr(L, name).argtypes = [C.POINTER(_State)]
            getattr(L, name).restype = None
        L.wso_bitonic_sort_stage.argtypes = [C.POINTER(_State), C.c_uint32, C.c_uint32]
        L.wso_step.argtypes = [C.POINTER(_State), C.c_int]
        L.wso_max_threads.restype = C.c_int
        L.wso_set_threads.argtypes = [C.c_int]
        # a GPU box shows every host core but grants a share of them: cap the OpenMP team
        L.wso_set_threads(default_threads())
        _lib = L
    return _lib


SORT_EXACT = 0
SORT_FAST = 1


def default_props():
    p = Props()
    lib().wso_default_props(C.byref(p))
    return p


def smoothing_kernel(props):
    k = Kernel()
    lib().wso_smoothing_kernel(C.byref(props), C.byref(k))
    return k


def cube_fluid(ni, nj, nk, r=0.1):
    out = np.empty((ni * nj * nk, 3), np.float32)
    lib().wso_cube_fluid(ni, nj, nk, r, out.ctypes.data)
    return out


def get_ext(position, size, padding=0.1):
    pos = np.asarray(position, np.float32)
    sz = np.asarray(size, np.float32)
    mn = np.zeros(4, np.float32)
    mx = np.zeros(4, np.float32)
    lib().wso_get_ext(pos.ctypes.data, sz.ctypes.data, padding, mn.ctypes.data, mx.ctypes.data)
    return mn, mx


def default_gravity():
    g = np.zeros(4, np.float32)
    lib().wso_default_gravity(g.ctypes.data)
    return g


def bit_sorter_stages(n):
    cnt = lib().wso_bit_sorter_stages(n, None, 0)
    arr = (BitSorter * cnt)()
    lib().wso_bit_sorter_stages(n, arr, cnt)
    return [(s.block, s.dim) for s in arr]


def uniform_cloud(n, seed, ext_min, ext_max):
    out = np.empty((n, 3), np.float32)
    mn = np.ascontiguousarray(ext_min, np.float32)
    mx = np.ascontiguousarray(ext_max, np.float32)
    lib().wso_uniform_cloud(n, seed, mn.ctypes.data, mx.ctypes.data, out.ctypes.data)
    return out


def get_cell(pos, h):
    p = np.ascontiguousarray(pos, np.float32)
    c = np.zeros(3, np.int32)
    lib().wso_get_cell(p.ctypes.data, h, c.ctypes.data)
    return c


def hash_cell(cell, n):
    c = np.ascontiguousarray(cell, np.int32)
    return int(lib().wso_hash_cell(c.ctypes.data, n))


class Oracle:
    """The reference worker's buffer set + pass sequence (src/fluid_compute.rs:277-366)."""

    def __init__(self, positions, props=None, ext_min=None, ext_max=None, gravity=None):
        L = lib()
        positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        n = positions.shape[0]
        self.n = n
        self.particles = np.zeros(n, PARTICLE_DTYPE)
        L.wso_make_particles(positions.ctypes.data, n, self.particles.ctypes.data)
        self.particle_indicies = np.arange(n, dtype=np.uint32)
        self.particle_cell_indicies = np.arange(n, dtype=np.uint32)
        self.cell_offsets = np.arange(n, dtype=np.uint32)
        self.st = _State()
        self.st.num_particles = n
        self.st.props = props if props is not None else default_props()
        if ext_min is None:
            ext_min, ext_max = get_ext((0, 0, 0), (16, 9, 9), 0.1)
        g = default_gravity() if gravity is None else np.asarray(gravity, np.float32)
        for i in range(4):
            self.st.ext_min[i] = float(ext_min[i]) if i < len(ext_min) else 0.0
            self.st.ext_max[i] = float(ext_max[i]) if i < len(ext_max) else 0.0
            self.st.gravity[i] = float(g[i]) if i < len(g) else 0.0
        self.st.reverse_order = 0
        self.refresh_kernel()
        self._bind()

    def _bind(self):
        self.st.particles = self.particles.ctypes.data
        self.st.particle_indicies = self.particle_indicies.ctypes.data
        self.st.particle_cell_indicies = self.particle_cell_indicies.ctypes.data
        self.st.cell_offsets = self.cell_offsets.ctypes.data

    def refresh_kernel(self):
        """update() re-derives the kernel constants every frame (fluid_compute.rs:480)."""
        lib().wso_smoothing_kernel(C.byref(self.st.props), C.byref(self.st.kernel))

    def set_particles(self, particles):
        self.particles[...] = particles
        self._bind()

    def set_reverse_order(self, flag):
        self.st.reverse_order = 1 if flag else 0

    # individual passes
    def hash_particles(self):
        lib().wso_hash_particles(C.byref(self.st))

    def sort(self, mode=SORT_EXACT):
        (lib().wso_sort_exact if mode == SORT_EXACT else lib().wso_sort_fast)(C.byref(self.st))

    def bitonic_stage(self, block, dim):
        lib().wso_bitonic_sort_stage(C.byref(self.st), block, dim)

    def calculate_cell_offsets(self):
        lib().wso_calculate_cell_offsets(C.byref(self.st))

    def update_density(self):
        lib().wso_update_density(C.byref(self.st))

    def update_pressure_force(self):
        lib().wso_update_pressure_force(C.byref(self.st))

    def integrate(self):
        lib().wso_integrate(C.byref(self.st))

    def step(self, mode=SORT_EXACT):
        lib().wso_step(C.byref(self.st), mode)

    def sorted_keys(self):
        return self.particle_cell_indicies[self.particle_indicies]


def max_threads():
    return lib().wso_max_threads()


def set_threads(n):
    lib().wso_set_threads(n)
