"""ctypes binding of libwsfluid.so plus a host-side mirror of the reference's fluid
worker interface (src/fluid_compute.rs), used by the tests, bench.py and smoke().

This module is plumbing over the C ABI in include/wsfluid.h.  It never computes physics
itself and has no CPU fallback: if the HIP library is missing or no gfx950 device is
visible, construction raises.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

INF = 999999999  # assets/simulation.wgsl:36

# FluidParticle, src/fluid_compute.rs:106-115
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

WS_FLAG_PROFILE = 1
WS_FLAG_REFERENCE_ORDER = 2
WS_FLAG_IEEE_DIVISION = 4
WS_FLAG_GRAPH = 8
WS_FLAG_EXACT_MESSAGES = 16
WS_FLAG_LAGGED_MESSAGES = 32
WS_FLAG_FIXED_MESSAGES = 64
WS_FLAG_NO_OVERLAP = 128
WS_FLAG_GRAPH_MULTIRANK = 256
WS_ABI_VERSION = 2
KERNEL_IDS = {"cell_scan": 0, "cell_scatter": 1, "reorder": 2, "density": 3, "force_integrate_bin": 4, "bin": 5}

# every symbol include/wsfluid.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "ws_default_params", "ws_get_smoothing_kernel", "ws_cube_fluid", "ws_get_ext",
    "ws_bit_sorter_stage_count", "ws_status_string", "ws_abi_version", "ws_create", "ws_destroy",
    "ws_step", "ws_ready", "ws_sync", "ws_set_params", "ws_read_positions", "ws_read_particles",
    "ws_reset", "ws_write_particles", "ws_pin_host_buffer", "ws_unpin_host_buffer", "ws_read_speeds", "ws_read_positions_begin", "ws_read_positions_end", "ws_read_positions_view", "ws_slab_counters", "ws_rccl_unique_id", "ws_rccl_transport_create",
    "ws_rccl_transport_destroy", "ws_rccl_last_error", "ws_rccl_transport_communicators",
    "ws_local_hub_create", "ws_local_hub_destroy", "ws_local_transport_create", "ws_local_transport_destroy", "ws_read_sort_view", "ws_last_error", "ws_num_particles",
    "ws_steps_done", "ws_kernel_name", "ws_profile_read", "ws_profile_reset", "ws_profile_select",
    "ws_grid_dims", "ws_read_stats", "ws_slab_assign", "ws_slab_create", "ws_slab_read_particles", "ws_slab_rebalance", "ws_slab_balanced_cuts",
]


class WsParams(C.Structure):
    """ws_params: FluidStaticProps + Gravity + FluidContainerExt."""

    _fields_ = [
        ("delta_time", C.c_float),
        ("collision_damping", C.c_float),
        ("smoothing_radius", C.c_float),
        ("target_density", C.c_float),
        ("pressure_scalar", C.c_float),
        ("near_pressure_scalar", C.c_float),
        ("viscosity_strength", C.c_float),
        ("reserved0", C.c_float),
        ("gravity", C.c_float * 4),
        ("ext_min", C.c_float * 4),
        ("ext_max", C.c_float * 4),
    ]


class WsSmoothingKernel(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("pow2", "pow2_der", "pow3", "pow3_der", "spikey_pow3")]


class WsDeviceCfg(C.Structure):
    _fields_ = [
        ("device", C.c_int32),
        ("flags", C.c_uint32),
        ("rank", C.c_uint32),
        ("world_size", C.c_uint32),
        ("capacity", C.c_uint32),
        ("ghost_capacity", C.c_uint32),
        ("reserved", C.c_uint32 * 2),
        ("stream", C.c_void_p),
    ]


class WsError(RuntimeError):
    def __init__(self, status, text):
        super().__init__("wsfluid status %d: %s" % (status, text))
        self.status = status


_lib = None


def lib_path():
    return _build.LIB


def load_library():
    """Load libwsfluid.so, building it first if it is stale.  Raises if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("WSFLUID_LIBRARY")  # an alternative build of the same ABI (kernel A/B runs)
    if not path:
        if _build.needs_build():
            _build.build_library()
        path = _build.LIB
    _lib = bind_library(path)
    return _lib


def bind_library(path):
    """dlopen a build of the wsfluid ABI and declare its prototypes (the product library, or the test-only
    reference-order build the tests load explicitly)."""
    L = C.CDLL(path)
    vp, u32, fp = C.c_void_p, C.c_uint32, C.POINTER(C.c_float)
    L.ws_default_params.argtypes = [C.POINTER(WsParams)]
    L.ws_get_smoothing_kernel.argtypes = [C.POINTER(WsParams), C.POINTER(WsSmoothingKernel)]
    L.ws_cube_fluid.argtypes = [u32, u32, u32, C.c_float, vp]
    L.ws_get_ext.argtypes = [vp, vp, C.c_float, vp, vp]
    L.ws_bit_sorter_stage_count.argtypes = [u32]
    L.ws_bit_sorter_stage_count.restype = u32
    L.ws_status_string.argtypes = [C.c_int]
    L.ws_status_string.restype = C.c_char_p
    L.ws_abi_version.restype = u32
    L.ws_create.argtypes = [C.POINTER(WsParams), vp, u32, C.POINTER(WsDeviceCfg), C.POINTER(vp)]
    L.ws_destroy.argtypes = [vp]
    L.ws_step.argtypes = [vp]
    L.ws_ready.argtypes = [vp, C.POINTER(C.c_int)]
    L.ws_sync.argtypes = [vp]
    L.ws_set_params.argtypes = [vp, C.POINTER(WsParams)]
    L.ws_read_positions.argtypes = [vp, vp]
    L.ws_read_particles.argtypes = [vp, vp]
    L.ws_reset.argtypes = [vp, vp]
    L.ws_pin_host_buffer.argtypes = [vp, vp, C.c_uint64]
    L.ws_read_speeds.argtypes = [vp, vp]
    L.ws_read_positions_begin.argtypes = [vp, vp]
    L.ws_read_positions_end.argtypes = [vp]
    L.ws_read_positions_view.argtypes = [vp, C.POINTER(vp)]
    L.ws_unpin_host_buffer.argtypes = [vp, vp]
    L.ws_write_particles.argtypes = [vp, vp]
    L.ws_read_sort_view.argtypes = [vp, vp, vp, vp]
    L.ws_last_error.argtypes = [vp]
    L.ws_last_error.restype = C.c_char_p
    L.ws_num_particles.argtypes = [vp]
    L.ws_num_particles.restype = u32
    L.ws_steps_done.argtypes = [vp]
    L.ws_steps_done.restype = C.c_uint64
    L.ws_kernel_name.argtypes = [u32]
    L.ws_kernel_name.restype = C.c_char_p
    L.ws_profile_read.argtypes = [vp, u32, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.ws_profile_reset.argtypes = [vp]
    L.ws_profile_select.argtypes = [vp, u32]
    L.ws_grid_dims.argtypes = [vp, vp]
    L.ws_read_stats.argtypes = [vp, vp]
    return L


# ---------------------------------------------------------------------------------------
# host-side mirror of the reference's resources (names follow the reference)
# ---------------------------------------------------------------------------------------
def default_params():
    """FluidStaticProps::default + Gravity::default + FluidContainer::default().get_ext(0.1)."""
    p = WsParams()
    load_library().ws_default_params(C.byref(p))
    return p


def get_smoothing_kernel(params):
    """FluidStaticProps::get_smoothing_kernel, src/fluid_compute.rs:55-63."""
    k = WsSmoothingKernel()
    load_library().ws_get_smoothing_kernel(C.byref(params), C.byref(k))
    return k


def cube_fluid(ni, nj, nk, particle_rad=0.1):
    """helpers::cube_fluid, src/helpers.rs:3-20."""
    out = np.empty((ni * nj * nk, 3), np.float32)
    load_library().ws_cube_fluid(ni, nj, nk, particle_rad, out.ctypes.data)
    return out


def get_ext(position, size, padding=0.1):
    """FluidContainer::get_ext, src/fluid_container.rs:42-50."""
    pos = np.asarray(position, np.float32)
    sz = np.asarray(size, np.float32)
    mn = np.zeros(4, np.float32)
    mx = np.zeros(4, np.float32)
    load_library().ws_get_ext(pos.ctypes.data, sz.ctypes.data, padding, mn.ctypes.data, mx.ctypes.data)
    return mn, mx


def make_params(container_size=(16.0, 9.0, 9.0), container_position=(0.0, 0.0, 0.0), padding=0.1, **overrides):
    p = default_params()
    mn, mx = get_ext(container_position, container_size, padding)
    for i in range(4):
        p.ext_min[i] = float(mn[i])
        p.ext_max[i] = float(mx[i])
    for k, v in overrides.items():
        if k == "gravity":
            for i in range(4):
                p.gravity[i] = float(v[i]) if i < len(v) else 0.0
        else:
            setattr(p, k, v)
    return p


class FluidWorker:
    """The role of AppComputeWorker<FluidWorker> (src/fluid_compute.rs:239-366): owns the
    device buffers, `run()` enqueues one step, `ready()` polls, `read_vec("particles")`
    returns the 80-byte records in original-id order."""

    def __init__(self, positions, params=None, device=0, profile=False, reference_order=False, ieee_division=False,
                 library=None, graph=False):
        self._L = library if library is not None else load_library()
        self._h = C.c_void_p()
        positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        self.n = positions.shape[0]
        self.params = params if params is not None else default_params()
        cfg = WsDeviceCfg()
        cfg.device = device
        cfg.flags = ((WS_FLAG_PROFILE if profile else 0) | (WS_FLAG_REFERENCE_ORDER if reference_order else 0)
                     | (WS_FLAG_IEEE_DIVISION if ieee_division else 0) | (WS_FLAG_GRAPH if graph else 0))
        st = self._L.ws_create(C.byref(self.params), positions.ctypes.data, self.n, C.byref(cfg), C.byref(self._h))
        if st != 0:
            raise WsError(st, (self._L.ws_last_error(None) or b"").decode())

    @classmethod
    def build(cls, positions, params=None, **kw):
        return cls(positions, params, **kw)

    def _check(self, st):
        if st != 0:
            raise WsError(st, (self._L.ws_last_error(self._h) or b"").decode())

    def close(self):
        if self._h:
            self._L.ws_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # AppComputeWorker::run / ready
    def run(self, steps=1):
        for _ in range(steps):
            self._check(self._L.ws_step(self._h))

    step = run

    def ready(self):
        r = C.c_int(0)
        self._check(self._L.ws_ready(self._h, C.byref(r)))
        return bool(r.value)

    def sync(self):
        self._check(self._L.ws_sync(self._h))

    # worker.write("fluid_props" | "smoothing_kernel" | "gravity", ..)
    def set_params(self, params):
        self._check(self._L.ws_set_params(self._h, C.byref(params)))
        self.params = params

    def read_positions(self):
        out = np.empty((self.n, 3), np.float32)
        self._check(self._L.ws_read_positions(self._h, out.ctypes.data))
        return out

    def read_positions_begin(self, buf):
        """Start an asynchronous id-order position readback into `buf` ((n, 3) float32, ideally pinned); steps
        enqueued afterwards overlap with the copy.  Finish with read_positions_end()."""
        assert buf.dtype == np.float32 and buf.shape == (self.n, 3) and buf.flags.c_contiguous
        self._check(self._L.ws_read_positions_begin(self._h, buf.ctypes.data))

    def read_positions_end(self):
        self._check(self._L.ws_read_positions_end(self._h))

    def read_positions_begin_owned(self):
        """The same into one of the two page-locked buffers the library owns (ws_read_positions_begin(h, NULL))."""
        self._check(self._L.ws_read_positions_begin(self._h, None))

    def read_positions_view(self, n=None):
        """(n, 3) float32 view of the library-owned buffer the last finished readback filled (no copy; valid until
        the second-next read_positions_begin_owned)."""
        p = C.c_void_p()
        self._check(self._L.ws_read_positions_view(self._h, C.byref(p)))
        n = self.n if n is None else n
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n, 3))

    def read_speeds(self):
        """|velocity| per particle in original-id order (the reference's speed colouring input)."""
        out = np.empty(self.n, np.float32)
        self._check(self._L.ws_read_speeds(self._h, out.ctypes.data))
        return out

    def read_positions_into(self, buf):
        """ws_read_positions into a caller-owned (n, 3) float32 array (pin it with pin_host_buffer for PCIe rate)."""
        assert buf.dtype == np.float32 and buf.shape == (self.n, 3) and buf.flags.c_contiguous
        self._check(self._L.ws_read_positions(self._h, buf.ctypes.data))
        return buf

    def pin_host_buffer(self, buf):
        self._check(self._L.ws_pin_host_buffer(self._h, buf.ctypes.data, buf.nbytes))

    def unpin_host_buffer(self, buf):
        self._check(self._L.ws_unpin_host_buffer(self._h, buf.ctypes.data))

    def read_vec(self, name="particles"):
        if name != "particles":
            raise KeyError(name)
        out = np.empty(self.n, PARTICLE_DTYPE)
        self._check(self._L.ws_read_particles(self._h, out.ctypes.data))
        return out

    def write_slice(self, name, data):
        if name != "particles":
            raise KeyError(name)
        data = np.ascontiguousarray(data, PARTICLE_DTYPE)
        assert data.shape[0] == self.n
        self._check(self._L.ws_write_particles(self._h, data.ctypes.data))

    def reset(self, positions):
        positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        assert positions.shape[0] == self.n
        self._check(self._L.ws_reset(self._h, positions.ctypes.data))

    def sort_view(self):
        """(particle_cell_indicies, particle_indicies, cell_offsets) as the reference holds them."""
        keys = np.empty(self.n, np.uint32)
        perm = np.empty(self.n, np.uint32)
        off = np.empty(self.n, np.uint32)
        self._check(self._L.ws_read_sort_view(self._h, keys.ctypes.data, perm.ctypes.data, off.ctypes.data))
        return keys, perm, off

    def steps_done(self):
        return int(self._L.ws_steps_done(self._h))

    def grid_dims(self):
        d = np.zeros(3, np.uint32)
        self._check(self._L.ws_grid_dims(self._h, d.ctypes.data))
        return tuple(int(x) for x in d)

    def stats(self):
        out = np.zeros(16, np.uint32)
        self._check(self._L.ws_read_stats(self._h, out.ctypes.data))
        st = {"mask_overflow": int(out[0]), "cells_merged": tuple(int(x) for x in out[1:4]), "graph_steps": int(out[4]),
              "tile_schedule": bool(out[15])}
        if out[6]:  # a slab handle: peaks of the fixed-capacity messages against their capacities
            st.update(halo_peak=int(out[5]), halo_capacity=int(out[6]), migration_peak=int(out[7]), migration_capacity=int(out[8]), far_peak=int(out[9]), far_capacity=int(out[10]),
                      migration_now=int(out[11]), halo_now=int(out[12]), far_now=int(out[13]))
        return st

    def profile(self):
        """{kernel name: (total_ms, launches)} since the last profile_reset (needs profile=True)."""
        out = {}
        for name, k in KERNEL_IDS.items():
            ms = C.c_double(0)
            cnt = C.c_uint64(0)
            self._check(self._L.ws_profile_read(self._h, k, C.byref(ms), C.byref(cnt)))
            out[name] = (ms.value, int(cnt.value))
        return out

    def profile_select(self, mask):
        self._check(self._L.ws_profile_select(self._h, mask & 0xFFFFFFFF))

    def profile_reset(self):
        self._check(self._L.ws_profile_reset(self._h))
