"""The C-ABI library: loads, exports every symbol include/wsfluid.h declares, and its pure-host
functions (no device needed) agree with the oracle bit for bit.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "wsfluid.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ws_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(ws):
    lib = ws.load_library()
    names = declared_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(ws.fluid.ABI_SYMBOLS) == names  # the Python binding tracks the header
    assert lib.ws_abi_version() == 2 == ws.fluid.WS_ABI_VERSION  # round 5: ws_transport.struct_size + four callbacks, new flags


def test_struct_layouts_match_the_reference_records(ws):
    assert C.sizeof(ws.fluid.WsParams) == 80          # 8 floats + 3 vec4
    assert C.sizeof(ws.fluid.WsSmoothingKernel) == 20  # src/fluid_compute.rs:30-38
    assert ws.PARTICLE_DTYPE.itemsize == 80            # src/fluid_compute.rs:106-115
    offs = {n: ws.PARTICLE_DTYPE.fields[n][1] for n in ws.PARTICLE_DTYPE.names}
    assert offs == {"position": 0, "density": 16, "pressure": 24, "velocity": 32, "acceleration": 48,
                    "predicted_position": 64}


def test_host_functions_match_oracle_bitwise(ws, oracle):
    assert np.array_equal(ws.cube_fluid(64, 32, 32), oracle.cube_fluid(64, 32, 32))
    assert np.array_equal(ws.cube_fluid(3, 5, 7, 0.07), oracle.cube_fluid(3, 5, 7, 0.07))
    for size in [(16, 9, 9), (16, 18, 0.2), (64, 36, 36), (5.5, 3.25, 1.0)]:
        a = ws.get_ext((0.5, -1, 2), size, 0.1); b = oracle.get_ext((0.5, -1, 2), size, 0.1)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    p = ws.default_params(); q = oracle.default_props()
    for f in ("delta_time", "collision_damping", "smoothing_radius", "target_density", "pressure_scalar",
              "near_pressure_scalar", "viscosity_strength"):
        assert getattr(p, f) == getattr(q, f)
    assert list(p.gravity) == list(oracle.default_gravity())
    for h in (0.25, 0.35, 0.15, 0.3):
        p.smoothing_radius = h; q.smoothing_radius = h
        a = ws.get_smoothing_kernel(p); b = oracle.smoothing_kernel(q)
        for f in ("pow2", "pow2_der", "pow3", "pow3_der", "spikey_pow3"):
            assert getattr(a, f) == getattr(b, f), (h, f)
    lib = ws.load_library()
    for n in (1, 5, 4096, 65536, 1 << 22, 1 << 26):
        assert lib.ws_bit_sorter_stage_count(n) == len(oracle.bit_sorter_stages(n))


def test_uniform_cloud_generator_matches_oracle(ws, oracle):
    pos, params = ws.workloads.make_workload("c1", "cloud")
    ref = oracle.uniform_cloud(pos.shape[0], ws.workloads.cloud_seed("c1"), list(params.ext_min), list(params.ext_max))
    assert np.array_equal(pos, ref)
    assert ws.workloads.block_for(4194304) == (256, 128, 128)
    assert ws.workloads.container_for_block((256, 128, 128)) == pytest.approx((64.0, 36.0, 36.0))


def test_status_strings_and_kernel_names(ws):
    lib = ws.load_library()
    assert lib.ws_status_string(0) == b"ok"
    assert b"device" in lib.ws_status_string(2)
    assert lib.ws_kernel_name(4) == b"force_integrate_bin"


def test_create_rejects_bad_arguments_without_touching_a_device(ws):
    lib = ws.load_library()
    h = C.c_void_p()
    p = ws.default_params()
    pos = np.zeros((4, 3), np.float32)
    assert lib.ws_create(C.byref(p), None, 4, None, C.byref(h)) == 1 and not h
    assert lib.ws_create(C.byref(p), pos.ctypes.data, 0, None, C.byref(h)) == 1
    p.smoothing_radius = 0.0
    assert lib.ws_create(C.byref(p), pos.ctypes.data, 4, None, C.byref(h)) == 1
    assert b"smoothing_radius" in lib.ws_last_error(None)
    assert lib.ws_step(None) == 1 and lib.ws_destroy(None) == 0


def test_slab_create_wants_a_complete_transport(ws):
    """A slab with peers needs all three callbacks (include/wsfluid.h, ws_transport): a table without alltoall_dev -- the
    per-destination far messages and the status words -- is refused before any device is touched; a world of one may
    leave it out (it makes no transport call at all) and gets as far as the device check."""
    lib = ws.load_library()
    T = ws.slab
    lib.ws_slab_create.argtypes = [C.POINTER(ws.fluid.WsParams), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                   C.POINTER(ws.fluid.WsDeviceCfg), C.POINTER(T.WsTransport), C.POINTER(C.c_void_p)]
    noop3 = T.SENDRECV_T(lambda *a: 0)
    noop = T.ALLGATHER_DEV_T(lambda *a: 0)
    p = ws.default_params()
    pos, ids = np.zeros((4, 3), np.float32), np.arange(4, dtype=np.uint32)
    cfg = ws.fluid.WsDeviceCfg()
    cfg.device, cfg.rank, cfg.world_size = 0, 0, 2
    h = C.c_void_p()
    incomplete = T.WsTransport(C.sizeof(T.WsTransport), None, noop3, noop, T.ALLGATHER_DEV_T())
    # a table from a host built against ABI version 1 (three callbacks, no struct_size: whatever lies there is smaller
    # than this library's struct) is refused before any member behind it is looked at
    short = T.WsTransport(C.sizeof(T.WsTransport) - 8, None, noop3, noop, noop)
    assert lib.ws_slab_create(C.byref(p), pos.ctypes.data, ids.ctypes.data, 4, 8, C.byref(cfg), C.byref(short), C.byref(h)) == 1
    assert b"struct_size" in lib.ws_last_error(None) and not h
    assert lib.ws_slab_create(C.byref(p), pos.ctypes.data, ids.ctypes.data, 4, 8, C.byref(cfg), C.byref(incomplete), C.byref(h)) == 1
    assert b"alltoall_dev" in lib.ws_last_error(None) and not h
    cfg.world_size = 1
    st = lib.ws_slab_create(C.byref(p), pos.ctypes.data, ids.ctypes.data, 4, 4, C.byref(cfg), C.byref(incomplete), C.byref(h))
    assert st in (0, 2)  # created (GPU box) or WS_ERR_NO_DEVICE (here): past the argument checks either way
    if st == 0:
        lib.ws_destroy.argtypes = [C.c_void_p]
        lib.ws_destroy(h)


def test_no_cpu_fallback(ws):
    """Without a GPU the product must fail loudly, never fall back to a CPU path."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ws.WsError) as e:
        ws.FluidWorker(ws.cube_fluid(4, 4, 4))
    assert e.value.status == 2  # WS_ERR_NO_DEVICE


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "water-sandbox_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("oracle's", "").lower() or f == "README", (dirpath, f)


def test_product_library_does_not_contain_the_reference_order_port(ws):
    """The literal HIP restatement of the reference's WGSL passes (kr_* kernels) is test infrastructure: it is
    compiled into tests/libwsfluid_refcheck.so only, and the shipped library refuses WS_FLAG_REFERENCE_ORDER."""
    product = open(ws.build.build_library(), "rb").read()
    check = open(ws.build.build_refcheck_library(), "rb").read()
    for name in (b"kr_bitonic", b"kr_density", b"kr_force", b"kr_integrate"):
        assert name not in product, name
        assert name in check, name
    assert b"k_force_listed" in product and b"k_density_listed" in product
    cfg = ws.fluid.WsDeviceCfg()
    cfg.flags = ws.fluid.WS_FLAG_REFERENCE_ORDER
    h = C.c_void_p()
    p = ws.default_params()
    pos = np.zeros((4, 3), np.float32)
    assert ws.load_library().ws_create(C.byref(p), pos.ctypes.data, 4, C.byref(cfg), C.byref(h)) == 6  # WS_ERR_UNSUPPORTED
    assert not h


def test_header_is_plain_c_and_binds_without_hip(ws, tmp_path):
    """include/wsfluid.h must be consumable by a C compiler (bindgen / cgo / a C host): compile a C11 translation
    unit against it with gcc -pedantic, link it to libwsfluid.so and run the host-only entry points."""
    import subprocess

    src = tmp_path / "host.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "wsfluid.h"
int main(void) {
    ws_params p; ws_smoothing_kernel k; float lattice[3 * 8]; float mn[4], mx[4];
    const float position[3] = {0.f, 0.f, 0.f}, size[3] = {16.f, 9.f, 9.f};
    if (sizeof(ws_particle80) != 80 || sizeof(ws_params) != 80) return 2;
    if (ws_abi_version() != WS_ABI_VERSION) return 3;
    if (ws_default_params(&p) != WS_OK || ws_get_smoothing_kernel(&p, &k) != WS_OK) return 4;
    if (ws_cube_fluid(2, 2, 2, 0.1f, lattice) != WS_OK || ws_get_ext(position, size, 0.1f, mn, mx) != WS_OK) return 5;
    if (ws_bit_sorter_stage_count(4096) != 78) return 6;
    if (ws_step(NULL) != WS_ERR_INVALID_ARG) return 7;
    printf("%.6f %.3f %.1f %s\n", (double)p.delta_time, (double)k.pow2, (double)mx[0], ws_status_string(WS_ERR_NO_DEVICE));
    return 0;
}
''')
    exe = tmp_path / "host"
    libdir = os.path.dirname(ws.build.build_library())
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe), "-L", libdir, "-lwsfluid", "-Wl,-rpath," + libdir])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.split()[:3] == ["0.016667", "2444.620", "7.9"]


def test_no_smoothing_radius_runs_out_of_cell_tables(ws):
    """Host-only (ws_slab_assign derives the global grid): every radius a HUD key press can reach at any BASELINE
    config yields a grid (merged cells beyond the budget), never WS_ERR_OUT_OF_MEMORY; a radius so small that cell
    coordinates stop being exact in f32 is an invalid argument, not an allocation failure."""
    lib = ws.load_library()
    lib.ws_slab_assign.argtypes = [C.POINTER(ws.fluid.WsParams), C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    pos = np.zeros((4, 3), np.float32)
    out = np.zeros(4, np.uint32)
    for name in ("c3", "c4", "c5"):
        for h in (0.35, 0.25, 0.15, 0.05, 0.01):
            p = ws.make_params(container_size=ws.workloads.CONFIGS[name][1], smoothing_radius=h)
            assert lib.ws_slab_assign(C.byref(p), pos.ctypes.data, 4, 2, out.ctypes.data) == 0, (name, h)
    p = ws.make_params(container_size=ws.workloads.CONFIGS["c5"][1], smoothing_radius=1e-6)
    assert lib.ws_slab_assign(C.byref(p), pos.ctypes.data, 4, 2, out.ctypes.data) == 1  # WS_ERR_INVALID_ARG


def test_balanced_cuts_are_monotone_complete_and_balanced(ws):
    """Host-only: the re-cut rule of ws_slab_rebalance.  For any histogram the cuts start at 0, end at nx, increase
    strictly (every slab keeps a layer), and no slab exceeds its fair share by more than the layer that crosses it."""
    lib = ws.load_library()
    lib.ws_slab_balanced_cuts.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    rng = np.random.default_rng(7)
    for nx, world in ((68, 3), (68, 8), (260, 8), (9, 8), (5, 5), (1028, 8)):
        for shape in ("flat", "piled", "spike", "empty-left"):
            hist = {"flat": np.full(nx, 1000), "piled": (np.arange(nx) ** 3) % 100000 + 1,
                    "spike": np.where(np.arange(nx) == nx // 2, 10 ** 6, 3),
                    "empty-left": np.where(np.arange(nx) < nx // 2, 0, rng.integers(1, 5000, nx))}[shape].astype(np.uint32)
            cuts = np.zeros(world + 1, np.uint32)
            assert lib.ws_slab_balanced_cuts(hist.ctypes.data, nx, world, cuts.ctypes.data) == 0
            assert cuts[0] == 0 and cuts[-1] == nx and np.all(np.diff(cuts.astype(np.int64)) >= 1), (nx, world, shape, cuts)
            total, share = int(hist.sum()), int(hist.sum()) / world
            for r in range(world):
                owned = int(hist[cuts[r]:cuts[r + 1]].sum())
                if cuts[r + 1] - cuts[r] > 1:   # a slab of several layers never overshoots by more than its last layer
                    assert owned <= share + int(hist[cuts[r]:cuts[r + 1]].max()) + 1, (nx, world, shape, r, owned, share)
    assert lib.ws_slab_balanced_cuts(hist.ctypes.data, 4, 5, cuts.ctypes.data) == 1   # more slabs than layers
