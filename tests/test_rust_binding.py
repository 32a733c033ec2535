"""The Rust side cannot be compiled in this image (no rustc / cargo), so its FFI surface is checked mechanically
instead: the `extern "C"` block and `#[repr(C)] WsParams` of rust/fluid_compute.rs -- and the copy of the block printed
in INTEGRATION.md -- are parsed and compared with include/wsfluid.h (and with the ctypes structs the tests drive the
library through): names, arity, pointer / integer widths, const-ness of pointers, struct field order and types."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def c_prototypes():
    """{name: (return class, [param classes])} of every function include/wsfluid.h declares."""
    text = _strip_c_comments(open(os.path.join(ROOT, "include", "wsfluid.h")).read())
    text = re.sub(r"typedef struct ws_transport \{.*?\} ws_transport;", "", text, flags=re.S)  # callback members
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(ws_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        plist = [] if params in ("", "void") else [_c_class(p) for p in _split_params(params)]
        protos[name] = (_c_class(ret + " x") if ret != "void" else "void", plist)
    return protos


def _split_params(params):
    return [p.strip() for p in params.split(",")]


def _c_class(decl):
    """'const float *pos_xyz' -> 'ptr const'; 'uint32_t n' -> 'u32'; arrays decay to pointers."""
    decl = decl.strip()
    if "[" in decl or "*" in decl:
        before_star = decl.split("*")[0] if "*" in decl else decl.split("[")[0]
        return "ptr const" if re.search(r"\bconst\b", before_star) else "ptr mut"
    ty = " ".join(decl.split()[:-1])
    return {"uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "int": "i32", "float": "f32", "ws_status": "i32",
            "double": "f64"}[ty]


def _rust_class(ty):
    ty = ty.strip()
    if ty.startswith("*const"):
        return "ptr const"
    if ty.startswith("*mut"):
        return "ptr mut"
    return {"u32": "u32", "i32": "i32", "u64": "u64", "c_int": "i32", "f32": "f32", "f64": "f64"}[ty]


def rust_extern_block(text):
    """{name: (return class, [param classes])} of the first `extern "C" { ... }` block in `text`."""
    block = re.search(r'extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S).group(1)
    protos = {}
    for m in re.finditer(r"fn\s+(ws_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        name, params, ret = m.group(1), m.group(2).strip(), (m.group(3) or "").strip()
        plist = [_rust_class(p.split(":", 1)[1]) for p in params.split(",") if p.strip()]
        protos[name] = (_rust_class(ret) if ret else "void", plist)
    return protos


def c_struct_fields(name):
    text = _strip_c_comments(open(os.path.join(ROOT, "include", "wsfluid.h")).read())
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S).group(1)
    out = []
    for m in re.finditer(r"(float|uint32_t|int32_t)\s+([a-z_0-9]+)(?:\[(\d+)\])?\s*;", body):
        out.append((m.group(2), m.group(1), int(m.group(3) or 1)))
    return out


def rust_struct_fields(text, name):
    m = re.search(r"#\[repr\(C\)\][^\n]*\n(?:#\[[^\n]*\]\n)*pub struct %s \{(.*?)\n\}" % name, text, flags=re.S)
    assert m, "no #[repr(C)] pub struct %s" % name
    out = []
    for f in re.finditer(r"pub\s+([a-z_0-9]+)\s*:\s*(?:\[(f32|u32|i32);\s*(\d+)\]|(f32|u32|i32))\s*,", m.group(1)):
        out.append((f.group(1), f.group(2) or f.group(4), int(f.group(3) or 1)))
    return out


RUST = open(os.path.join(ROOT, "rust", "fluid_compute.rs")).read()
INTEGRATION = open(os.path.join(ROOT, "INTEGRATION.md")).read()


@pytest.mark.parametrize("where,text", [("rust/fluid_compute.rs", RUST), ("INTEGRATION.md", INTEGRATION)])
def test_extern_block_agrees_with_the_header(where, text):
    header = c_prototypes()
    block = rust_extern_block(text)
    assert len(block) >= 11, (where, sorted(block))
    for name, (ret, params) in block.items():
        assert name in header, "%s binds %s, which include/wsfluid.h does not declare" % (where, name)
        h_ret, h_params = header[name]
        assert ret == h_ret, "%s %s: return %s, header %s" % (where, name, ret, h_ret)
        assert len(params) == len(h_params), "%s %s: %d parameters, header %d" % (where, name, len(params), len(h_params))
        for k, (a, b) in enumerate(zip(params, h_params)):
            assert a == b, "%s %s parameter %d: %s, header %s" % (where, name, k, a, b)


def test_the_two_copies_of_the_block_are_the_same():
    assert rust_extern_block(RUST) == rust_extern_block(INTEGRATION)


def test_ws_params_layout_agrees_in_rust_c_and_ctypes(ws):
    c_fields = c_struct_fields("ws_params")
    r_fields = rust_struct_fields(RUST, "WsParams")
    conv = {"float": "f32", "uint32_t": "u32", "int32_t": "i32"}
    assert [(n, conv[t], k) for n, t, k in c_fields] == r_fields
    def elem(t):
        return t._type_ if hasattr(t, "_length_") else t

    py = [(n, {C.c_float: "f32"}[elem(t)], getattr(t, "_length_", 1)) for n, t in ws.fluid.WsParams._fields_]
    assert py == r_fields
    assert sum(k for _, _, k in r_fields) * 4 == C.sizeof(ws.fluid.WsParams) == 80


def test_status_constants_of_the_shim_match_the_header():
    text = _strip_c_comments(open(os.path.join(ROOT, "include", "wsfluid.h")).read())
    enum = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"\b(WS_[A-Z_]+)\s*=\s*(\d+)", text))
    for m in re.finditer(r"const (WS_[A-Z_]+): c_int = (\d+);", RUST):
        assert enum[m.group(1)] == int(m.group(2)), m.group(1)
    # a rejected parameter set must not take the app down (the shim used to panic on it)
    body = RUST[RUST.index("fn update("):RUST.index("fn despawn_liquid(")]
    assert "WS_ERR_OUT_OF_MEMORY" in body and "WS_ERR_INVALID_ARG" in body
    assert 'check(h, unsafe { ws_set_params' not in body


def test_the_shim_checks_the_abi_version_the_header_declares():
    """the struct layouts changed under an unchanged version once.  The shim carries the version it was
    written against, compares it with ws_abi_version() before anything crosses the boundary, and this test ties the
    constant to the header."""
    header = int(re.search(r"#define WS_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "wsfluid.h")).read()).group(1))
    shim = int(re.search(r"const WS_ABI_VERSION: u32 = (\d+);", RUST).group(1))
    assert shim == header
    build = RUST[RUST.index("impl Plugin for FluidComputeWorkerPlugin") if "impl Plugin for FluidComputeWorkerPlugin" in RUST else 0:]
    assert build.index("ws_abi_version()") < build.index("ws_create(&params")
    hpp = open(os.path.join(ROOT, "water-sandbox_amd", "host", "fluid_compute.hpp")).read()
    assert hpp.count("check_abi();") >= 2 and "ws_abi_version() != WS_ABI_VERSION" in hpp


def test_header_parser_sees_every_declared_function(ws):
    assert sorted(c_prototypes()) == sorted(ws.fluid.ABI_SYMBOLS)


def _call_sites(text, name):
    """Argument counts of every call `name(...)` in `text` outside the extern block (top-level commas + 1)."""
    out = []
    for m in re.finditer(r"\b%s\s*\(" % re.escape(name), text):
        i, depth, commas, empty = m.end(), 1, 0, True
        while depth and i < len(text):
            c = text[i]
            if c in "([{":
                depth += 1
            elif c in ")]}":
                depth -= 1
            elif c == "," and depth == 1:
                commas += 1
            if depth and not c.isspace():
                empty = False
            i += 1
        out.append(0 if empty else commas + 1)
    return out


def test_every_call_site_in_the_shim_passes_as_many_arguments_as_the_prototype_takes():
    block = rust_extern_block(RUST)
    body = RUST[RUST.index('extern "C"'):]
    body = body[body.index("\n}") + 2:]          # everything after the extern block
    body = re.sub(r"//[^\n]*", "", body)          # comments mention the functions too
    called = 0
    for name, (_, params) in block.items():
        for argc in _call_sites(body, name):
            assert argc == len(params), "%s called with %d arguments, declared with %d" % (name, argc, len(params))
            called += 1
    assert called >= len(block)                    # every bound function is used
    for name in re.findall(r"\b(ws_[a-z0-9_]+)\s*\(", body):
        assert name in block, "the shim calls %s without binding it" % name


def test_the_shim_keeps_the_reference_plugin_surface_and_system_sets():
    """Names and schedule placement the reference defines (src/fluid_compute.rs:369-435,
    src/schedule.rs:24-36) and the shim must keep so that main.rs / hud.rs need no change."""
    for needle in ("pub struct FluidPlugin", "pub struct FluidComputePlugin", "pub struct FluidStaticProps",
                   "pub struct FluidParticlesInitial", "struct FluidParticleLabel(usize)", "fn setup(", "fn update(",
                   "fn despawn_liquid(",
                   "add_systems(OnExit(GameState::Menu), setup)",
                   "add_systems(Update, update.in_set(InGameSet::EntityUpdates))",
                   "add_systems(Update, despawn_liquid.in_set(InGameSet::DespawnEntities))",
                   "in_set(ShaderPhysicsSet::Prepare)", "in_set(ShaderPhysicsSet::Pass)",
                   "init_resource::<FluidStaticProps>()", "init_resource::<FluidParticlesInitial>()"):
        assert needle in RUST, needle
    # the defaults of FluidStaticProps (src/fluid_compute.rs:20-27,:67-79)
    for field, value in (("collision_damping", "0.95"), ("smoothing_radius", "0.25"), ("target_density", "10."),
                         ("pressure_scalar", "22."), ("near_pressure_scalar", "2."), ("viscosity_strength", "0.1")):
        assert re.search(r"%s:\s*%s\s*," % (field, re.escape(value)), RUST), field
    assert RUST.count("{") == RUST.count("}") and RUST.count("(") == RUST.count(")")
