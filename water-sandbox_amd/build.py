"""Build libwsfluid.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwsfluid.so")

SOURCES = ["ws_kernels.hip", "ws_api.cpp", "ws_rccl.cpp", "ws_local.cpp"]
HEADERS = [os.path.join(CSRC, "ws_internal.h"), os.path.join(CSRC, "ws_devhooks.h"), os.path.join(CSRC, "ws_slab.inc"), os.path.join(ROOT, "include", "wsfluid.h")]
# Test-only build of the same sources plus the reference-order validation mode (tests/refcheck/: a literal HIP
# restatement of the reference's six WGSL passes and its host glue, used by the tests as a second, independent
# restatement).  Sources and binary live under tests/; it is built by __graft_entry__.build() and is never loaded by the
# product package.  Without -DWS_WITH_REFCHECK the product sources contain no reference-order code path at all.
REFCHECK_DIR = os.path.join(ROOT, "tests", "refcheck")
REFCHECK_LIB = os.path.join(ROOT, "tests", "libwsfluid_refcheck.so")
REFCHECK_EXTRA = [os.path.join(REFCHECK_DIR, f) for f in ("ws_refcheck.inc", "ws_refcheck_host.inc", "ws_refcheck.h")]

# -ffp-contract=off: every float op in the kernels is one IEEE binary32 op, written in the
# reference WGSL's evaluation order (no FMA contraction), see ws_kernels.hip.
HIPCC_FLAGS = [
    "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
    "-fPIC", "-shared", "-Wall", "-Wno-unused-value", "-Wno-unused-result",
]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(lib, extra=()):
    if not os.path.exists(lib):
        return True
    if os.environ.get("WS_NO_REBUILD"):  # developer jobs on the GPU box: measure the binaries that were sent, whatever the
        return False                     # sources looked like when the snapshot was taken (tools/jobs.sh)
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + list(extra)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    return _stale(LIB)


def _compile(lib, defines, verbose):
    cmd = [hipcc()] + HIPCC_FLAGS + defines + [
        "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", lib,
    ] + [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib


def build_library(force=False, verbose=False):
    """Compile the shared library if it is missing or older than its sources."""
    if not force and not needs_build():
        return LIB
    return _compile(LIB, [], verbose)


def build_refcheck_library(force=False, verbose=False):
    """Compile the TEST-ONLY library (product sources + the reference-order validation kernels)."""
    if not force and not _stale(REFCHECK_LIB, REFCHECK_EXTRA):
        return REFCHECK_LIB
    return _compile(REFCHECK_LIB, ["-DWS_WITH_REFCHECK", "-I", REFCHECK_DIR], verbose)


# TEST-ONLY / developer build of the product sources with -DWS_DEV_HOOKS: the environment hooks of csrc/ws_devhooks.h
# (WS_VARIANT, WS_CELL_BUDGET, WS_RCCL_LIBRARY, the message-size floors, stream priorities ...) exist in this build alone.
# The tests that need one of them load it explicitly (tests/util.py dev_library()); the product library reads no
# environment variable at all.
DEV_LIB = os.path.join(ROOT, "tests", "libwsfluid_dev.so")


def build_dev_library(force=False, verbose=False):
    if not force and not _stale(DEV_LIB):
        return DEV_LIB
    return _compile(DEV_LIB, ["-DWS_DEV_HOOKS"], verbose)


# TEST-ONLY: a stand-in for librccl (tests/fake_rccl/) whose "ranks" are host threads of one process on one GPU, so that
# a one-GPU box can drive csrc/ws_rccl.cpp with real peers.  Loaded only when WS_RCCL_LIBRARY names it.
FAKE_RCCL_SRC = os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.hip")
FAKE_RCCL_LIB = os.path.join(ROOT, "tests", "libfakerccl.so")


def build_fake_rccl(force=False, verbose=False):
    if not force and os.path.exists(FAKE_RCCL_LIB) and os.path.getmtime(FAKE_RCCL_LIB) >= os.path.getmtime(FAKE_RCCL_SRC):
        return FAKE_RCCL_LIB
    cmd = [hipcc(), "-x", "hip", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", FAKE_RCCL_LIB,
           FAKE_RCCL_SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return FAKE_RCCL_LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
    print(build_refcheck_library(force=True, verbose=True))
    print(build_dev_library(force=True, verbose=True))
    print(build_fake_rccl(force=True, verbose=True))
