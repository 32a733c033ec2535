"""Build libwsfluid.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwsfluid.so")

SOURCES = ["ws_kernels.hip", "ws_api.cpp", "ws_rccl.cpp"]
HEADERS = [os.path.join(CSRC, "ws_internal.h"), os.path.join(CSRC, "ws_slab.inc"), os.path.join(ROOT, "include", "wsfluid.h")]

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


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile the shared library if it is missing or older than its sources."""
    if not force and not needs_build():
        return LIB
    cmd = [hipcc()] + HIPCC_FLAGS + [
        "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", LIB,
    ] + [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
