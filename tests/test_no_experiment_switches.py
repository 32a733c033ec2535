"""The shipped kernels carry no experiment switches (VERDICT r3 hygiene): the ablations ("wrong results" by design), the
trip-count instrumentation and the alternatives that were measured and not adopted live as patches under
tools/ab/patches/, applied to a scratch copy by tools/ab_build.sh.  One stray -D can therefore not ship a library that
computes garbage.  The patches must keep applying to the current sources, or they are dead text."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "water-sandbox_amd", "csrc")
EXPERIMENT_MACROS = ("ND_ABLATE", "WS_EXP_COUNT", "ND_FULL_STORE", "NF_PREFETCH_RUN", "WS_XCD_TILES", "g_exp", "ws_exp_read")
PATCHES = sorted(f[:-6] for f in os.listdir(os.path.join(ROOT, "tools", "ab", "patches")) if f.endswith(".patch"))


def test_product_sources_have_no_experiment_switches():
    for f in os.listdir(CSRC):
        text = open(os.path.join(CSRC, f), errors="replace").read()
        for m in EXPERIMENT_MACROS:
            assert m not in text, "%s mentions %s" % (f, m)
    # the only conditional compilation left in the kernels: six tuning defaults and the test-only build's include
    conds = re.findall(r"^#\s*(?:if|ifdef|ifndef)\s+(\w+)", open(os.path.join(CSRC, "ws_kernels.hip")).read(), flags=re.M)
    assert sorted(conds) == sorted(["ND_P", "ND_K", "ND_MASK_WORDS", "NF_WORDSYNC_MAX", "NF_WORDSYNC_MAX_BIG", "NF_WORDSYNC_BIG_TILES", "NF_WORDSYNC_MAX_IEEE", "NF_WORDSYNC_MAX_TINY", "NF_WORDSYNC_TINY_TILES", "NF_P", "NF_P_SMALL", "NF_SMALL_BELOW", "WS_REORDER_BLOCK", "WS_WITH_REFCHECK"]), conds


# every environment hook the sources know (csrc/ws_devhooks.h: compiled in with -DWS_DEV_HOOKS only)
DEV_HOOKS = ("WS_VARIANT", "WS_CELL_BUDGET", "WS_COPY_STREAM_PRIORITY", "WS_RCCL_LIBRARY", "WS_RCCL_SINGLE_COMM",
             "WS_SLAB_EXACT_ONE_RANK", "WS_SLAB_FLOOR_MIGRATION", "WS_SLAB_FLOOR_FAR", "WS_SLAB_FLOOR_HALO", "WS_SLAB_COMM_PRIORITY",
             "WS_TILE_SCHEDULE", "WS_SCHED_CLASSES", "WS_SCHED_GROUP", "WS_FAIL_REGRID")


def test_product_sources_read_the_environment_through_the_dev_hook_macro_only():
    """No getenv in the product sources: whatever looks at the environment goes through WS_DEV_ENV (csrc/ws_devhooks.h),
    which is `nullptr` unless the build defines WS_DEV_HOOKS (tests/libwsfluid_dev.so, tools/ab_build.sh)."""
    hooks = set()
    for f in os.listdir(CSRC):
        text = open(os.path.join(CSRC, f), errors="replace").read()
        if f != "ws_devhooks.h":
            assert "getenv" not in text, f
        hooks |= set(re.findall(r'WS_DEV_ENV\("(\w+)"\)', text))
    assert hooks == set(DEV_HOOKS), sorted(hooks ^ set(DEV_HOOKS))
    for f in ("build.py", "fluid.py", "slab.py", "workloads.py", "__init__.py"):
        text = open(os.path.join(ROOT, "water-sandbox_amd", f)).read()
        for name in DEV_HOOKS:
            assert name not in text or f == "build.py", "%s mentions %s" % (f, name)


def test_product_binary_contains_no_environment_hook():
    """The shipped libwsfluid.so does not contain the NAME of any hook: no environment variable can
    steer it -- in particular none can put another collective library under a production process (WS_RCCL_LIBRARY)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("ws_build", os.path.join(ROOT, "water-sandbox_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    lib = b.build_library()
    blob = open(lib, "rb").read()
    for name in DEV_HOOKS:
        assert name.encode() not in blob, name
    assert b"getenv" not in blob
    dev = open(b.build_dev_library(), "rb").read()  # ... and the developer build does (the test means something)
    assert all(name.encode() in dev for name in DEV_HOOKS) and b"getenv" in dev


def test_product_build_defines_nothing():
    import importlib.util

    spec = importlib.util.spec_from_file_location("ws_build", os.path.join(ROOT, "water-sandbox_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert not [f for f in b.HIPCC_FLAGS if f.startswith("-D")], b.HIPCC_FLAGS
    src = open(os.path.join(ROOT, "water-sandbox_amd", "build.py")).read()
    assert re.findall(r'"-D(\w+)', src) == ["WS_WITH_REFCHECK", "WS_DEV_HOOKS"]  # the two test-only libraries alone


@pytest.mark.skipif(shutil.which("patch") is None, reason="no patch(1)")
@pytest.mark.parametrize("name", PATCHES)
def test_experiment_patches_still_apply(name):
    with tempfile.TemporaryDirectory() as work:
        os.makedirs(os.path.join(work, "water-sandbox_amd"))
        shutil.copytree(CSRC, os.path.join(work, "water-sandbox_amd", "csrc"))
        with open(os.path.join(ROOT, "tools", "ab", "patches", name + ".patch")) as p:
            out = subprocess.run(["patch", "-p1", "--dry-run", "-d", work], stdin=p, capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
