import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O


@pytest.fixture(scope="session")
def ws():
    import water_sandbox_amd as ws

    ws.load_library()
    return ws


@pytest.fixture(scope="session")
def refcheck(ws):
    """The TEST-ONLY build of the library that also contains the reference-order validation kernels
    (tests/libwsfluid_refcheck.so; the product library refuses WS_FLAG_REFERENCE_ORDER)."""
    return ws.fluid.bind_library(ws.build.build_refcheck_library())
