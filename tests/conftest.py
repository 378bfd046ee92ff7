import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    from mbn_amd import import_package
    p = import_package()
    if not (os.path.exists(p.LIB_PATH) and os.path.exists(p.HOST_LIB_PATH)):
        p.build()
    return p


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def ctx(pkg):
    c = pkg.Context(0)     # raises (no fallback) when there is no HIP device
    yield c
    c.close()
