import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "piecewise-icp_amd"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import _oracle
    _oracle.lib()
    return _oracle


@pytest.fixture(scope="session")
def ctx():
    """HIP context on device 0.  Fails (does not skip) if the native library or the GPU is missing:
    the product has no fallback path."""
    import pwicp_amd as P
    c = P.Context(0)
    yield c
    c.close()
