import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    so = os.path.join(ROOT, "picha_amd", "libpicha_hip.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()


@pytest.fixture(scope="session", autouse=True)
def built_libraries():
    _ensure_built()
    import oracle
    oracle.build()
    yield
