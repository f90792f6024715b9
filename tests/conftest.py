import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libxpoly_ref.so (the real reference build)")


@pytest.fixture(scope="session")
def port():
    from oracle.checker import Port
    return Port()


@pytest.fixture(scope="session")
def ref():
    from oracle.checker import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref/libxpoly_ref.so not built (needs /root/reference)")
    return Ref()


@pytest.fixture(scope="session")
def ctx():
    import xpoly_amd
    c = xpoly_amd.Context(0)
    yield c
    c.close()
