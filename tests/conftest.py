import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


HOOKS_SO = os.path.join(ROOT, "xpoly_amd", "libxpoly_amd_hooks.so")


def hooks_env(**over):
    """Environment of a child process that loads the -DXPG_TEST_HOOKS build of the library (xpoly_amd/build.py build_hooks):
    fault injection, forced routes and the lab's A/B knobs exist only there; the product library ignores them."""
    env = dict(os.environ, XPG_SO_PATH=HOOKS_SO)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.update(over)
    return env


def needs_hooks(fn):
    """A test that sets a hook-only switch: it runs in a child pytest process on the hooks build (the parent process has the
    product library loaded and keeps it); the parent asserts the child's verdict."""
    import functools
    import subprocess

    @functools.wraps(fn)
    def wrapper(*a, **k):
        if os.environ.get("XPG_SO_PATH") == HOOKS_SO:
            return fn(*a, **k)
        assert os.path.exists(HOOKS_SO), "%s is missing: python -m xpoly_amd.build" % HOOKS_SO
        node = os.environ["PYTEST_CURRENT_TEST"].rsplit(" ", 1)[0]
        r = subprocess.run([sys.executable, "-m", "pytest", node, "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"], cwd=ROOT,
                           env=hooks_env(), capture_output=True, text=True, timeout=3000)
        assert r.returncode == 0, "on the hooks build:\n" + r.stdout[-4000:] + r.stderr[-2000:]
    return wrapper


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
