import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _scratch_cwd(tmp_path_factory):
    """the compiled reference (oracle/_ref) writes its warnings to Fortran unit 66, which nobody opened: a file `fort.66` in the
    current directory.  The tests run from a scratch directory so that it does not land in the source tree (every path the tests
    use is absolute, built from ROOT)."""
    d = tmp_path_factory.mktemp("cwd")
    old = os.getcwd()
    os.chdir(d)
    yield
    os.chdir(old)


@pytest.fixture(scope="session")
def orc():
    """the CPU oracle (test infrastructure): oracle/liboracle.so, built on demand"""
    from oracle.pyoracle import Oracle, build
    build()
    return Oracle()


@pytest.fixture(scope="session")
def ctx():
    """the product context; only -m gpu tests use it.  No fallback: fails loudly without a GPU."""
    import dazimsurftomo_amd as dz
    dz.build()
    c = dz.Context(0)
    yield c
    c.close()
