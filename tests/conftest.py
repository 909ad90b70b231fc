import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
