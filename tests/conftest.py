import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The in-tree HIP C-ABI library, built on demand (hipcc cross-compiles without a GPU)."""
    from clima_amd import build, lib
    build.build()
    return lib.load()


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure only)."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def small_tables():
    from clima_amd import synthetic as S
    return S.modern_earth_tables(nw=40)
