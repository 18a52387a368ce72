import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # make sure the in-tree libraries exist (no-op when up to date; hipcc cross-compiles without a GPU)
    import __graft_entry__
    __graft_entry__.build()


@pytest.fixture(scope="session")
def sf():
    from util import sf as _sf
    return _sf


@pytest.fixture(scope="session")
def oracle():
    import oracle as _o
    _o.blas_init("auto", threads=4)
    return _o
