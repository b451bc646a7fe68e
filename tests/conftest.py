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
    from oracle import ref_oracle
    ref_oracle.lib()
    return ref_oracle


@pytest.fixture(scope="session")
def bq():
    """The product's host interface; building the library is part of the CPU check (hipcc cross-compiles)."""
    import __graft_entry__ as g
    g.build()
    from tsqr_gpu_amd import blockqr
    return blockqr
