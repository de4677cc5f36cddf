import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def zg():
    """The C-ABI library, initialised on device 0.  Fails loudly when the HIP build is missing."""
    from zig_gpt2_amd import _lib

    lib = _lib.load()
    _lib.check(lib.zg_init(0))
    yield lib
