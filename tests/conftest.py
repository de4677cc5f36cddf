import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "fuzz: the long form of the seeded sweeps (minutes; run with -m fuzz, skipped otherwise)")


def pytest_collection_modifyitems(config, items):
    # the long sweeps carry both markers: `-m gpu` (the round-end suite) must not drag them in; only an expression naming `fuzz` does
    if "fuzz" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="long sweep: run with -m fuzz")
    for it in items:
        if "fuzz" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def zg():
    """The C-ABI library, initialised on device 0.  Fails loudly when the HIP build is missing."""
    from zig_gpt2_amd import _lib

    lib = _lib.load()
    _lib.check(lib.zg_init(0))
    yield lib
