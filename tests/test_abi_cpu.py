"""CPU-only checks of the drop-in boundary: the C-ABI library is built in-tree, loads, and exports
every symbol include/zgpt2.h declares with a binding in zig_gpt2_amd/_lib.py.  No compute calls."""
import ctypes
import os
import re

import pytest

from zig_gpt2_amd import _lib


@pytest.fixture(scope="module")
def header_symbols():
    text = open(_lib.HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zg_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built():
    assert os.path.exists(_lib.SO_PATH), "run __graft_entry__.build() first"


def test_every_header_symbol_is_exported_and_bound(header_symbols):
    assert len(header_symbols) >= 25
    lib = ctypes.CDLL(_lib.SO_PATH)
    for name in header_symbols:
        assert hasattr(lib, name), f"{name} declared in include/zgpt2.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in _lib.py"
    extra = set(_lib.SIGNATURES) - set(header_symbols)
    assert not extra, f"bound but not declared in the header: {extra}"


def test_uninitialised_calls_fail_loudly():
    lib = _lib.load()
    # No GPU here: zg_init must fail with a HIP error, and compute entry points must refuse to run
    # rather than fall back to anything.
    import numpy as np

    x = np.zeros(8, np.float32)
    try:
        import torch

        if torch.cuda.is_available():
            pytest.skip("GPU present: covered by the gpu tests")
    except ImportError:
        pass
    rc = lib.zg_gelu(x.ctypes.data, x.size)
    assert rc == -1  # ZG_ERR_NOT_INITIALIZED
    assert b"zg_init" in lib.zg_last_error()


def test_product_package_does_not_import_the_oracle():
    import sys
    import subprocess

    code = "import sys; import zig_gpt2_amd, zig_gpt2_amd.ops, zig_gpt2_amd.gpt; assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'"
    subprocess.check_call([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(_lib.HEADER)))
    root = os.path.dirname(_lib.SO_PATH.rsplit("/lib/", 1)[0] + "/x")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".zig")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in text and "from oracle" not in text and "zgpt2_oracle" not in text, f


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/zgpt2.h must compile as C99 with nothing but the standard headers (no HIP, no C++)."""
    import subprocess

    src = tmp_path / "hdr.c"
    src.write_text('#include "%s"\nint main(void) { zg_gpt_options o; zg_gpt_config c; o.share_weights_with = 0; o.own_stream = 1; '
                   'o.stream_priority = 0; c.n_embed = 768; return (int)(sizeof(o) + sizeof(c) + ZG_N_BLOCK_SLOTS + ZG_DIST_ID_BYTES) == 0; }\n' % _lib.HEADER)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-c", str(src), "-o", str(tmp_path / "hdr.o")])
