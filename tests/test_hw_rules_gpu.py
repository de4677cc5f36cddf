"""Hardware rules the kernels rely on beyond what the compiler documents, pinned by stand-alone probes (tools/microbench):
* ds_read_b64_tr_b16's lane map — the prompt attention reads V fragments through it (attn_prefill.hip);
* the range check of raw buffer accesses covers voffset + soffset — the 128-row prompt GEMM switches DMA pieces off through the
  scalar offset, the prompt attention bounds its K / V reads by the descriptor's end (LLVM documents voffset only).
A probe that fails here means the kernels above are wrong on this hardware / driver, whatever their own tests say."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bin")


@pytest.mark.parametrize("probe,expect", [("tr_read_probe", "lane map as assumed: yes"), ("soffset_bounds_probe", "covers voffset + soffset: yes")])
def test_probe(probe, expect):
    exe = os.path.join(BIN, probe)
    if not os.path.exists(exe):
        pytest.fail(f"{exe} is not built (make -C tools/microbench); it ships with the snapshot")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and expect in out.stdout, out.stdout + out.stderr
