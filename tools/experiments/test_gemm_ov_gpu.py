"""EXPERIMENT (not part of the product test run): gemm_ov_kernel — the epilogue of a tile under the next tile's main loop, measured
slower than gemm_s4 at two tiles per workgroup (profiles/NOTEBOOK.md) — computes the same fp32 values in the same order: its bf16
output must be bitwise that of the product's gemm_s4, also with several tiles per workgroup.
    make -C tools/experiments && python -m pytest tools/experiments/test_gemm_ov_gpu.py -m "gpu and experiment" """
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from zig_gpt2_amd import _lib, synth  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.experiment]
EXP = os.path.join(ROOT, "zig_gpt2_amd", "lib", "libzgpt2_exp.so")


@pytest.mark.parametrize("m,n,k,wgs", [(256, 192, 128, 0), (1100, 776, 320, 3), (2048, 1536, 768, 5)])
@pytest.mark.parametrize("gelu", [False, True])
def test_overlapped_epilogue_gemm_equals_the_four_wave_kernel(m, n, k, wgs, gelu, monkeypatch):
    import torch

    if not os.path.exists(EXP):
        pytest.skip("libzgpt2_exp.so not built (make -C tools/experiments)")
    zg = _lib.load()
    _lib.check(zg.zg_init(0))
    exp = C.CDLL(EXP)
    if wgs:
        monkeypatch.setenv("ZGPT2_GEMM_WGS", str(wgs))
    monkeypatch.setenv("ZGPT2_GEMM_KERNEL", "s4")
    a = torch.from_numpy(synth.to_bf16_bits(synth.fill_normal(41, m * k, 0.0, 1.0)).view(np.int16)).cuda()
    b = torch.from_numpy(synth.to_bf16_bits(synth.fill_normal(42, n * k, 0.0, 0.05)).view(np.int16)).cuda()
    bias = torch.from_numpy(synth.fill_normal(43, n, 0.0, 0.5)).cuda()
    ref = torch.zeros(m * n, dtype=torch.int16, device="cuda")
    got = torch.zeros(m * n, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    _lib.check(zg.zg_gemm_bf16_nt(a.data_ptr(), b.data_ptr(), bias.data_ptr(), ref.data_ptr(), m, n, k, int(gelu), 1))
    vp = C.c_void_p
    exp.zg_exp_gemm_ov.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
    _lib.check(exp.zg_exp_gemm_ov(a.data_ptr(), b.data_ptr(), bias.data_ptr(), got.data_ptr(), m, n, k, int(gelu)))
    _lib.check(zg.zg_synchronize())
    assert torch.equal(ref, got)
