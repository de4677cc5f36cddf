"""MFMA GEMM (batched-regime Linear, src/ops.zig:21-46 for M >> 1) vs the CPU oracle's Linear and a
plain fp32 matmul of the same bf16-rounded operands.  Tolerance: operands are identical bf16
values on both sides and accumulation is fp32, so only accumulation order differs — reference
tolerance (src/tests.zig:4-20) with the sweep floor of golden_io.assert_ref_close."""
import ctypes as C

import numpy as np
import pytest

import oracle
from golden_io import assert_ref_close
from zig_gpt2_amd import _lib, synth

pytestmark = pytest.mark.gpu


def run_gemm(zg, a, b, bias, gelu, out_bf16):
    import torch

    m, k = a.shape
    n = b.shape[0]
    ad = torch.from_numpy(synth.to_bf16_bits(a).view(np.int16).reshape(m, k)).cuda()
    bd = torch.from_numpy(synth.to_bf16_bits(b).view(np.int16).reshape(n, k)).cuda()
    biasd = None if bias is None else torch.from_numpy(bias).cuda()
    cd = torch.zeros((m, n), dtype=torch.bfloat16 if out_bf16 else torch.float32, device="cuda")
    _lib.check(zg.zg_gemm_bf16_nt(ad.data_ptr(), bd.data_ptr(), None if biasd is None else biasd.data_ptr(),
                                  cd.data_ptr(), m, n, k, int(gelu), int(out_bf16)))
    _lib.check(zg.zg_synchronize())
    return cd.float().cpu().numpy()


@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (128, 256, 768), (256, 3072, 768), (384, 128, 3072), (1024, 768, 768)])
@pytest.mark.parametrize("gelu", [False, True])
def test_gemm_matches_oracle_linear(zg, m, n, k, gelu):
    a = synth.fill_normal(1000 + m, m * k, 0.0, 1.0, bf16=True).reshape(m, k)
    b = synth.fill_normal(2000 + n, n * k, 0.0, 0.05, bf16=True).reshape(n, k)
    bias = synth.fill_normal(3000 + n, n, 0.0, 0.5)
    exp = oracle.linear_forward(k, n, b, bias, a)
    if gelu:
        exp = oracle.gelu(exp)
    got = run_gemm(zg, a, b, bias, gelu, out_bf16=False)
    assert_ref_close(exp, got, f"gemm {m}x{n}x{k} gelu={gelu}", scale_floor=4e-6)


def test_gemm_asymmetric_identity_detects_transposes(zg):
    """A = [I | 0] picks rows of B^T: C[i, j] = B[j, i] for i < K — catches swapped row/column maps."""
    m, n, k = 128, 256, 128
    a = np.zeros((m, k), np.float32)
    a[np.arange(k), np.arange(k)] = 1.0
    b = synth.round_bf16((np.arange(n * k, dtype=np.float32).reshape(n, k) % 251) - 125.0)
    got = run_gemm(zg, a, b, None, False, out_bf16=False)
    assert np.array_equal(got, b.T[:m])


def test_gemm_bf16_output_rounds_to_nearest(zg):
    m, n, k = 128, 128, 192
    a = synth.fill_normal(5, m * k, 0.0, 1.0, bf16=True).reshape(m, k)
    b = synth.fill_normal(6, n * k, 0.0, 0.1, bf16=True).reshape(n, k)
    f32 = run_gemm(zg, a, b, None, False, out_bf16=False)
    b16 = run_gemm(zg, a, b, None, False, out_bf16=True)
    assert np.array_equal(b16, synth.round_bf16(f32))


def test_gemm_rejects_unsupported_shapes(zg):
    import torch

    t = torch.zeros(128 * 128, dtype=torch.int16, device="cuda")
    c = torch.zeros(128 * 128, dtype=torch.float32, device="cuda")
    assert zg.zg_gemm_bf16_nt(t.data_ptr(), t.data_ptr(), None, c.data_ptr(), 100, 128, 64, 0, 0) == -5
    h = np.zeros(128 * 64, np.uint16)
    assert zg.zg_gemm_bf16_nt(h.ctypes.data, t.data_ptr(), None, c.data_ptr(), 128, 128, 64, 0, 0) == -6
