"""One whole-prompt Linear at a time (zg_debug_prefill_linear: Linear.forward of src/ops.zig:21-46 for M = batch x prompt rows, as
zg_gpt_prefill launches it) on both GEMM families — the persistent four-wave kernel of gemm_s4.hip with the three activation
planes in one K loop (K slices + partial slabs for the residual adds, GELU + three-plane split), and the 128-row kernels of
prefill.hip — against a float64 product of the same operands.  The operands are exact (fp32 rows as three bf16 planes, bf16
weights), so the only error is fp32 accumulation order: the reference tolerance of src/tests.zig:4-20 applies."""
import numpy as np
import pytest
import torch

from golden_io import assert_ref_close
from zig_gpt2_amd import _lib, synth

pytestmark = pytest.mark.gpu


def bf16_round(x):
    return (synth.to_bf16_bits(x).astype(np.uint32) << 16).view(np.float32)


def split3(x):
    """fp32 [M, K] -> bf16 bits [M, 3K] = hi | mid | lo with hi + mid + lo == x exactly."""
    hi = bf16_round(x)
    r = (x - hi).astype(np.float32)
    mid = bf16_round(r)
    lo = bf16_round((r - mid).astype(np.float32))
    assert np.array_equal((hi.astype(np.float64) + mid + lo).astype(np.float32), x)
    return np.concatenate([synth.to_bf16_bits(hi), synth.to_bf16_bits(mid), synth.to_bf16_bits(lo)], axis=1)


def planes_to_f64(bits, n):
    f = (bits.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    return f[:, :n] + f[:, n:2 * n] + f[:, 2 * n:]


def gelu64(x):
    return 0.5 * x * (1.0 + np.tanh(x * 0.7978845608 * (1.0 + 0.044715 * x * x)))


@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("M,N,K,epilogue,slices,wgs", [
    (600, 768, 256, 1, 1, None), (600, 768, 256, 1, 2, None), (600, 768, 768, 1, 4, None), (300, 320, 768, 1, 3, 2),
    (1030, 384, 1536, 1, 2, 3), (600, 768, 256, 2, 0, None), (520, 1600, 320, 2, 0, 3), (257, 192, 128, 2, 0, None)])
def test_prefill_linear_matches_float64(zg, monkeypatch, kernel, M, N, K, epilogue, slices, wgs):
    if kernel == 2 and slices > 1:
        pytest.skip("K slices are the four-wave kernel's")
    if wgs:
        monkeypatch.setenv("ZGPT2_GEMM_WGS", str(wgs))
    x = synth.fill_normal(11, M * K, 0, 1.0).reshape(M, K)
    w = bf16_round(synth.fill_normal(12, N * K, 0, 0.05).reshape(N, K))
    bias = synth.fill_normal(13, N, 0, 0.1)
    c0 = synth.fill_normal(14, M * N, 0, 1.0).reshape(M, N)
    ref = x.astype(np.float64) @ w.astype(np.float64).T + bias
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    a_d, w_d, b_d = dev(split3(x).view(np.int16)), dev(synth.to_bf16_bits(w).view(np.int16)), dev(bias)
    ws = torch.zeros(max(slices, 4) * M * N, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()  # (the fills above run on torch's stream, the library launches on its own)
    before = zg.zg_debug_gemm_launches()
    if epilogue == 1:
        c_d = dev(c0)
        _lib.check(zg.zg_debug_prefill_linear(a_d.data_ptr(), w_d.data_ptr(), b_d.data_ptr(), c_d.data_ptr(), M, N, K, 1, kernel, slices, ws.data_ptr(), ws.numel()))
        torch.cuda.synchronize()
        assert_ref_close(ref + c0, c_d.cpu().numpy(), f"resid {M}x{N}x{K} kernel {kernel} slices {slices}", scale_floor=2e-6)
    else:
        c_d = torch.zeros((M, 3 * N), dtype=torch.int16, device="cuda")
        _lib.check(zg.zg_debug_prefill_linear(a_d.data_ptr(), w_d.data_ptr(), b_d.data_ptr(), c_d.data_ptr(), M, N, K, 2, kernel, 0, ws.data_ptr(), ws.numel()))
        torch.cuda.synchronize()
        got = planes_to_f64(c_d.cpu().numpy().view(np.uint16), N)
        assert_ref_close(gelu64(ref), got, f"gelu-split {M}x{N}x{K} kernel {kernel}", scale_floor=2e-6)
    took_s4 = zg.zg_debug_gemm_launches() - before
    assert took_s4 == (1 if kernel == 1 else 0), "the forced GEMM family did not run"
