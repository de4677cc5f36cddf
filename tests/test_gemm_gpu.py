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
    cd = torch.full((m, n), float("nan"), dtype=torch.bfloat16 if out_bf16 else torch.float32, device="cuda")
    torch.cuda.synchronize()  # the library launches on its own non-blocking stream
    _lib.check(zg.zg_gemm_bf16_nt(ad.data_ptr(), bd.data_ptr(), None if biasd is None else biasd.data_ptr(),
                                  cd.data_ptr(), m, n, k, int(gelu), int(out_bf16)))
    _lib.check(zg.zg_synchronize())
    return cd.float().cpu().numpy()


@pytest.mark.parametrize("m,n,k", [(128, 128, 128), (128, 256, 768), (256, 3072, 768), (384, 128, 3072), (1024, 768, 768),
                                   (100, 136, 192), (1000, 200, 128), (1023, 2304, 768), (16, 8, 1600),
                                   (64, 126, 128), (300, 131, 192), (257, 7, 128)])  # fp32 output: widths that are not multiples of 4
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
    m, n, k = 128, 136, 192
    a = synth.fill_normal(5, m * k, 0.0, 1.0, bf16=True).reshape(m, k)
    b = synth.fill_normal(6, n * k, 0.0, 0.1, bf16=True).reshape(n, k)
    f32 = run_gemm(zg, a, b, None, False, out_bf16=False)
    b16 = run_gemm(zg, a, b, None, False, out_bf16=True)
    assert np.array_equal(b16, synth.round_bf16(f32))


def test_gemm_rejects_unsupported_shapes(zg):
    import torch

    t = torch.zeros(128 * 128, dtype=torch.int16, device="cuda")
    c = torch.zeros(128 * 128, dtype=torch.float32, device="cuda")
    assert zg.zg_gemm_bf16_nt(t.data_ptr(), t.data_ptr(), None, c.data_ptr(), 128, 128, 64, 0, 0) == -5   # K < 128
    assert zg.zg_gemm_bf16_nt(t.data_ptr(), t.data_ptr(), None, c.data_ptr(), 128, 128, 100, 0, 0) == -5  # K % 64
    assert zg.zg_gemm_bf16_nt(t.data_ptr(), t.data_ptr(), None, c.data_ptr(), 128, 132, 128, 0, 1) == -5  # bf16 output: N % 8
    assert zg.zg_gemm_bf16_nt(t.data_ptr(), t.data_ptr(), None, c.data_ptr(), 128, 126, 128, 0, 0) == 0   # fp32 output: any width
    h = np.zeros(128 * 128, np.uint16)
    assert zg.zg_gemm_bf16_nt(h.ctypes.data, t.data_ptr(), None, c.data_ptr(), 128, 128, 128, 0, 0) == -6


@pytest.mark.parametrize("bn,wgs", [(192, 8), (256, 8), (192, 3), (256, 5), (192, 1)])
@pytest.mark.parametrize("out_bf16", [False, True])
def test_gemm_many_tiles_per_workgroup(zg, bn, wgs, out_bf16, monkeypatch):
    """The persistent kernel with few workgroups: every workgroup walks several tiles (tile hand-over with the
    next tile's operands already in flight, odd K-step counts, ragged edges), both tile widths, both output types
    (the bf16 epilogue goes through an LDS image); repeated runs must agree bit for bit (race screen: a missing
    barrier behind the tile hand-over once showed up only with more than two tiles per workgroup)."""
    monkeypatch.setenv("ZGPT2_GEMM_KERNEL", "s4" if bn == 192 else "p8:256")  # 192-wide tiles: the four-wave kernel; 256-wide: the eight-wave one
    monkeypatch.setenv("ZGPT2_GEMM_WGS", str(wgs))
    m, n, k = 1100, 776, 320
    a = synth.fill_normal(7, m * k, 0.0, 1.0, bf16=True).reshape(m, k)
    b = synth.fill_normal(8, n * k, 0.0, 0.05, bf16=True).reshape(n, k)
    bias = synth.fill_normal(9, n, 0.0, 0.5)
    exp = oracle.gelu(oracle.linear_forward(k, n, b, bias, a))
    runs = [run_gemm(zg, a, b, bias, True, out_bf16=out_bf16) for _ in range(6)]
    if out_bf16:
        assert np.abs(runs[0] - exp).max() <= 0.01 * np.abs(exp).max()
        assert np.array_equal(runs[0], synth.round_bf16(run_gemm(zg, a, b, bias, True, out_bf16=False)))
    else:
        assert_ref_close(exp, runs[0], f"gemm bn={bn} wgs={wgs}", scale_floor=4e-6)
    assert all(np.array_equal(runs[0], r) for r in runs[1:])


def test_gemm_768x3072_full_size_properties(zg):
    """BASELINE's GEMM point at full size (M = 8192: 512 tiles, two per workgroup): linearity in A (rows scaled by
    powers of two scale the pre-activation exactly) and agreement of a row sample with the oracle."""
    import torch

    m, n, k = 8192, 3072, 768
    a = synth.fill_normal(11, m * k, 0.0, 1.0, bf16=True).reshape(m, k)
    b = synth.fill_normal(12, n * k, 0.0, 0.02, bf16=True).reshape(n, k)
    bias = synth.fill_normal(13, n, 0.0, 0.02)
    y = run_gemm(zg, a, b, bias, False, out_bf16=False)
    y2 = run_gemm(zg, 2.0 * a, b, None, False, out_bf16=False)
    y0 = run_gemm(zg, a, b, None, False, out_bf16=False)
    assert np.array_equal(y2, 2.0 * y0)
    rows = np.arange(0, m, 257)
    assert_ref_close(oracle.linear_forward(k, n, b, bias, a[rows]), y[rows], "gemm 8192 row sample", scale_floor=4e-6)


def test_gemm_768x3072_timed_instantiation(zg):
    """The instantiation bench.py times — bias + GELU + bf16 result at M = 8192 (gemm_s4_kernel<3, true, true, 0>, 512 tiles,
    write-through stores): a row sample against the oracle within bf16 rounding, and the whole result equal to the fp32-output
    run of the same kernel family rounded to the nearest bf16."""
    m, n, k = 8192, 3072, 768
    a = synth.fill_uniform(21, m * k, -1.0, 1.0, bf16=True).reshape(m, k)
    b = synth.fill_normal(22, n * k, 0.0, 0.02, bf16=True).reshape(n, k)
    bias = synth.fill_normal(23, n, 0.0, 0.02)
    got = run_gemm(zg, a, b, bias, True, out_bf16=True)
    f32 = run_gemm(zg, a, b, bias, True, out_bf16=False)
    assert np.array_equal(got, synth.round_bf16(f32))
    rows = np.arange(3, m, 251)
    exp = oracle.gelu(oracle.linear_forward(k, n, b, bias, a[rows]))
    # bf16 keeps 8 significant bits: half an ulp is 2^-9 of the value; the fp32 pre-activation itself is held to the
    # reference tolerance by the fp32-output tests above
    assert np.all(np.abs(got[rows] - exp) <= 2.0 ** -8 * np.abs(exp) + 1e-6), np.abs(got[rows] - exp).max()
    assert_ref_close(exp, f32[rows], "gemm 8192 gelu row sample (fp32 out)", scale_floor=4e-6)


def test_gemm_k_beyond_the_packed_arguments_runs_on_the_eight_wave_kernel(zg):
    """K = 16384 (256 K-steps) is past gemm_s4_kernel's packed arguments: the dispatcher must fall through to gemm_p8_kernel
    instead of failing; a ragged fp32 width, which only the four-wave kernel stores, is refused with ZG_ERR_UNSUPPORTED."""
    import torch

    m, n, k = 128, 256, 16384
    a = synth.fill_normal(31, m * k, 0.0, 1.0, bf16=True).reshape(m, k)
    b = synth.fill_normal(32, n * k, 0.0, 0.01, bf16=True).reshape(n, k)
    bias = synth.fill_normal(33, n, 0.0, 0.5)
    got = run_gemm(zg, a, b, bias, False, out_bf16=False)
    assert_ref_close(oracle.linear_forward(k, n, b, bias, a), got, "gemm K=16384", scale_floor=4e-6)
    t = torch.zeros(128 * 16384, dtype=torch.int16, device="cuda")
    c = torch.zeros(128 * 128, dtype=torch.float32, device="cuda")
    assert zg.zg_gemm_bf16_nt(t.data_ptr(), t.data_ptr(), None, c.data_ptr(), 128, 126, 16384, 0, 0) == -5
