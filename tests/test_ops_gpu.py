"""Op-tier parity on a real MI355X, through the C ABI (zig_gpt2_amd.ops -> libzgpt2_hip.so).

The first block replays the 8 tests of the reference's src/tests.zig on the committed golden
vectors (tests/golden/ops.npz, produced by the reference's generate_test_data.py) at the reference
tolerance (src/tests.zig:4-20).  The second block compares against the CPU oracle on seeded inputs
over shapes / edge cases the reference's tests do not reach.
"""
import os

import numpy as np
import pytest

import oracle
from golden_io import assert_ref_close, load_ops
from zig_gpt2_amd import _lib, ops, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    return load_ops()


def z(*shape):
    return np.zeros(shape, np.float32)


# ------------------------------------------------------------------ src/tests.zig, test by test
def test_linear_golden(zg, g):  # src/tests.zig:22-78
    y = z(3, 3072)
    ops.Linear(768, 3072, g["linear_weight"], g["linear_bias"]).forward(g["linear_inputs"], y)
    assert_ref_close(g["linear_outputs"], y, "Linear")
    y2 = z(3, 3072)
    ops.Linear(768, 3072, g["linear_weight"], None).forward(g["linear_inputs"], y2)
    assert_ref_close(g["linear_outputs_no_bias"], y2, "Linear no bias")


def test_embedding_golden(zg, g):  # src/tests.zig:80-114
    y = z(3, 768)
    ops.Embedding(768, g["embedding_weight"]).forward(g["embedding_inputs"].astype(np.uint64), y)
    assert_ref_close(g["embedding_outputs"], y, "Embedding")
    assert np.array_equal(y, g["embedding_outputs"])  # a gather is bit-exact


def test_layernorm_golden(zg, g):  # src/tests.zig:116-155
    x = g["layer_norm_inputs"].copy()
    ops.LayerNorm(768, g["layer_norm_weight"], g["layer_norm_bias"]).forward(x)
    assert_ref_close(g["layer_norm_outputs"], x, "LayerNorm")


def test_split_qkv_golden(zg, g):  # src/tests.zig:157-209
    attn = ops.CausalSelfAttention(12, 768, None, None)
    for i, n in enumerate(["split_q", "split_k", "split_v"]):
        y = z(1, 5, 768)
        attn.split_qkv(5, g["split_inputs"], i, y)
        assert np.array_equal(y, g[n]), n


def test_transpose_golden(zg, g):  # src/tests.zig:211-243
    y = z(1, 12, 5, 64)
    ops.CausalSelfAttention.transpose((5, 12, 64), g["transpose_inputs"], y)
    assert np.array_equal(y, g["transpose_outputs"])


def test_attn_forward_golden(zg, g):  # src/tests.zig:245-334: incremental decode vs causal attention
    e, T = 768, 5
    attn = ops.CausalSelfAttention(12, e, ops.Linear(e, 3 * e, g["attn_c_attn_weight"], g["attn_c_attn_bias"]),
                                   ops.Linear(e, e, g["attn_c_proj_weight"], g["attn_c_proj_bias"]))
    k_cache, v_cache, actual = z(T * e), z(T * e), z(T, e)
    _qkv, _q, _k, _v, _attn = z(3 * e), z(e), z(T * e), z(T * e), z(T)
    for s in range(T):
        attn.forward(s + 1, g["attn_inputs"][0, s], k_cache[: (s + 1) * e], v_cache[: (s + 1) * e], actual[s],
                     _qkv, _q, _k[: (s + 1) * e], _v[: (s + 1) * e], _attn[: s + 1])
        assert_ref_close(g["attn_outputs"][0, s], actual[s], f"attn step {s}")


def test_sdpa_golden(zg, g):  # fixtures the reference generates but never tests
    q, k, v, exp = (g[n][0] for n in ("sdpa_q", "sdpa_k", "sdpa_v", "sdpa_outputs"))
    for s in range(5):
        y = z(12, 64)
        ops.scaled_dot_product_attention(np.ascontiguousarray(q[:, s]), np.ascontiguousarray(k[:, : s + 1]),
                                         np.ascontiguousarray(v[:, : s + 1]), 12, s + 1, 64, y, z(s + 1))
        assert_ref_close(exp[:, s], y, f"sdpa step {s}")


def test_gelu_golden(zg, g):  # src/tests.zig:336-360
    x = g["gelu_inputs"].copy()
    ops.gelu(x)
    assert_ref_close(g["gelu_outputs"], x, "gelu")


def test_softmax_golden(zg, g):  # src/tests.zig:362-388: row by row
    x = g["softmax_inputs"].copy()
    for b in range(3):
        ops.softmax(x[b])
    assert_ref_close(g["softmax_outputs"], x, "softmax")


# ------------------------------------------------------------------ seeded sweeps vs the oracle
@pytest.mark.parametrize("m,k,n", [(1, 768, 2304), (1, 3072, 768), (3, 768, 3072), (8, 768, 768), (9, 384, 1152),
                                   (17, 128, 65), (1, 1600, 4800), (2, 6400, 1600), (5, 1536, 384), (1, 768, 50257),
                                   (4, 100, 37), (1, 8, 1), (2, 64, 3), (1, 6400, 1600), (1, 8192, 264), (1, 2048, 72),
                                   (1, 4104, 40), (4, 6400, 64)])
def test_linear_sweep(zg, m, k, n):
    w = synth.fill_normal(100 + n, n * k, 0, 0.05).reshape(n, k)
    b = synth.fill_normal(200 + n, n, 0, 0.05)
    x = synth.fill_normal(300 + m, m * k, 0, 1.0).reshape(m, k)
    y = z(m, n)
    ops.Linear(k, n, w, b).forward(x, y)
    assert_ref_close(oracle.linear_forward(k, n, w, b, x), y, f"Linear {m}x{k}x{n}", scale_floor=2e-6)


@pytest.mark.parametrize("m,k,n", [(1024, 768, 3072), (16, 128, 8), (300, 3072, 768), (1023, 768, 2304), (64, 1600, 6400),
                                   (16, 768, 50257), (40, 128, 37), (257, 192, 131), (33, 256, 1031)])  # widths that are not multiples of 4
def test_linear_large_batch_runs_on_the_matrix_cores(zg, m, k, n):
    """Linear.forward with batch >= 16 (ops.zig:22: batch = inputs.len / in_features) takes the MFMA GEMM — fp32
    operands split exactly into bf16 planes — and still meets the reference tolerance against the fp32 oracle."""
    w = synth.fill_normal(100 + n, n * k, 0, 0.05).reshape(n, k)
    b = synth.fill_normal(200 + n, n, 0, 0.05)
    x = synth.fill_normal(300 + m, m * k, 0, 1.0).reshape(m, k)
    y = z(m, n)
    before = zg.zg_debug_gemm_launches()
    ops.Linear(k, n, w, b).forward(x, y)
    assert zg.zg_debug_gemm_launches() == before + 1, "batch >= 16 Linear did not take the MFMA path"
    assert_ref_close(oracle.linear_forward(k, n, w, b, x), y, f"Linear {m}x{k}x{n} (MFMA)", scale_floor=2e-6)
    # batch < 16 stays on the GEMV kernels
    before = zg.zg_debug_gemm_launches()
    y8 = z(8, n)
    ops.Linear(k, n, w, b).forward(x[:8], y8)
    assert zg.zg_debug_gemm_launches() == before
    assert_ref_close(y[:8], y8, "GEMV path vs MFMA path", scale_floor=2e-6)


def test_linear_wider_than_the_four_wave_gemm_arguments(zg):
    """in_features = 16384 at batch 16: rows of three planes (49152 elements) and 256 K-steps per plane are beyond what
    gemm_s4_kernel's packed arguments express — the Linear must fall through to the eight-wave GEMM instead of failing
    (the reference Linear has no size limit, src/ops.zig:21-46).  A ragged width, which only the four-wave kernel stores,
    goes to the GEMV kernels — in K chunks of 8192 (round 5; it answered ZG_ERR_UNSUPPORTED before)."""
    m, k, n = 16, 16384, 128
    w = synth.fill_normal(100 + n, n * k, 0, 0.02).reshape(n, k)
    b = synth.fill_normal(200 + n, n, 0, 0.05)
    x = synth.fill_normal(300 + m, m * k, 0, 1.0).reshape(m, k)
    y = z(m, n)
    before = zg.zg_debug_gemm_launches()
    ops.Linear(k, n, w, b).forward(x, y)
    assert zg.zg_debug_gemm_launches() == before + 1
    assert_ref_close(oracle.linear_forward(k, n, w, b, x), y, f"Linear {m}x{k}x{n}", scale_floor=2e-6)
    w130 = np.ascontiguousarray(np.resize(w, (130, k)))
    y130 = z(m, 130)
    ops.Linear(k, 130, w130, None).forward(x, y130)
    assert zg.zg_debug_gemm_launches() == before + 1, "the ragged width must not take the MFMA path"
    assert_ref_close(oracle.linear_forward(k, 130, w130, None, x), y130, f"Linear {m}x{k}x130 (K chunks)", scale_floor=2e-6)


@pytest.mark.parametrize("m,k,n,bias", [(1, 8200, 37, True), (3, 12328, 70, True), (9, 20000, 16, False), (2, 16384, 2100, True),
                                        (2, 8771, 37, True), (5, 16389, 70, False), (17, 9001, 24, True)])
def test_linear_wider_than_8192_runs_in_k_chunks(zg, m, k, n, bias):
    """Linear.forward has no size limit (src/ops.zig:21-46): in_features beyond the 8192 the GEMV kernels hold in LDS runs as K chunks
    (chunk 0 with the bias, later chunks through the residual epilogue onto the same rows; W in row blocks of 2048 for the 2100-row
    case); batch 9 = two row groups.  in_features that is not a multiple of 8 (found by tools/fuzz_ops.py): the ragged chunk runs first,
    on the plain store epilogue — the accumulating epilogue takes whole chunks only."""
    w = synth.fill_normal(100 + n, n * k, 0, 0.02).reshape(n, k)
    b = synth.fill_normal(200 + n, n, 0, 0.05) if bias else None
    x = synth.fill_normal(300 + m, m * k, 0, 1.0).reshape(m, k)
    y = z(m, n)
    ops.Linear(k, n, w, b).forward(x, y)
    assert_ref_close(oracle.linear_forward(k, n, w, b, x), y, f"Linear {m}x{k}x{n} (K chunks)", scale_floor=2e-6)


def test_registered_mirror_is_never_used_for_activations(zg):
    """src/tests.zig frees its weights (`defer allocator.free`) and allocates same-sized buffers next: a host
    address that once held a registered weight may come back as an `inputs` slice.  Activations must never be
    looked up in the registry, and a parameter is only matched with its exact length."""
    k, n, m = 64, 48, 48  # weight [n, k] and inputs [m, k] have the same byte size
    w = synth.fill_normal(1, n * k, 0, 0.05).reshape(n, k)
    buf = np.empty(m * k, np.float32)          # "freed weight" whose address is reused for the inputs
    buf[:] = synth.fill_normal(2, m * k)
    _lib.check(zg.zg_register_tensor(buf.ctypes.data, buf.size))
    buf[:] = synth.fill_normal(3, m * k)       # new contents at the same address: the mirror is now stale
    x = buf.reshape(m, k)
    y = z(m, n)
    ops.Linear(k, n, w, None).forward(x, y)
    assert_ref_close(oracle.linear_forward(k, n, w, None, x), y, "inputs at a once-registered address", scale_floor=2e-6)
    # same address as a PARAMETER with a different length: staged, not matched
    w2 = buf[: (n // 2) * k].reshape(n // 2, k)
    y2 = z(m, n // 2)
    xin = synth.fill_normal(4, m * k).reshape(m, k)
    ops.Linear(k, n // 2, w2, None).forward(xin, y2)
    assert_ref_close(oracle.linear_forward(k, n // 2, w2, None, xin), y2, "parameter with another length", scale_floor=2e-6)
    _lib.check(zg.zg_unregister_tensor(buf.ctypes.data))
    _lib.check(zg.zg_unregister_tensor(buf.ctypes.data))  # idempotent


def test_linear_registered_weight_and_device_pointers(zg):
    import torch

    k, n, m = 768, 1000, 2
    w = synth.fill_normal(1, n * k, 0, 0.05).reshape(n, k)
    x = synth.fill_normal(2, m * k).reshape(m, k)
    exp = oracle.linear_forward(k, n, w, None, x)
    _lib.check(zg.zg_register_tensor(w.ctypes.data, w.size))
    y = z(m, n)
    ops.Linear(k, n, w, None).forward(x, y)
    assert_ref_close(exp, y, "registered weight")
    _lib.check(zg.zg_unregister_all())
    wd, xd = torch.from_numpy(w).cuda(), torch.from_numpy(x).cuda()
    yd = torch.zeros(m, n, device="cuda")
    ops.Linear(k, n, wd, None).forward(xd, yd)
    assert_ref_close(exp, yd.cpu().numpy(), "device pointers")


@pytest.mark.parametrize("rows,n", [(1, 768), (3, 768), (7, 1600), (2, 384), (1, 5), (5, 64)])
def test_layernorm_sweep(zg, rows, n):
    gam = synth.fill_normal(5, n, 1.0, 0.1)
    bet = synth.fill_normal(6, n, 0.0, 0.1)
    x = synth.fill_normal(7 + rows, rows * n, 0.3, 2.0).reshape(rows, n)
    y = x.copy()
    ops.LayerNorm(n, gam, bet).forward(y)
    assert_ref_close(oracle.layernorm_forward(n, gam, bet, x), y, "LayerNorm")


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 768, 1024, 50257])
def test_softmax_sweep(zg, n):
    x = synth.fill_normal(40 + n, n, 0.0, 3.0)
    y = x.copy()
    ops.softmax(y)
    assert_ref_close(oracle.softmax(x), y, f"softmax {n}")
    assert abs(float(y.sum(dtype=np.float64)) - 1.0) < 1e-5


def test_gelu_tails_and_sizes(zg):
    x = np.concatenate([synth.fill_normal(9, 3073, 0, 2.0), np.array([-30, -12, -6, -3, -1e-4, 0, 1e-4, 3, 6, 12, 30], np.float32)])
    y = x.copy()
    ops.gelu(y)
    assert_ref_close(oracle.gelu(x), y, "gelu")
    assert np.isfinite(y).all()


def test_empty_slices_are_noops(zg):
    e = np.zeros(0, np.float32)
    ops.gelu(e)
    ops.softmax(e)
    ops.LayerNorm(8, np.ones(8, np.float32), np.zeros(8, np.float32)).forward(e)
    ops.Linear(8, 4, np.zeros((4, 8), np.float32), None).forward(e, e)


@pytest.mark.parametrize("b,h,t", [(1, 12, 1), (1, 12, 63), (1, 12, 64), (1, 12, 65), (2, 6, 255), (1, 2, 256),
                                   (3, 2, 257), (1, 12, 1024), (1, 25, 513)])
def test_sdpa_sweep(zg, b, h, t):
    q = synth.fill_normal(50 + t, b * h * 64, 0, 1.0)
    k = synth.fill_normal(51 + t, b * h * t * 64, 0, 1.0)
    v = synth.fill_normal(52 + t, b * h * t * 64, 0, 1.0)
    y = z(b * h * 64)
    ops.scaled_dot_product_attention(q, k, v, h, t, 64, y, z(t))
    assert_ref_close(oracle.sdpa(q, k, v, h, t, 64), y, f"sdpa b{b} h{h} t{t}", scale_floor=2e-6)


def test_sdpa_one_dominant_key(zg):
    """Forces the cross-wave / cross-split max rescale: one key far above the rest, placed in each split."""
    h, t = 2, 700
    for hot in (0, 100, 300, 699):
        q = synth.fill_normal(60, h * 64, 0, 1.0)
        k = synth.fill_normal(61, h * t * 64, 0, 0.3).reshape(h, t, 64)
        k[:, hot] = q.reshape(h, 64) * 4.0
        v = synth.fill_normal(62, h * t * 64, 0, 1.0)
        y = z(h * 64)
        ops.scaled_dot_product_attention(q, np.ascontiguousarray(k), v, h, t, 64, y, z(t))
        assert_ref_close(oracle.sdpa(q, np.ascontiguousarray(k).ravel(), v, h, t, 64), y, f"hot key {hot}", scale_floor=2e-6)


def test_attn_forward_long_incremental(zg):
    """KV-cache decode over 300 steps (crosses the 256-position split) vs the oracle's step-by-step attention."""
    e, hds, T = 128, 2, 300
    caw = synth.fill_normal(70, 3 * e * e, 0, 0.08).reshape(3 * e, e)
    cab = synth.fill_normal(71, 3 * e, 0, 0.05)
    cpw = synth.fill_normal(72, e * e, 0, 0.08).reshape(e, e)
    cpb = synth.fill_normal(73, e, 0, 0.05)
    xs = synth.fill_normal(74, T * e, 0, 1.0).reshape(T, e)
    ref = oracle.CausalSelfAttention(hds, e, caw, cab, cpw, cpb, T)
    attn = ops.CausalSelfAttention(hds, e, ops.Linear(e, 3 * e, caw, cab), ops.Linear(e, e, cpw, cpb))
    k_cache, v_cache = z(T * e), z(T * e)
    _qkv, _q, _k, _v, _attn = z(3 * e), z(e), z(T * e), z(T * e), z(T)
    for s in range(T):
        out = z(e)
        attn.forward(s + 1, xs[s], k_cache[: (s + 1) * e], v_cache[: (s + 1) * e], out, _qkv, _q,
                     _k[: (s + 1) * e], _v[: (s + 1) * e], _attn[: s + 1])
        exp = ref.forward(s + 1, xs[s])
        if s % 17 == 0 or s in (255, 256, 257, T - 1):
            assert_ref_close(exp, out, f"step {s}", scale_floor=2e-6)
    # the caller-owned caches hold the same rows as the reference's (ops.zig:152,157)
    assert_ref_close(ref.k_cache, k_cache, "k_cache")
    assert_ref_close(ref.v_cache, v_cache, "v_cache")


def test_attn_forward_cache_mirror_follows_the_callers_cache(zg):
    """zg_attn_forward keeps a device mirror of a caller-owned HOST cache (keyed by its address) and uploads nothing while the
    caller walks the positions in order; whatever else the caller does — restart at position 1 with other inputs, jump ahead
    over rows it filled itself, repeat a position, swap the two caches, drop the mirror after editing a row — the result must
    be what the reference computes from the cache contents the caller holds (ops.zig:129-173)."""
    e, hds, T = 128, 2, 40
    caw = synth.fill_normal(170, 3 * e * e, 0, 0.08).reshape(3 * e, e)
    cab = synth.fill_normal(171, 3 * e, 0, 0.05)
    cpw = synth.fill_normal(172, e * e, 0, 0.08).reshape(e, e)
    cpb = synth.fill_normal(173, e, 0, 0.05)
    attn = ops.CausalSelfAttention(hds, e, ops.Linear(e, 3 * e, caw, cab), ops.Linear(e, e, cpw, cpb))
    k_cache, v_cache = z(T * e), z(T * e)
    _qkv, _q, _k, _v, _attn = z(3 * e), z(e), z(T * e), z(T * e), z(T)

    def step(ref, t, x, kc=None, vc=None):
        kc, vc = (k_cache if kc is None else kc), (v_cache if vc is None else vc)
        out = z(e)
        attn.forward(t, x, kc[: t * e], vc[: t * e], out, _qkv, _q, _k[: t * e], _v[: t * e], _attn[:t])
        assert_ref_close(ref.forward(t, x), out, f"position {t}", scale_floor=2e-6)
        assert_ref_close(ref.k_cache[: t * e], kc[: t * e], f"k rows at {t}")
        assert_ref_close(ref.v_cache[: t * e], vc[: t * e], f"v rows at {t}")
        assert_ref_close(ref._qkv, _qkv, f"_qkv at {t}")  # what the reference leaves in its scratch (ops.zig:143,171)
        assert_ref_close(ref._q, _q, f"_q at {t}")

    xs = synth.fill_normal(174, 3 * T * e, 0, 1.0).reshape(3 * T, e)
    ref = oracle.CausalSelfAttention(hds, e, caw, cab, cpw, cpb, T)
    for t in range(1, 13):               # in order: the mirror is extended row by row
        step(ref, t, xs[t])
    ref = oracle.CausalSelfAttention(hds, e, caw, cab, cpw, cpb, T)
    for t in range(1, 6):                # a new sequence on the same buffers
        step(ref, t, xs[T + t])
    # the caller fills rows 5..8 itself (another producer), then continues at position 10: a jump
    fill_k, fill_v = synth.fill_normal(175, 4 * e, 0, 0.5), synth.fill_normal(176, 4 * e, 0, 0.5)
    for c, f in ((k_cache, fill_k), (ref.k_cache, fill_k), (v_cache, fill_v), (ref.v_cache, fill_v)):
        c[5 * e: 9 * e] = f
    step(ref, 10, xs[2 * T])
    step(ref, 10, xs[2 * T + 1])         # the same position again (row 9 overwritten)
    step(ref, 11, xs[2 * T + 2])
    # edit an earlier row in place and drop the mirror, as the header prescribes
    k_cache[2 * e: 3 * e] = 0.25
    ref.k_cache[2 * e: 3 * e] = 0.25
    from zig_gpt2_amd import _lib
    _lib.check(zg.zg_unregister_tensor(_lib.ptr(k_cache)))
    step(ref, 12, xs[2 * T + 3])
    # the roles of the two buffers swapped: keys in what was the value cache
    ref2 = oracle.CausalSelfAttention(hds, e, caw, cab, cpw, cpb, T)
    for t in range(1, 5):
        step(ref2, t, xs[t + 20], kc=v_cache, vc=k_cache)


def test_attn_forward_cache_mirror_pool_starts_over_when_full():
    """The mirrors of caller-owned caches live in a pool allocated at zg_init (ZGPT2_KV_MIRROR_MB).  A process that keeps making
    new caches (every test of this file does) must not end on the slow whole-cache staging for good: when the pool cannot take
    the caches of a call, all mirrors are dropped and it starts over.  Own process: a 2 MiB pool, caches of 0.5 MiB per slot."""
    import subprocess
    import sys

    code = """
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import oracle
from zig_gpt2_amd import _lib, ops, synth
zg = _lib.load(); _lib.check(zg.zg_init(0))
e, hds, T = 128, 2, 6
caw = synth.fill_normal(70, 3 * e * e, 0, 0.08).reshape(3 * e, e); cab = synth.fill_normal(71, 3 * e, 0, 0.05)
cpw = synth.fill_normal(72, e * e, 0, 0.08).reshape(e, e); cpb = synth.fill_normal(73, e, 0, 0.05)
attn = ops.CausalSelfAttention(hds, e, ops.Linear(e, 3 * e, caw, cab), ops.Linear(e, e, cpw, cpb))
z = lambda n: np.zeros(n, np.float32)
keep = []
for rnd in range(7):  # 14 caches x 0.5 MiB slots through a 2 MiB pool: it must start over several times
    ref = oracle.CausalSelfAttention(hds, e, caw, cab, cpw, cpb, T)
    kc, vc = z(T * e), z(T * e); keep += [kc, vc]
    xs = synth.fill_normal(700 + rnd, T * e, 0, 1.0).reshape(T, e)
    for s in range(T):
        out = z(e)
        attn.forward(s + 1, xs[s], kc[: (s + 1) * e], vc[: (s + 1) * e], out, z(3 * e), z(e), z(T * e), z(T * e), z(T))
        exp = ref.forward(s + 1, xs[s])
        assert np.abs(out - exp).max() < 1e-4 * max(1.0, np.abs(exp).max()), (rnd, s)
    assert np.allclose(kc, ref.k_cache, atol=1e-6) and np.allclose(vc, ref.v_cache, atol=1e-6)
# an older cache continues after its mirror was dropped: rows re-uploaded from the caller's buffer
print("ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, ZGPT2_KV_MIRROR_MB="2"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_device_twin_of_an_elementwise_result_is_only_used_for_unchanged_bytes(zg):
    """The op tier keeps the result of an in-place LayerNorm / gelu on a HOST buffer in device memory as well, and the next Linear
    that is handed the same buffer reads that twin instead of uploading the bytes again (in src/main.zig every Linear's input is
    the previous op's output).  The twin must not survive a change of the caller's buffer: same pointer, one float edited (the
    host loops of src/main.zig:136-145 do exactly that), a shorter slice, another buffer."""
    e, n = 768, 320
    w = synth.fill_normal(301, n * e, 0, 0.05).reshape(n, e)
    lin = ops.Linear(e, n, w, None)
    g, b = synth.fill_normal(302, e, 1, 0.1), synth.fill_normal(303, e, 0, 0.1)
    ln = ops.LayerNorm(e, g, b)

    def ref_linear(x):
        return (x.reshape(-1, e).astype(np.float64) @ w.astype(np.float64).T).ravel()

    x = synth.fill_normal(304, e, 0.2, 1.5)
    ln.forward(x)                                   # x now holds the LayerNorm output; the library keeps a device twin
    y = z(n)
    lin.forward(x, y)                               # served from the twin
    assert np.abs(y - ref_linear(x)).max() < 1e-4
    x[5] += 3.0                                     # the caller edits ONE element on the host
    lin.forward(x, y)
    assert np.abs(y - ref_linear(x)).max() < 1e-4, "stale device twin used after the caller changed its buffer"
    ln.forward(x)
    lin2 = ops.Linear(e // 2, n, np.ascontiguousarray(w[:, : e // 2]), None)
    lin2.forward(x[: e // 2], y)                    # a shorter slice of the same buffer: another length, no twin
    assert np.abs(y - (x[: e // 2].astype(np.float64) @ w[:, : e // 2].astype(np.float64).T)).max() < 1e-4
    other = x.copy()
    lin.forward(other, y)                           # equal bytes at another address: uploaded as usual, same result
    assert np.abs(y - ref_linear(x)).max() < 1e-4
    h = synth.fill_normal(305, 4 * e, 0, 1.0)
    w2 = synth.fill_normal(306, n * 4 * e, 0, 0.03).reshape(n, 4 * e)
    ops.gelu(h)                                     # the gelu -> mlp c_proj pair of src/main.zig:79-81
    ops.Linear(4 * e, n, w2, None).forward(h, y)
    assert np.abs(y - (h.astype(np.float64) @ w2.astype(np.float64).T)).max() < 1e-4
    h[100] = -h[100] + 0.5
    ops.Linear(4 * e, n, w2, None).forward(h, y)
    assert np.abs(y - (h.astype(np.float64) @ w2.astype(np.float64).T)).max() < 1e-4


def test_error_behaviour(zg):
    x = z(10)
    assert zg.zg_layernorm_forward(768, x.ctypes.data, x.ctypes.data, 1e-5, x.ctypes.data, 10) == -2  # ZG_ERR_SHAPE
    assert b"multiple" in zg.zg_last_error()
    w = z(4, 8)
    assert zg.zg_linear_forward(8, 4, w.ctypes.data, None, x.ctypes.data, 8, x.ctypes.data, 3) == -2
    with pytest.raises(_lib.ZgError):
        ops.Embedding(8, z(4, 8)).forward(np.array([5], np.uint64), z(8))  # index out of range
    with pytest.raises(_lib.ZgError):
        ops.scaled_dot_product_attention(z(4096), z(4096), z(4096), 1, 1, 4096, z(4096), z(1))  # head_dim beyond the general kernel's 2048


def test_null_slices_are_refused_not_dereferenced(zg):
    """A NULL pointer with a non-empty length is an argument error on every op (found by tools/fuzz_errors.py: Linear's outputs,
    Embedding's embeddings, split_qkv's and transpose's slices reached a kernel — a GPU memory fault takes the process down)."""
    x, w = z(64), z(4, 8)
    ERR_ARG = -6
    assert zg.zg_linear_forward(8, 4, w.ctypes.data, None, x.ctypes.data, 8, None, 4) == ERR_ARG
    idx = np.zeros(2, np.uint64)
    assert zg.zg_embedding_forward(8, w.ctypes.data, 32, idx.ctypes.data, 2, None, 16) == ERR_ARG
    assert zg.zg_split_qkv(8, 1, None, 24, 0, x.ctypes.data, 8) == ERR_ARG
    assert zg.zg_split_qkv(8, 1, x.ctypes.data, 24, 0, None, 8) == ERR_ARG
    assert zg.zg_transpose(1, 1, 8, None, 8, x.ctypes.data, 8) == ERR_ARG
    assert zg.zg_transpose(1, 1, 8, x.ctypes.data, 8, None, 8) == ERR_ARG
    assert zg.zg_linear_forward(8, 4, w.ctypes.data, None, x.ctypes.data, 8, x.ctypes.data, 4) == 0  # (and the library carries on)


@pytest.mark.parametrize("hds,hd,T", [(3, 32, 70), (2, 48, 300), (5, 80, 33), (1, 128, 513), (4, 7, 19), (2, 300, 40)])
def test_attention_with_a_head_dimension_other_than_64(zg, hds, hd, T):
    """src/ops.zig:249-307 takes any head_dim; the GPT-2 family has 64 (the fast kernels), everything else runs the op tier's
    general attention kernel: scaled_dot_product_attention on head-major q / k / v and CausalSelfAttention.forward over an
    incremental cache, both against the oracle."""
    q = synth.fill_normal(401, hds * hd, 0, 1.0)
    k = synth.fill_normal(402, hds * T * hd, 0, 1.0)
    v = synth.fill_normal(403, hds * T * hd, 0, 1.0)
    out = z(hds * hd)
    ops.scaled_dot_product_attention(q, k, v, hds, T, hd, out, z(T))
    assert_ref_close(oracle.sdpa(q, k, v, hds, T, hd), out, f"sdpa heads {hds} head_dim {hd} T {T}", scale_floor=2e-6)
    e, steps = hds * hd, min(T, 40)
    caw = synth.fill_normal(404, 3 * e * e, 0, 0.04).reshape(3 * e, e)
    cab = synth.fill_normal(405, 3 * e, 0, 0.05)
    cpw = synth.fill_normal(406, e * e, 0, 0.04).reshape(e, e)
    cpb = synth.fill_normal(407, e, 0, 0.05)
    xs = synth.fill_normal(408, steps * e, 0, 1.0).reshape(steps, e)
    ref = oracle.CausalSelfAttention(hds, e, caw, cab, cpw, cpb, steps)
    attn = ops.CausalSelfAttention(hds, e, ops.Linear(e, 3 * e, caw, cab), ops.Linear(e, e, cpw, cpb))
    k_cache, v_cache = z(steps * e), z(steps * e)
    _qkv, _q, _k, _v, _attn = z(3 * e), z(e), z(steps * e), z(steps * e), z(steps)
    for s_ in range(steps):
        o = z(e)
        attn.forward(s_ + 1, xs[s_], k_cache[: (s_ + 1) * e], v_cache[: (s_ + 1) * e], o, _qkv, _q, _k[: (s_ + 1) * e], _v[: (s_ + 1) * e], _attn[: s_ + 1])
        assert_ref_close(ref.forward(s_ + 1, xs[s_]), o, f"attn head_dim {hd} step {s_}", scale_floor=2e-6)
