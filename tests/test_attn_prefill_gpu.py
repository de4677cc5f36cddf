"""The causal prompt attention kernel alone (zg_debug_attn_prefill: scaled_dot_product_attention of src/ops.zig:249-307 for all prompt
positions at once, on the bf16 matrix cores with exact three-plane splits) against a float64 softmax(q k^T / 8) v per head:
prompt lengths around every tile boundary, K / V from the qkv rows and from head-major caches (with NaN behind the prompt's
last row: a cache holds anything there), whole rows and split key ranges, a dominant key (the deferred rescale must fire)."""
import numpy as np
import pytest
import torch

from zig_gpt2_amd import _lib, synth

pytestmark = pytest.mark.gpu


def ref_attention(qkv, B, P, E, H):
    x = qkv.astype(np.float64).reshape(B, P, 3, H, 64)
    q, k, v = x[:, :, 0], x[:, :, 1], x[:, :, 2]
    s = np.einsum("bqhd,bkhd->bhqk", q, k) / 8.0
    s = np.where(np.tril(np.ones((P, P), bool))[None, None], s, -np.inf)
    p = np.exp(s - s.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    return np.einsum("bhqk,bkhd->bqhd", p, v).reshape(B * P, E)


def planes_to_f64(bits, n):
    f = (bits.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    return f[:, :n] + f[:, n:2 * n] + f[:, 2 * n:]


@pytest.mark.parametrize("B,P,H,tiles,cache,spike", [
    (1, 1, 2, 0, False, False), (2, 31, 2, 0, False, False), (1, 32, 3, 0, True, False), (2, 33, 2, 1, True, False),
    (1, 64, 12, 0, True, False), (1, 80, 12, 0, True, False), (1, 95, 12, 0, True, False), (1, 96, 12, 0, True, False),
    (1, 97, 12, 0, False, False), (3, 129, 2, 2, True, True), (1, 300, 4, 3, True, True), (2, 257, 3, 0, False, True),
    (1, 1023, 2, 0, True, False), (1, 1023, 2, 5, False, True)])
def test_attn_prefill_matches_float64(zg, monkeypatch, B, P, H, tiles, cache, spike):
    E, ctx = 64 * H, ((P + 63) // 64) * 64 + 64
    qkv = synth.fill_normal(21 + P, B * P * 3 * E, 0, 1.0).reshape(B * P, 3 * E)
    if spike:  # one key far above the rest for the queries behind it: the running maximum jumps by more than the deferral
        qkv[P // 2, E:E + 64] *= 9.0
        qkv[P // 2 + 1:, :64] += 3.0 * np.sign(qkv[P // 2, E:E + 64])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    qkv_d = dev(qkv)
    out_d = torch.zeros((B * P, 3 * E), dtype=torch.int16, device="cuda")
    ws = torch.zeros(B * H * P * 40 * 66 + 16, dtype=torch.float32, device="cuda") if P <= 320 or tiles else torch.zeros(16 << 20, dtype=torch.float32, device="cuda")
    kc = vc = None
    if cache:
        x = qkv.reshape(B, P, 3, H, 64)
        full = np.full((2, B, H, ctx, 64), np.nan, np.float32)  # rows behind the prompt: anything, NaN included
        full[0, :, :, :P] = x[:, :, 1].transpose(0, 2, 1, 3)
        full[1, :, :, :P] = x[:, :, 2].transpose(0, 2, 1, 3)
        kc, vc = dev(full[0]), dev(full[1])
        qkv_d[:, E:] = float("nan")  # the k / v columns of the rows must not be read
    torch.cuda.synchronize()  # (the fills above run on torch's stream, the library launches on its own)
    _lib.check(zg.zg_debug_attn_prefill(qkv_d.data_ptr(), out_d.data_ptr(), B, P, E, H, kc.data_ptr() if cache else None,
                                        vc.data_ptr() if cache else None, ctx, ws.data_ptr(), ws.numel(), tiles))
    torch.cuda.synchronize()
    got = planes_to_f64(out_d.cpu().numpy().view(np.uint16), E)
    ref = ref_attention(qkv, B, P, E, H)
    assert np.isfinite(got).all(), f"{int((~np.isfinite(got)).sum())} non-finite outputs"
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 2e-6, err
