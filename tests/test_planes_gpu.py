"""Lock-step batch with bf16 weights: the activation planes between the kernels of a Block (GemvArgs.pl_in / pl_out,
zg_common.h plane_elem), the four-wave plane-fed Linear, the attention-side head merge and the tagged hand-overs —
against the CPU oracle, and against the paths they replace (the bits of ZGPT2_DECODE_PATHS_OFF
switch them off per handle), which stay in the library as the route for shapes the new kernels do not take.

Tolerance: greedy ids against independent oracle generations (golden_io.assert_greedy_ids_match: identical unless the
oracle's own top-2 margin is inside the north_star bound); between the variants only the summation order of fp32 partial sums differs: 1e-5 of the logit scale, greedy ids identical."""
import numpy as np
import pytest

import oracle
from golden_io import assert_greedy_ids_match
from zig_gpt2_amd import gpt as zgpt
from zig_gpt2_amd import synth
from zig_gpt2_amd.synth import GPTConfig

pytestmark = pytest.mark.gpu

# ZGPT2_DECODE_PATHS_OFF bits (zg_common.h): 1 planes between kernels, 2 the four-wave Linear, 4 tagged hand-overs, 8 tile statistics
VARIANTS = {"default": {}, "tickets": {"ZGPT2_DECODE_PATHS_OFF": "4"}, "16-wave": {"ZGPT2_DECODE_PATHS_OFF": "2"},
            "16-wave tickets": {"ZGPT2_DECODE_PATHS_OFF": "6"}, "LDS planes": {"ZGPT2_DECODE_PATHS_OFF": "1"},
            "statistics from x": {"ZGPT2_DECODE_PATHS_OFF": "8"}}


def run_variant(monkeypatch, env, cfg, w, batch, prompts, n_steps, **kw):
    monkeypatch.delenv("ZGPT2_DECODE_PATHS_OFF", raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    m = zgpt.GPT(cfg, batch=batch, **kw)
    m.load_weights(w)
    ids = m.generate(prompts, n_steps)
    lg = m.forward(3, [int(p[0]) for p in prompts])
    lg2 = m.forward(3, [int(p[0]) for p in prompts])  # the same step again: new tags, same numbers
    m.close()
    assert np.array_equal(lg, lg2)
    return ids, lg


@pytest.mark.parametrize("name,batch", [("tiny", 3), ("nano-char", 5), ("medium-slice", 2), ("medium-slice", 8), ("xl-slice", 8)])
def test_plane_variants_agree_and_match_oracle(zg, monkeypatch, name, batch):
    """E = 128 / 384 (one / two 64-k pairs per wave, no K slices), E = 1024 (four pairs, mlp c_proj in four K slices with
    the tagged hand-over), E = 1600 (beyond the four-wave kernel: the 16-wave kernel reads the planes) — every variant
    against the first, the default against independent oracle generations."""
    cfg = synth.CONFIGS[name]
    w = synth.make_weights(cfg, seed=300 + batch, bf16=True)
    prompts = [synth.rand_tokens(3000 + b, 1 + (b * 2) % 5, cfg.vocab_size) for b in range(batch)]
    n_steps = min(cfg.context_size, 48)
    res = {v: run_variant(monkeypatch, env, cfg, w, batch, prompts, n_steps) for v, env in VARIANTS.items()}
    ids0, lg0 = res["default"]
    scale = np.abs(lg0).max()
    for v, (ids, lg) in res.items():
        assert np.array_equal(ids, ids0), v
        assert np.abs(lg - lg0).max() <= 1e-5 * scale, (v, np.abs(lg - lg0).max(), scale)
    for b in range(min(batch, 3)):
        ids_ref, lgr = oracle.GPT(cfg, w).generate_greedy(prompts[b], n_steps, want_logits=True)
        top = np.sort(lgr, axis=1)
        n = len(prompts[b])
        assert_greedy_ids_match(ids_ref[n:], ids0[b, n:], top[:, -1], top[:, -2], f"{name} row {b}")


def test_long_context_merges_more_than_four_splits(zg, monkeypatch):
    """Context 1536: up to six attention splits per head — the merging split polls them one at a time (the unrolled
    hand-over covers four).  Two sequences against independent oracle generations over the whole context, tagged and
    ticket hand-overs."""
    cfg = GPTConfig(97, 1536, 1, 2, 128)
    w = synth.make_weights(cfg, seed=77, bf16=True)
    prompts = [synth.rand_tokens(770 + b, 1 + b, cfg.vocab_size) for b in range(2)]
    n_steps = cfg.context_size
    out = {}
    for v in ("default", "tickets"):
        ids, _ = run_variant(monkeypatch, VARIANTS[v], cfg, w, 2, prompts, n_steps)
        out[v] = ids
    assert np.array_equal(out["default"], out["tickets"])
    for b in range(2):
        ids_ref, lgr = oracle.GPT(cfg, w).generate_greedy(prompts[b], n_steps, want_logits=True)
        top = np.sort(lgr, axis=1)
        n = len(prompts[b])
        assert_greedy_ids_match(ids_ref[n:], out["default"][b, n:], top[:, -1], top[:, -2], f"ctx 1536 row {b}")


def test_fp16_cache_with_planes(zg, monkeypatch):
    """The 16-bit KV cache (opt-in) under the attention-side merge: same tokens with and without the planes.  A last-bit
    difference of a K / V element (summation order of the two c_attn kernels) can flip its rounding to fp16, i.e. move it
    by 2^-11 of its value: the logits agree to 2e-4 of their scale here, not to the 1e-5 of the fp32 cache."""
    cfg = synth.CONFIGS["nano-char"]
    w = synth.make_weights(cfg, seed=5, bf16=True)
    prompts = [synth.rand_tokens(50 + b, 2, cfg.vocab_size) for b in range(4)]
    a, la = run_variant(monkeypatch, {}, cfg, w, 4, prompts, cfg.context_size, kv_f16=True)
    b, lb = run_variant(monkeypatch, {"ZGPT2_DECODE_PATHS_OFF": "1"}, cfg, w, 4, prompts, cfg.context_size, kv_f16=True)
    assert np.array_equal(a, b)
    assert np.abs(la - lb).max() <= 2e-4 * np.abs(la).max()


def test_timed_out_hand_over_fails_the_call(zg, monkeypatch):
    """The pollers of the tagged hand-overs give up after a bounded number of polls, raise a fault word, and the call that
    drains the stream fails instead of returning tokens computed from partial sums that never arrived.  With the bound
    at zero every poll that does not find its writers' tags at once is a time-out (the first poll of a K slice practically
    never does); the next handle, with the normal bound, is unaffected."""
    from zig_gpt2_amd import _lib

    cfg = synth.CONFIGS["medium-slice"]  # E = 1024: mlp c_proj in four K slices, hand-over in every layer of every step
    w = synth.make_weights(cfg, seed=9, bf16=True)
    prompts = [synth.rand_tokens(90 + b, 1, cfg.vocab_size) for b in range(8)]
    monkeypatch.setenv("ZGPT2_TAG_SPIN_LIMIT", "0")
    m = zgpt.GPT(cfg, batch=8)
    m.load_weights(w)
    with pytest.raises(_lib.ZgError, match="hand-over"):
        for _ in range(4):  # (one generation is several hundred hand-overs; four to be sure one of them has to wait)
            m.generate(prompts, cfg.context_size)
    m.close()
    monkeypatch.delenv("ZGPT2_TAG_SPIN_LIMIT")
    ids, _ = run_variant(monkeypatch, {}, cfg, w, 8, prompts, 32)
    ids2, _ = run_variant(monkeypatch, VARIANTS["tickets"], cfg, w, 8, prompts, 32)
    assert np.array_equal(ids, ids2)
