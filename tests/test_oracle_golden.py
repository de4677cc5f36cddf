"""Pins oracle/zgpt2_oracle.c against the reference's own PyTorch oracles (CPU, no GPU needed).

ops.npz  = outputs of /root/reference/generate_test_data.py — the generator behind the 8 tests of
src/tests.zig; shapes and tolerance follow those tests.  gpt_*.npz = the GPT of
/root/reference/generate_nano_gpt.py run greedily (see tests/golden/make_golden.py).
"""
import numpy as np
import pytest

import oracle
from golden_io import assert_greedy_ids_match, assert_model_close, assert_ref_close, load_gpt, load_ops
from zig_gpt2_amd import synth


@pytest.fixture(scope="module")
def ops():
    return load_ops()


def test_prng_twins_bit_exact():
    for seed, n in [(0, 1), (7, 1000), (12345, 70001)]:
        a = synth.fill_normal(seed, n, 1.0, 0.02, bf16=True)
        b = oracle.fill_normal(seed, n, 1.0, 0.02, round_bf16=True)
        assert a.tobytes() == b.tobytes()
        a = synth.fill_uniform(seed, n, -0.3, 0.7)
        b = oracle.fill_uniform(seed, n, -0.3, 0.7)
        assert a.tobytes() == b.tobytes()
    w = synth.fill_normal(3, 4096, 0.0, 0.02, bf16=True)
    assert (w.view(np.uint32) & 0xFFFF == 0).all()


def test_linear(ops):  # src/tests.zig:22-78
    y = oracle.linear_forward(768, 3072, ops["linear_weight"], ops["linear_bias"], ops["linear_inputs"])
    assert_ref_close(ops["linear_outputs"], y, "Linear")
    y = oracle.linear_forward(768, 3072, ops["linear_weight"], None, ops["linear_inputs"])
    assert_ref_close(ops["linear_outputs_no_bias"], y, "Linear no bias")


def test_embedding(ops):  # src/tests.zig:80-114
    y = oracle.embedding_forward(768, ops["embedding_weight"], ops["embedding_inputs"])
    assert_ref_close(ops["embedding_outputs"], y, "Embedding")


def test_layernorm(ops):  # src/tests.zig:116-155
    y = oracle.layernorm_forward(768, ops["layer_norm_weight"], ops["layer_norm_bias"], ops["layer_norm_inputs"])
    assert_ref_close(ops["layer_norm_outputs"], y, "LayerNorm")


def test_split_qkv(ops):  # src/tests.zig:157-209 (batch 0 is the only pinned one)
    for i, n in enumerate(["split_q", "split_k", "split_v"]):
        y = oracle.split_qkv(768, 5, ops["split_inputs"], i)
        assert_ref_close(ops[n], y, n)


def test_transpose(ops):  # src/tests.zig:211-243
    y = oracle.transpose(5, 12, 64, ops["transpose_inputs"])
    assert_ref_close(ops["transpose_outputs"], y, "transpose")


def test_sdpa(ops):
    """sdpa_* fixtures (generate_test_data.py:119) are unconsumed by src/tests.zig; row s of the
    causal attention equals decode-step attention over the first s+1 keys (src/ops.zig:249-307)."""
    q, k, v, exp = (ops[n][0] for n in ("sdpa_q", "sdpa_k", "sdpa_v", "sdpa_outputs"))
    for s in range(5):
        y = oracle.sdpa(np.ascontiguousarray(q[:, s]), np.ascontiguousarray(k[:, : s + 1]),
                        np.ascontiguousarray(v[:, : s + 1]), 12, s + 1, 64)
        assert_ref_close(exp[:, s], y, f"sdpa step {s}")


def test_attn_forward_incremental(ops):  # src/tests.zig:245-334 (the KV-cache test)
    attn = oracle.CausalSelfAttention(12, 768, ops["attn_c_attn_weight"], ops["attn_c_attn_bias"],
                                      ops["attn_c_proj_weight"], ops["attn_c_proj_bias"], 5)
    for s in range(5):
        y = attn.forward(s + 1, ops["attn_inputs"][0, s])
        assert_ref_close(ops["attn_outputs"][0, s], y, f"attn step {s}")


def test_gelu(ops):  # src/tests.zig:336-360
    assert_ref_close(ops["gelu_outputs"], oracle.gelu(ops["gelu_inputs"]), "gelu")


def test_softmax(ops):  # src/tests.zig:362-388 (row by row)
    x = ops["softmax_inputs"]
    y = np.stack([oracle.softmax(x[b]) for b in range(3)])
    assert_ref_close(ops["softmax_outputs"], y, "softmax")


@pytest.mark.parametrize("name", ["tiny", "tiny3", "nano-char", "124M", "tiny-p24", "tiny3-p40", "124M-p48"])
def test_gpt_greedy_matches_reference_gpt(name):
    """Full model: oracle's generate loop (src/main.zig:322-342, greedy) vs the reference GPT."""
    cfg, g = load_gpt(name)
    w = synth.make_weights(cfg, seed=int(g["weight_seed"]), bf16=True)
    m = oracle.GPT(cfg, w)
    n_steps = len(g["out_tokens"])
    ids, logits = m.generate_greedy(g["prompt"], n_steps, want_logits=True)
    n_prompt = len(g["prompt"])
    assert_greedy_ids_match(g["out_tokens"][n_prompt:], ids[n_prompt:], g["top1"], g["top2"], name)
    # teacher-forced logits on the golden's fed sequence
    lg = m.forced_logits(g["fed"], n_prompt)
    worst = assert_model_close(g["logits"], lg[:, g["logit_cols"]], f"{name} logits")
    print(f"{name}: worst normalised rel err {worst:.2e}")
