"""BASELINE.json's configurations at FULL size on the GPU (the other model tests use slices the oracle finishes in
seconds): GPT-2 124M with the 8-prompts-per-GPU load of configs[2] through all 1024 context positions, and GPT-2 XL
(48 layers, 1600 wide, V = 50257) through all 1024 positions.  Size-independent properties over the whole run
(identical prompts give identical rows, hipGraph replay equals eager launches, batch 8 equals batch 1) plus the
oracle (the C restatement of src/ops.zig + src/main.zig:322-342 with greedy argmax) on what it finishes in seconds.
A differing greedy id is tolerated only where the oracle's own top-2 logits are within 1e-4 (a numerical tie)."""
import numpy as np
import pytest

import oracle
from golden_io import assert_greedy_ids_match, assert_model_close
from zig_gpt2_amd import gpt as zgpt
from zig_gpt2_amd import ops, synth

pytestmark = pytest.mark.gpu


def fast_weights(cfg, seed):
    """Same tensor set / shapes / init scale as synth.make_weights (bf16-representable), from numpy's generator:
    the portable PRNG of synth takes minutes for the 1.56 G elements of XL."""
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape, mean, _ in synth.tensor_specs(cfg):
        v = rng.standard_normal(int(np.prod(shape)), dtype=np.float32)
        v *= np.float32(0.02)
        v += np.float32(mean)
        w[name] = synth.round_bf16(v).reshape(shape)
    return w


def first_mismatch(a, b):
    d = np.nonzero(np.asarray(a) != np.asarray(b))[0]
    return int(d[0]) if len(d) else None


def test_124m_eight_prompts_full_context(zg):
    """configs[2]'s per-GPU load: 8 sequences in lock step (matrix-core batched Linears, 8 private 1024-position KV
    caches) for the whole context."""
    cfg = synth.CONFIGS["124M"]
    w = synth.make_weights(cfg, seed=7, bf16=True)
    ctx = cfg.context_size
    p0 = synth.rand_tokens(701, 3, cfg.vocab_size)
    prompts = [p0] * 6 + [synth.rand_tokens(702, 1, cfg.vocab_size), synth.rand_tokens(703, 5, cfg.vocab_size)]
    m8 = zgpt.GPT(cfg, batch=8)
    m8.load_weights(w)
    ids8 = m8.generate(prompts, ctx)
    m8.close()
    assert ids8.shape == (8, ctx) and int(ids8.max()) < cfg.vocab_size
    for b in range(1, 6):  # identical prompts -> identical rows, all 1024 positions
        assert np.array_equal(ids8[0], ids8[b]), f"row {b} differs from row 0 at {first_mismatch(ids8[0], ids8[b])}"
    for b, p in enumerate(prompts):
        assert np.array_equal(ids8[b, : len(p)], p)
    # oracle: the whole run of row 0 (a few seconds of CPU), the first 64 steps of rows 6 and 7
    ref0, lg0 = oracle.GPT(cfg, w).generate_greedy(p0, ctx, want_logits=True)
    top0 = np.sort(lg0, axis=1)
    assert_greedy_ids_match(ref0[len(p0):], ids8[0, len(p0):], top0[:, -1], top0[:, -2], "124M x8 row 0, 1024 ctx")
    for b in (6, 7):
        n = len(prompts[b])
        ref, lg = oracle.GPT(cfg, w).generate_greedy(prompts[b], 64, want_logits=True)
        top = np.sort(lg, axis=1)
        assert_greedy_ids_match(ref[n:], ids8[b, n:64], top[:, -1], top[:, -2], f"124M x8 row {b}")
    # the batch-1 path (VALU GEMVs, hipGraph) produces the same tokens, position for position
    m1 = zgpt.GPT(cfg, batch=1)
    m1.load_weights(w)
    ids1 = m1.generate([p0], ctx)[0]
    m1.close()
    assert_greedy_ids_match(ref0[len(p0):], ids1[len(p0):], top0[:, -1], top0[:, -2], "124M batch 1, 1024 ctx")
    if first_mismatch(ids1, ids8[0]) is None:
        return
    i = first_mismatch(ids1, ids8[0])  # only legal at a numerical tie of the oracle's logits
    assert float(top0[i - len(p0), -1] - top0[i - len(p0), -2]) < 1e-4, f"batch 1 and batch 8 diverge at {i}"


def test_124m_eight_prompts_fp16_kv_cache(zg, monkeypatch):
    """The 16-bit KV cache (ZG_GPT_KV_F16: halves the attention traffic of the 8-prompt load) at configs[2]'s per-GPU load
    over the whole context.  MEASURED: its teacher-forced logits deviate by up to 1.7e-3 of the logit scale from the
    fp32-cache handle at 1024 positions (11 significant bits per cached element) — outside north_star's 1e-3, which is
    why fp32 stays the default and the bench's headline; the bound asserted here (2.5e-3) documents the option, and the
    short-context bound (1e-3 over 64 positions) is in test_gpt_gpu.py.  The 16-byte-load kernel must reproduce the
    8-byte-load one."""
    cfg = synth.CONFIGS["124M"]
    w = synth.make_weights(cfg, seed=7, bf16=True)
    ctx = cfg.context_size
    prompts = [synth.rand_tokens(720 + b, 1 + b % 4, cfg.vocab_size) for b in range(8)]
    m32 = zgpt.GPT(cfg, batch=8)
    m32.load_weights(w)
    ids32 = m32.generate(prompts, ctx)
    m16 = zgpt.GPT(cfg, batch=8, kv_f16=True)
    m16.load_weights(w)
    ids16 = m16.generate(prompts, ctx)
    # teacher-forced logits of both handles on the fp32 handle's tokens, at positions spread over the context
    worst = 0.0
    for s in range(ctx):
        want = s in (0, 1, 63, 64, 255, 256, 511, 777, 1023)
        toks = [int(ids32[b, s]) for b in range(8)]
        l32 = m32.forward(s + 1, toks, compute_logits=want)
        l16 = m16.forward(s + 1, toks, compute_logits=want)
        if want:
            for b in range(8):
                rms = float(np.sqrt(np.mean(l32[b].astype(np.float64) ** 2)))
                worst = max(worst, float(np.abs(l16[b] - l32[b]).max()) / rms)
    assert worst <= 2.5e-3, f"fp16 KV cache: worst logit deviation {worst:.2e} of the logit scale"
    print(f"fp16 KV cache at 124M x 8 x 1024: worst logit deviation {worst:.2e} of the logit scale")
    agree = float((ids16 == ids32).mean())
    assert agree > 0.5, agree  # a greedy run may leave the fp32 run's path at a near-tie and never return: ids are checked teacher-forced above
    m32.close()
    m16.close()


def test_xl_full_size(zg):
    """GPT-2 XL at full size: 48 layers of E = 1600 (K = 1600 / 6400 kernels, 25 heads), lm_head 1600 -> 50257."""
    cfg = synth.CONFIGS["xl"]
    w = fast_weights(cfg, 11)
    ctx = cfg.context_size
    prompt = synth.rand_tokens(1101, 2, cfg.vocab_size)
    m = zgpt.GPT(cfg, batch=1)
    m.load_weights(w)
    ids = m.generate([prompt], ctx)[0]
    lg_dev = m.forward(1, [int(prompt[0])])[0]
    m.close()
    assert ids.shape == (ctx,) and int(ids.max()) < cfg.vocab_size and np.array_equal(ids[:2], prompt)
    assert np.isfinite(lg_dev).all()
    # hipGraph replay == eager launches over the whole context
    me = zgpt.GPT(cfg, batch=1, use_graph=False)
    me.load_weights(w)
    ids_e = me.generate([prompt], ctx)[0]
    me.close()
    assert np.array_equal(ids, ids_e), f"graph and eager runs diverge at {first_mismatch(ids, ids_e)}"
    # oracle: first 24 steps (0.4 s per token on the host) and the logits of position 1
    ref = oracle.GPT(cfg, w)
    ids_ref, lg = ref.generate_greedy(prompt, 24, want_logits=True)
    top = np.sort(lg, axis=1)
    assert_greedy_ids_match(ids_ref[2:], ids[2:24], top[:, -1], top[:, -2], "XL full size")
    exp = ref.forward(1, int(prompt[0]))
    assert_model_close(exp, lg_dev, "XL logits at position 1")


@pytest.mark.parametrize("m", [1, 8])
def test_lm_head_1600_to_50257(zg, m):
    """XL's vocabulary projection as an op (Linear 1600 -> 50257, no bias: src/main.zig:191-193) and its argmax."""
    from golden_io import assert_ref_close

    k, n = 1600, 50257
    wt = synth.fill_normal(31, n * k, 0, 0.02).reshape(n, k)
    x = synth.fill_normal(32 + m, m * k, 0, 1.0).reshape(m, k)
    y = np.zeros((m, n), np.float32)
    ops.Linear(k, n, wt, None).forward(x, y)
    exp = oracle.linear_forward(k, n, wt, None, x)
    assert_ref_close(exp, y, f"lm_head {m}x{k}x{n}", scale_floor=2e-6)
    assert np.array_equal(exp.argmax(axis=1), y.argmax(axis=1))
