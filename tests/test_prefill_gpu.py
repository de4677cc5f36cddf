"""Whole-prompt forward (zg_gpt_prefill) on a real MI355X.

The reference feeds a prompt one position at a time through GPT.forward (src/main.zig:331-334); the
prefill pass must leave the KV caches and the last-position logits in the state those calls would:
checked against the CPU oracle (which does exactly that loop), against the golden vectors of the
reference's PyTorch GPT, and against the library's own decode path at the full 124M / 1024-token size.
Tolerance is the model tolerance of golden_io.assert_model_close (1e-3 relative, north_star).
"""
import numpy as np
import pytest

import oracle
from golden_io import assert_greedy_ids_match, assert_model_close, load_gpt
from zig_gpt2_amd import _lib
from zig_gpt2_amd import gpt as zgpt
from zig_gpt2_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture
def every_linear_on_s4(zg):
    """Test hook of the C ABI: every whole-prompt Linear of the process takes the persistent four-wave GEMM, whatever its tile count."""
    _lib.check(zg.zg_debug_prefill_route(1, 0))
    yield
    _lib.check(zg.zg_debug_prefill_route(0, 0))


def make(cfg, seed, **kw):
    w = synth.make_weights(cfg, seed=seed, bf16=True)
    m = zgpt.GPT(cfg, **kw)
    m.load_weights(w)
    return m, w


@pytest.mark.parametrize("name,lengths", [("tiny", [1, 2, 5, 31, 32, 33, 64]), ("tiny3", [7, 48]),
                                          ("nano-char", [3, 127, 128, 129, 256]), ("xl-slice", [40, 95])])
def test_prefill_logits_and_cache_match_oracle(zg, name, lengths):
    """logits of position n-1 after prefill(n tokens) == oracle after n GPT.forward calls; then one decode
    step on top of the prefilled caches == the oracle's next step (pins the KV cache contents)."""
    cfg = synth.CONFIGS[name]
    m, w = make(cfg, 71)
    for n in lengths:
        toks = synth.rand_tokens(700 + n, min(n + 1, cfg.context_size), cfg.vocab_size)
        ref = oracle.GPT(cfg, w)
        lg_ref = ref.forced_logits(toks, n - 1)
        lg = m.prefill([toks[:n]])
        assert_model_close(lg_ref[0], lg[0], f"{name} prefill n={n}")
        assert int(m.argmax()[0]) == int(np.argmax(lg[0]))
        if n < cfg.context_size:
            nxt = m.forward(n + 1, [toks[n]])
            assert_model_close(lg_ref[1], nxt[0], f"{name} decode after prefill n={n}")
    m.close()


@pytest.mark.parametrize("name,batch,lengths,wgs", [("tiny", 1, [5, 33, 64], None), ("tiny", 3, [21, 64], 2), ("tiny3", 2, [48], 3),
                                                    ("nano-char", 4, [129, 256], None), ("xl-slice", 1, [95], 5), ("medium-slice", 2, [80], None)])
def test_prefill_linears_on_the_persistent_four_wave_gemm(zg, monkeypatch, every_linear_on_s4, name, batch, lengths, wgs):
    """Large prompts run their Linears on gemm_s4 (three planes in one K loop; slab epilogue + reduce for the residual adds,
    GELU + split, qkv + cache append).  zg_debug_prefill_route(1, 0) sends every shape there: small models, ragged tile edges (M far
    below 256, N = 1600 = 8.33 tiles), several tiles per workgroup (ZGPT2_GEMM_WGS), K slices.  Same checks as the 128-row
    path: last-position logits and the decode step on top of the prefilled caches against the oracle."""
    if wgs:
        monkeypatch.setenv("ZGPT2_GEMM_WGS", str(wgs))
    cfg = synth.CONFIGS[name]
    m, w = make(cfg, 75, batch=batch)
    for n in lengths:
        toks = np.stack([synth.rand_tokens(750 + 7 * b + n, min(n + 1, cfg.context_size), cfg.vocab_size) for b in range(batch)])
        before = zg.zg_debug_gemm_launches()
        lg = m.prefill(toks[:, :n])
        assert zg.zg_debug_gemm_launches() - before >= 4 * cfg.n_layer, "the whole-prompt Linears did not run on gemm_s4"
        nxt = m.forward(n + 1, toks[:, n]) if n < cfg.context_size else None
        for b in range(batch):
            lg_ref = oracle.GPT(cfg, w).forced_logits(toks[b], n - 1)
            assert_model_close(lg_ref[0], lg[b], f"{name} s4 prefill n={n} row {b}")
            if nxt is not None:
                assert_model_close(lg_ref[1], nxt[b], f"{name} decode after s4 prefill n={n} row {b}")
    m.close()


def test_prefill_c_attn_hands_half_tiles_over_between_workgroups(zg, monkeypatch, every_linear_on_s4):
    """1.5 rounds of c_attn tiles (eight 1023-token prompts at 124M: 384 tiles on 256 CUs) run as whole tiles plus K halves of the
    last half round on ALL workgroups, the producer's accumulators handed to its consumer through memory (gemm_s4.hip, SK).
    Small stand-in with the same geometry: nano-char, 4 x 256 tokens = 4 x 6 tiles on 16 workgroups (16 whole + 8 shared)."""
    monkeypatch.setenv("ZGPT2_GEMM_WGS", "16")
    cfg = synth.CONFIGS["nano-char"]
    m, w = make(cfg, 77, batch=4)
    n = 256
    toks = np.stack([synth.rand_tokens(770 + b, n, cfg.vocab_size) for b in range(4)])
    for rep in range(3):  # (the flags carry the launch's epoch: repeated passes must not see an earlier pass's hand-over)
        lg = m.prefill(toks)
        for b in range(4):
            lg_ref = oracle.GPT(cfg, w).forced_logits(toks[b], n - 1)
            assert_model_close(lg_ref[0], lg[b], f"stream-K c_attn row {b} pass {rep}")
    m.close()


@pytest.mark.parametrize("kv_f16", [False, True])
def test_prefill_batched_rows_are_independent(zg, kv_f16):
    cfg = synth.CONFIGS["tiny"]
    m, w = make(cfg, 72, batch=3, kv_f16=kv_f16)
    n = 21
    toks = np.stack([synth.rand_tokens(720 + b, n + 1, cfg.vocab_size) for b in range(3)])
    lg = m.prefill(toks[:, :n])
    nxt = m.forward(n + 1, toks[:, n])
    for b in range(3):
        lg_ref = oracle.GPT(cfg, w).forced_logits(toks[b], n - 1)
        assert_model_close(lg_ref[0], lg[b], f"row {b} prefill")
        if kv_f16:  # fp16 cache: not the parity default, 1e-3 of the logit scale (test_gpt_gpu.py)
            rms = float(np.sqrt(np.mean(lg_ref[1].astype(np.float64) ** 2)))
            assert np.abs(nxt[b] - lg_ref[1]).max() <= 1e-3 * rms
        else:
            assert_model_close(lg_ref[1], nxt[b], f"row {b} decode after prefill")
    m.close()


@pytest.mark.parametrize("name", ["tiny-p24", "tiny3-p40", "124M-p48"])
def test_prefill_then_teacher_forced_logits_match_reference_gpt(zg, name):
    """Golden vectors of the reference's PyTorch GPT with long prompts (tests/golden/make_golden.py): the
    prompt goes through zg_gpt_prefill in one pass, every later position through GPT.forward with the
    reference's fed tokens; each step's logits must match the reference GPT's."""
    cfg, g = load_gpt(name)
    m, _ = make(cfg, int(g["weight_seed"]))
    n_prompt, n_steps = len(g["prompt"]), len(g["out_tokens"])
    m.prefill([g["prompt"]], compute_logits=False)
    worst = 0.0
    for s in range(n_prompt, n_steps):
        lg = m.forward(s + 1, [g["fed"][s]])
        worst = max(worst, assert_model_close(g["logits"][s - n_prompt], lg[0, g["logit_cols"]], f"{name} step {s}"))
    print(f"{name}: worst normalised rel err vs reference GPT after prefill {worst:.2e}")
    m.close()


@pytest.mark.parametrize("name", ["tiny", "tiny3", "nano-char", "124M", "tiny-p24", "tiny3-p40", "124M-p48"])
def test_generate_with_and_without_prefill_match_reference_gpt(zg, name):
    cfg, g = load_gpt(name)
    n_steps, n_prompt = len(g["out_tokens"]), len(g["prompt"])
    out = {}
    for prefill in (True, False):
        m, _ = make(cfg, int(g["weight_seed"]), prefill=prefill)
        ids = m.generate([g["prompt"]], n_steps)[0]
        assert np.array_equal(ids[:n_prompt], g["prompt"])
        assert_greedy_ids_match(g["out_tokens"][n_prompt:], ids[n_prompt:], g["top1"], g["top2"], f"{name} prefill={prefill}")
        out[prefill] = ids
        m.close()
    assert_greedy_ids_match(out[False][n_prompt:], out[True][n_prompt:], g["top1"], g["top2"], f"{name} prefill vs loop")


def test_generate_ragged_prompts_prefill_shared_prefix_length(zg):
    """Batched prompts of different lengths: positions below the shortest prompt are prefilled, the rest
    go through the decode loop; every row must equal an independent reference-style generation."""
    cfg = synth.CONFIGS["tiny"]
    m, w = make(cfg, 73, batch=4)
    lens = [9, 6, 17, 6]
    prompts = [synth.rand_tokens(730 + b, lens[b], cfg.vocab_size) for b in range(4)]
    ids = m.generate(prompts, cfg.context_size)
    for b in range(4):
        ids_ref, lg = oracle.GPT(cfg, w).generate_greedy(prompts[b], cfg.context_size, want_logits=True)
        top = np.sort(lg, axis=1)
        assert np.array_equal(ids[b, : lens[b]], prompts[b])
        assert_greedy_ids_match(ids_ref[lens[b]:], ids[b, lens[b]:], top[:, -1], top[:, -2], f"row {b}")
    # n_steps shorter than the prompts: nothing is generated, the prompt prefix comes back
    short = m.generate(prompts, 5)
    for b in range(4):
        assert np.array_equal(short[b], prompts[b][:5])
    m.close()


def test_prefill_124m_full_context_equals_decode_loop(zg):
    """BASELINE size: a 1023-token prompt prefilled in one pass vs fed token by token; the logits of the
    last position and of the following decode step must agree within the model tolerance, and the
    prefilled generation must continue with the same greedy ids."""
    cfg = synth.CONFIGS["124M"]
    w = synth.make_weights(cfg, seed=0, bf16=True)
    n = cfg.context_size - 1
    toks = synth.rand_tokens(74, n + 1, cfg.vocab_size)
    a = zgpt.GPT(cfg)
    a.load_weights(w)
    for s in range(n - 1):
        a.forward(s + 1, [toks[s]], compute_logits=False)
    lg_loop = a.forward(n, [toks[n - 1]])
    nxt_loop = a.forward(n + 1, [toks[n]])
    lg = a.prefill([toks[:n]])
    nxt = a.forward(n + 1, [toks[n]])
    assert np.isfinite(lg).all()
    assert_model_close(lg_loop[0], lg[0], "124M prefill 1023 vs loop")
    assert_model_close(nxt_loop[0], nxt[0], "124M decode after prefill vs loop")
    a.close()
    # oracle spot check at a length it finishes in seconds
    ref = oracle.GPT(cfg, w)
    k = 80
    lg_ref = ref.forced_logits(toks[: k + 1], k - 1)
    b = zgpt.GPT(cfg)
    b.load_weights(w)
    assert_model_close(lg_ref[0], b.prefill([toks[:k]])[0], "124M prefill 80 vs oracle")
    assert_model_close(lg_ref[1], b.forward(k + 1, [toks[k]])[0], "124M decode after prefill 80 vs oracle")
    b.close()


@pytest.mark.parametrize("weights_f32", [False, True])
def test_prefill_124m_eight_full_prompts_equal_one_prompt_passes(zg, weights_f32):
    """BASELINE config 3's prompt side at full size: eight 1023-token prompts in one pass — every Linear on the persistent
    four-wave GEMM (three / six plane pairs in one K loop, K slices + reduce, the stream-K hand-over of c_attn's 384 tiles at its
    real geometry), whole-row attention — against the same prompts passed one at a time, which run the 128-row GEMM family and
    split key ranges: last-position logits and the decode step on top of the caches, rows 0, 3 and 7."""
    cfg = synth.CONFIGS["124M"]
    w = synth.make_weights(cfg, seed=5, bf16=not weights_f32)
    n = cfg.context_size - 1
    toks = np.stack([synth.rand_tokens(790 + b, n + 1, cfg.vocab_size) for b in range(8)])
    m8 = zgpt.GPT(cfg, batch=8, weights_f32=weights_f32)
    m8.load_weights(w)
    before = zg.zg_debug_gemm_launches()
    lg8 = m8.prefill(toks[:, :n]).copy()
    assert zg.zg_debug_gemm_launches() - before >= 4 * cfg.n_layer, "the whole-prompt Linears did not run on gemm_s4"
    nxt8 = m8.forward(n + 1, toks[:, n]).copy()
    m8.close()
    assert np.isfinite(lg8).all() and np.isfinite(nxt8).all()
    m1 = zgpt.GPT(cfg, weights_f32=weights_f32)
    m1.load_weights(w)
    for b in (0, 3, 7):
        before = zg.zg_debug_gemm_launches()
        lg1 = m1.prefill([toks[b, :n]])
        assert zg.zg_debug_gemm_launches() == before, "a single prompt was expected on the 128-row GEMM family"
        assert_model_close(lg1[0], lg8[b], f"124M 8 x 1023 row {b} vs single pass (fp32 weights: {weights_f32})")
        assert_model_close(m1.forward(n + 1, [toks[b, n]])[0], nxt8[b], f"124M decode after 8 x 1023 row {b} (fp32 weights: {weights_f32})")
    m1.close()


@pytest.mark.parametrize("seed", range(12))
def test_prefill_shape_sweep_against_the_oracle(zg, seed):
    """Seeded sweep over (model, batch, prompt length, weight type, GEMM route): whatever the routing rules pick — 128-row kernels,
    persistent GEMM with 1-4 K slices, whole-row or split-range attention, merge kernel — one row's last-position logits and
    its next decode step against the oracle, and the other rows finite."""
    rng = np.random.default_rng(9000 + seed)
    name = ["tiny", "tiny3", "nano-char", "xl-slice", "medium-slice"][int(rng.integers(0, 5))]
    cfg = synth.CONFIGS[name]
    batch = int(rng.integers(1, 9))
    n = int(rng.integers(1, min(cfg.context_size, 200)))
    f32 = bool(rng.integers(0, 2))
    route = int(rng.integers(0, 3))  # 0 the library's rule, 1 everything on gemm_s4, 2 everything on the 128-row kernels
    w = synth.make_weights(cfg, seed=100 + seed, bf16=not f32)
    m = zgpt.GPT(cfg, batch=batch, weights_f32=f32)
    m.load_weights(w)
    toks = np.stack([synth.rand_tokens(9100 + 13 * seed + b, n + 1, cfg.vocab_size) for b in range(batch)])
    _lib.check(zg.zg_debug_prefill_route(route, 0))
    try:
        lg = m.prefill(toks[:, :n]).copy()
    finally:
        _lib.check(zg.zg_debug_prefill_route(0, 0))
    nxt = m.forward(n + 1, toks[:, n]) if n < cfg.context_size else None
    assert np.isfinite(lg).all()
    b = int(rng.integers(0, batch))
    what = f"sweep {seed}: {name} batch {batch} n {n} f32 {f32} route {route} row {b}"
    lg_ref = oracle.GPT(cfg, w).forced_logits(toks[b], n - 1)
    assert_model_close(lg_ref[0], lg[b], what)
    if nxt is not None:
        assert np.isfinite(nxt).all()
        assert_model_close(lg_ref[1], nxt[b], what + " (decode step on top)")
    m.close()


def test_prefill_errors(zg):
    cfg = synth.CONFIGS["tiny"]
    m = zgpt.GPT(cfg)
    with pytest.raises(_lib.ZgError):
        m.prefill([np.zeros(cfg.context_size + 1, np.uint64)])
    with pytest.raises(_lib.ZgError):
        m.prefill([[cfg.vocab_size]])
    m.close()
    f = zgpt.GPT(cfg, prefill=False)  # ZG_GPT_NO_PREFILL: no whole-prompt buffers in the arena
    with pytest.raises(_lib.ZgError):
        f.prefill([[1, 2, 3]])
    f.close()


@pytest.mark.parametrize("name,lengths", [("tiny", [1, 5, 33, 64]), ("tiny3", [7, 48]), ("xl-slice", [40])])
def test_prefill_with_fp32_weights(zg, name, lengths):
    """ZG_GPT_WEIGHTS_F32 handles (checkpoints that are not bf16-representable): the whole-prompt pass multiplies exact
    bf16 plane triples of BOTH operands (six plane products) and must meet the same tolerance against the fp32 oracle
    as the decode path; the decode step on top pins the cache contents."""
    cfg = synth.CONFIGS[name]
    w = synth.make_weights(cfg, seed=73, bf16=False)
    m = zgpt.GPT(cfg, weights_f32=True)
    m.load_weights(w)
    for n in lengths:
        toks = synth.rand_tokens(730 + n, min(n + 1, cfg.context_size), cfg.vocab_size)
        lg_ref = oracle.GPT(cfg, w).forced_logits(toks, n - 1)
        lg = m.prefill([toks[:n]])
        assert_model_close(lg_ref[0], lg[0], f"{name} fp32-weight prefill n={n}")
        if n < cfg.context_size:
            assert_model_close(lg_ref[1], m.forward(n + 1, [toks[n]])[0], f"{name} decode after fp32-weight prefill n={n}")
    # generate() uses the pass for the prompt and must reproduce the oracle's greedy tokens
    prompt = synth.rand_tokens(739, 9, cfg.vocab_size)
    n_steps = min(40, cfg.context_size)
    ids = m.generate([prompt], n_steps)[0]
    ids_ref, lgs = oracle.GPT(cfg, w).generate_greedy(prompt, n_steps, want_logits=True)
    top = np.sort(lgs, axis=1)
    assert_greedy_ids_match(ids_ref[9:], ids[9:], top[:, -1], top[:, -2], f"{name} fp32 weights generate")
    m.close()


@pytest.mark.parametrize("name,batch,n,wgs", [("tiny3", 2, 48, None), ("nano-char", 4, 256, 16), ("xl-slice", 1, 95, 5), ("medium-slice", 2, 80, None)])
def test_prefill_fp32_weights_on_the_persistent_four_wave_gemm(zg, monkeypatch, every_linear_on_s4, name, batch, n, wgs):
    """fp32 weights on gemm_s4: the six plane products a_i w_j (i + j <= 2) of the activation planes and the weight's three plane
    matrices are the plane pairs of ONE K loop (6 K / 64 K-steps per tile) instead of three passes over partial slabs.  Same
    checks as with bf16 weights, against the fp32 oracle; nano-char 4 x 256 on 16 workgroups also hands half tiles over."""
    if wgs:
        monkeypatch.setenv("ZGPT2_GEMM_WGS", str(wgs))
    cfg = synth.CONFIGS[name]
    w = synth.make_weights(cfg, seed=78, bf16=False)
    m = zgpt.GPT(cfg, batch=batch, weights_f32=True)
    m.load_weights(w)
    toks = np.stack([synth.rand_tokens(780 + 7 * b + n, min(n + 1, cfg.context_size), cfg.vocab_size) for b in range(batch)])
    before = zg.zg_debug_gemm_launches()
    lg = m.prefill(toks[:, :n])
    assert zg.zg_debug_gemm_launches() - before >= 4 * cfg.n_layer, "the whole-prompt Linears did not run on gemm_s4"
    nxt = m.forward(n + 1, toks[:, n]) if n < cfg.context_size else None
    for b in range(batch):
        lg_ref = oracle.GPT(cfg, w).forced_logits(toks[b], n - 1)
        assert_model_close(lg_ref[0], lg[b], f"{name} fp32-weight s4 prefill row {b}")
        if nxt is not None:
            assert_model_close(lg_ref[1], nxt[b], f"{name} decode after fp32-weight s4 prefill row {b}")
    m.close()


def test_prefill_two_plane_mode_is_inside_the_parity_bound(zg):
    """ZG_GPT_PREFILL_2PLANE: hi + mid planes of the activations only.  north_star's bound is 1e-3 relative; the
    two-plane error is ~2e-5 of the logit scale, so it is stated against the logit scale (not against the
    near-zero floor of assert_model_close, which the exact three-plane default meets)."""
    cfg = synth.CONFIGS["124M"]
    w = synth.make_weights(cfg, seed=74, bf16=True)
    toks = synth.rand_tokens(741, 96, cfg.vocab_size)
    m3 = zgpt.GPT(cfg)
    m3.load_weights(w)
    lg3 = m3.prefill([toks])[0]
    m3.close()
    m2 = zgpt.GPT(cfg, prefill_planes=2)
    m2.load_weights(w)
    lg2 = m2.prefill([toks])[0]
    m2.close()
    scale = float(np.abs(lg3).max())
    err = float(np.abs(lg2 - lg3).max()) / scale
    assert 0 < err < 1e-3, err
    assert int(np.argmax(lg2)) == int(np.argmax(lg3))
