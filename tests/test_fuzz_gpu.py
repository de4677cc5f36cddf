"""Seeded differential sweep of the model tier against the CPU oracle over irregular model shapes
(head counts that are not powers of two, odd vocabularies and contexts, every batch size, ragged
prompts): greedy generation (prefill + batched decode), teacher-forced logits and the prefill pass."""
import numpy as np
import pytest

import oracle
from golden_io import assert_greedy_ids_match, assert_model_close
from zig_gpt2_amd import gpt as zgpt
from zig_gpt2_amd import synth

pytestmark = pytest.mark.gpu

CASES = [
    # vocab, ctx, layers, heads, batch
    (97, 40, 1, 1, 1), (259, 33, 2, 3, 2), (1000, 70, 1, 5, 3), (513, 48, 2, 7, 4), (77, 130, 1, 9, 5),
    (2049, 36, 1, 11, 6), (300, 65, 2, 13, 7), (4099, 34, 1, 17, 8), (129, 257, 1, 4, 2), (640, 50, 1, 20, 8),
    (50257, 24, 1, 12, 3), (33, 96, 3, 6, 8),
]


@pytest.mark.parametrize("case", CASES, ids=[f"V{c[0]}_C{c[1]}_L{c[2]}_H{c[3]}_B{c[4]}" for c in CASES])
def test_random_shapes(zg, case):
    vocab, ctx, layers, heads, batch = case
    cfg = synth.GPTConfig(vocab, ctx, layers, heads, 64 * heads)
    seed = vocab * 7 + heads
    w = synth.make_weights(cfg, seed=seed, bf16=True)
    m = zgpt.GPT(cfg, batch=batch)
    m.load_weights(w)
    lens = [1 + (seed + 5 * b) % min(12, ctx - 2) for b in range(batch)]
    prompts = [synth.rand_tokens(seed + 100 + b, lens[b], vocab) for b in range(batch)]
    n_steps = min(ctx, 28)
    ids = m.generate(prompts, n_steps)
    for b in range(batch):
        ids_ref, lg = oracle.GPT(cfg, w).generate_greedy(prompts[b], n_steps, want_logits=True)
        top = np.sort(lg, axis=1)
        assert np.array_equal(ids[b, : lens[b]], prompts[b][:n_steps])
        assert_greedy_ids_match(ids_ref[lens[b]:], ids[b, lens[b]:], top[:, -1], top[:, -2], f"{case} row {b}")
    n = min(ctx - 1, 19)
    toks = np.stack([synth.rand_tokens(seed + 200 + b, n + 1, vocab) for b in range(batch)])
    lg = m.prefill(toks[:, :n])
    nxt = m.forward(n + 1, toks[:, n])
    for b in (0, batch - 1):
        lg_ref = oracle.GPT(cfg, w).forced_logits(toks[b], n - 1)
        assert_model_close(lg_ref[0], lg[b], f"{case} prefill row {b}")
        assert_model_close(lg_ref[1], nxt[b], f"{case} decode after prefill row {b}")
    m.close()
