"""Independent prompt groups on one GPU (zg_gpt_create_ex / zg_gpt_generate_enqueue_many, gpt.GPTGroups): G handles of
n / G sequences, each on its own stream, all reading ONE weight region, must give — row for row — the tokens of n independent
reference-style generations (the oracle: src/main.zig:322-342 with greedy argmax), whatever G is.  Integer work: ids identical
(a mismatch is tolerated only on an oracle near-tie, golden_io.assert_greedy_ids_match).
"""
import ctypes as C

import numpy as np
import pytest

import oracle
from golden_io import assert_greedy_ids_match
from zig_gpt2_amd import _lib
from zig_gpt2_amd import gpt as zgpt
from zig_gpt2_amd import synth

pytestmark = pytest.mark.gpu


def oracle_rows(cfg, w, prompts, n_steps):
    rows = []
    for p in prompts:
        ids, lg = oracle.GPT(cfg, w).generate_greedy(p, n_steps, want_logits=True)
        top = np.sort(lg, axis=1)
        rows.append((ids, top[:, -1], top[:, -2]))
    return rows


@pytest.mark.parametrize("groups", [1, 2, 4, 8])
def test_groups_equal_independent_oracle_runs(zg, groups):
    cfg = synth.CONFIGS["tiny"]
    w = synth.make_weights(cfg, seed=31, bf16=True)
    prompts = [synth.rand_tokens(300 + b, 1 + (b * 5) % 7, cfg.vocab_size) for b in range(8)]  # ragged: some groups prefill, some do not
    m = zgpt.GPTGroups(cfg, 8, groups)
    m.load_weights(w)
    n_steps = cfg.context_size
    ids = m.generate(prompts, n_steps)
    ids2 = m.generate(prompts, n_steps)  # a second generation on the same handles: caches cleared, graphs replayed
    assert np.array_equal(ids, ids2)
    for b, (ids_ref, t1, t2) in enumerate(oracle_rows(cfg, w, prompts, n_steps)):
        n = len(prompts[b])
        assert np.array_equal(ids[b, :n], prompts[b])
        assert_greedy_ids_match(ids_ref[n:], ids[b, n:], t1, t2, f"G={groups} row {b}")
    m.close()


def test_groups_match_the_lock_step_batch_at_124m(zg):
    """GPT-2 124M, 8 one-token prompts, 96 steps: 4 groups of 2 against one handle of 8 (which the full-size tests hold to the
    oracle) — the decode kernels differ between batch sizes (GEMV / plane-fed MFMA), the tokens must not."""
    cfg = synth.CONFIGS["124M"]
    w = synth.make_weights(cfg, seed=0, bf16=True)
    prompts = [synth.rand_tokens(2000 + b, 1, cfg.vocab_size) for b in range(8)]
    one = zgpt.GPT(cfg, batch=8)
    one.load_weights(w)
    want = one.generate(prompts, 96)
    one.close()
    for groups in (4, 8):
        m = zgpt.GPTGroups(cfg, 8, groups)
        m.load_weights(w)
        got = m.generate(prompts, 96)
        m.close()
        diff = np.argwhere(got != want)
        assert diff.size == 0, f"G={groups}: first differing (row, step) {diff[:4].tolist()}"


def test_weight_sharing_rules(zg):
    cfg = synth.CONFIGS["tiny"]
    w = synth.make_weights(cfg, seed=3, bf16=True)
    owner = zgpt.GPT(cfg, batch=1, own_stream=True)
    child = zgpt.GPT(cfg, batch=2, share_weights_with=owner, own_stream=True, stream_priority=1)
    # weights go in through the owner only
    with pytest.raises(_lib.ZgError):
        child.load_weights(w)
    owner.load_weights(w)
    # a handle of another config / weight type cannot borrow
    with pytest.raises(_lib.ZgError):
        zgpt.GPT(synth.CONFIGS["tiny3"], share_weights_with=owner)
    with pytest.raises(_lib.ZgError):
        zgpt.GPT(cfg, weights_f32=True, share_weights_with=owner)
    with pytest.raises(_lib.ZgError):
        zgpt.GPT(cfg, share_weights_with=child)  # only an owner lends
    # the owner cannot go first
    assert zg.zg_gpt_destroy(owner.h) != 0
    # the child sees the owner's weights: same tokens as the owner
    p = synth.rand_tokens(5, 3, cfg.vocab_size)
    a = owner.generate([p], 32)[0]
    b = child.generate([p, p], 32)
    assert np.array_equal(a, b[0]) and np.array_equal(a, b[1])
    # reloading through the owner reaches the child (folded LayerNorm vectors included)
    w2 = synth.make_weights(cfg, seed=4, bf16=True)
    owner.load_weights(w2)
    ids_ref, _ = oracle.GPT(cfg, w2).generate_greedy(p, 32, want_logits=True)
    assert np.array_equal(child.generate([p, p], 32)[0][:8], ids_ref[:8])
    # two handles on ONE stream are refused by the turn-by-turn enqueue
    x, y = zgpt.GPT(cfg), zgpt.GPT(cfg)
    harr = (C.c_void_p * 2)(x.h, y.h)
    mat = np.zeros((2, 1), np.uint64)
    lens = np.ones(2, np.uint64)
    assert zg.zg_gpt_generate_enqueue_many(harr, 2, _lib.ptr(mat), 1, _lib.ptr(lens), 8) != 0
    for h in (x, y, child, owner):
        h.close()
