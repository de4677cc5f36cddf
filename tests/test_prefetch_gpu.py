"""The side-stream L2 prefetcher of the decode chain (zig_gpt2_amd/csrc/prefetch.hip) on a real MI355X.  It may only
change speed: the generated ids must be those of a run without it (and of the oracle, which the model tests check
with the prefetcher on by default for one-sequence handles of small models); it must follow the whole chain, leave
when the generation ends, and take itself out when it finds no concurrency with the decode stream."""
import numpy as np
import pytest

import oracle
from golden_io import assert_greedy_ids_match
from zig_gpt2_amd import gpt as zgpt
from zig_gpt2_amd import synth

pytestmark = pytest.mark.gpu


def make(cfg, w, **kw):
    m = zgpt.GPT(cfg, **kw)
    m.load_weights(w)
    return m


@pytest.mark.parametrize("name,kw", [("nano-char", {}), ("nano-char", {"kv_f16": True}), ("nano-char", {"kv_b24": True}), ("nano-char", {"use_graph": False}),
                                     ("tiny3", {}), ("tiny3", {"prefill": False})])
def test_same_tokens_with_and_without_prefetcher(zg, name, kw):
    cfg = synth.CONFIGS[name]
    w = synth.make_weights(cfg, seed=21, bf16=True)
    prompt = synth.rand_tokens(211, 5, cfg.vocab_size)
    out = {}
    for on in (True, False):
        m = make(cfg, w, prefetch=on, **kw)
        for _ in range(2):  # the second call starts behind the first call's prefetcher
            ids = m.generate([prompt], cfg.context_size)[0]
        st = m.prefetch_stats()
        m.close()
        assert st["on"] == on
        out[on] = ids
    assert np.array_equal(out[True], out[False])
    if not (kw.get("kv_f16") or kw.get("kv_b24")):  # (narrow caches: ids are compared between the two runs only)
        ref, lg = oracle.GPT(cfg, w).generate_greedy(prompt, cfg.context_size, want_logits=True)
        top = np.sort(lg, axis=1)
        assert_greedy_ids_match(ref[len(prompt):], out[True][len(prompt):], top[:, -1], top[:, -2], f"{name} with prefetcher")


def test_prefetcher_follows_the_chain_and_leaves(zg):
    cfg = synth.CONFIGS["nano-char"]
    w = synth.make_weights(cfg, seed=22, bf16=True)
    m = make(cfg, w)
    steps = cfg.context_size
    m.generate([synth.rand_tokens(221, 1, cfg.vocab_size)], steps)
    st = m.prefetch_stats()
    m.close()
    assert st["on"] and not st["stalled"]
    assert sum(st["workgroups"]) == 8 * 12 and min(st["workgroups"]) >= 1, st  # 12 per XCD when dealt round robin
    assert st["exit"] == [1] * 8, st  # all left on the stop written behind the last step
    launches = steps * (2 + 5 * cfg.n_layer)
    assert min(st["jobs"]) > 0.5 * launches, (st, launches)  # it kept up with the chain


def test_stalled_prefetcher_takes_itself_out(zg, monkeypatch):
    """No progress within the idle limit (what a tool that serialises the streams produces): it leaves by itself, the
    generation is unaffected, and the handle stops launching it."""
    cfg = synth.CONFIGS["tiny3"]
    w = synth.make_weights(cfg, seed=23, bf16=True)
    prompt = synth.rand_tokens(231, 3, cfg.vocab_size)
    m0 = make(cfg, w, prefill=False, prefetch=False)
    ref = m0.generate([prompt], cfg.context_size)[0]
    m0.close()
    monkeypatch.setenv("ZGPT2_PF_IDLE", "0")
    m = make(cfg, w, prefill=False)
    ids = m.generate([prompt], cfg.context_size)[0]
    st = m.prefetch_stats()
    assert np.array_equal(ids, ref)
    assert 2 in st["exit"], st
    ids = m.generate([prompt], cfg.context_size)[0]  # this call sees how the last one ended
    assert np.array_equal(ids, ref)
    assert m.prefetch_stats()["stalled"]
    monkeypatch.delenv("ZGPT2_PF_IDLE")
    ids = m.generate([prompt], cfg.context_size)[0]
    assert np.array_equal(ids, ref) and m.prefetch_stats()["stalled"]
    m.close()


def test_stalled_prefetcher_is_tried_again(zg, monkeypatch):
    """One idle-limit exit is a strike, not a verdict (a busy host produces the same exit): after eight generate calls
    without it the handle launches it again; same tokens throughout."""
    cfg = synth.CONFIGS["tiny3"]
    w = synth.make_weights(cfg, seed=24, bf16=True)
    prompt = synth.rand_tokens(241, 3, cfg.vocab_size)
    monkeypatch.setenv("ZGPT2_PF_IDLE", "0")
    m = make(cfg, w, prefill=False)
    ref = m.generate([prompt], cfg.context_size)[0]          # stalls (idle limit 0)
    monkeypatch.delenv("ZGPT2_PF_IDLE")
    for _ in range(8):                                       # sees the stall: sits these out
        assert np.array_equal(m.generate([prompt], cfg.context_size)[0], ref)
        assert m.prefetch_stats()["stalled"]
    assert np.array_equal(m.generate([prompt], cfg.context_size)[0], ref)  # tried again, with the default idle limit
    st = m.prefetch_stats()
    assert st["on"] and not st["stalled"] and st["exit"] == [1] * 8, st
    m.close()


def test_default_policy(zg):
    """On for one sequence and Linears of a few MB, off where it was measured not to pay (8 prompts, XL-sized layers)."""
    small, wide = synth.CONFIGS["tiny"], synth.CONFIGS["xl-slice"]
    for cfg, kw, want in ((small, {}, True), (small, {"batch": 8}, False), (wide, {}, False), (small, {"prefetch": False}, False)):
        m = zgpt.GPT(cfg, **kw)
        assert m.prefetch_stats()["on"] == want, (cfg, kw)
        m.close()
