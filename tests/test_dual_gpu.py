"""Two-stream decode of one sequence (ZGPT2_DUAL=1; api_gpt.hip "dual", opt-in — measured at parity with the side-stream
prefetcher): the kernels of a step run from two hipGraphs on two streams and hand the residual
stream over as (value, tag) granules.  Same arithmetic in the same order: tokens and logits must be
IDENTICAL to the single-stream step, and equal to the oracle's within the model tolerance."""
import numpy as np
import pytest

import oracle
from zig_gpt2_amd import gpt, synth
from zig_gpt2_amd._lib import ZgError

pytestmark = pytest.mark.gpu

# four Blocks of GPT-2-small-like shapes that the K-split kernels take (n_embed <= 1024, 4 n_embed >= 2048)
CFG = synth.GPTConfig(1031, 96, 4, 8, 512)


def _run(monkeypatch, dual, steps=96, spin=None):
    monkeypatch.setenv("ZGPT2_DUAL", "1" if dual else "0")
    if spin is not None:
        monkeypatch.setenv("ZGPT2_TAG_SPIN_LIMIT", str(spin))
    w = synth.make_weights(CFG, seed=17, bf16=True)
    m = gpt.GPT(CFG, batch=1)
    m.load_weights(w)
    prompt = synth.rand_tokens(5, 3, CFG.vocab_size)
    try:
        ids = m.generate([prompt], steps)
        lg = [m.forward(t, [int(prompt[0])]) for t in (1, 2, 70)]
        hid = m.hidden()
    finally:
        m.close()
    return w, prompt, ids, lg, hid


def test_dual_decode_is_identical_to_the_single_stream_step(zg, monkeypatch):
    w, prompt, ids0, lg0, hid0 = _run(monkeypatch, False)
    _, _, ids1, lg1, hid1 = _run(monkeypatch, True)
    assert np.array_equal(ids0, ids1)
    for a, b in zip(lg0, lg1):
        assert np.array_equal(a, b)
    assert np.array_equal(hid0, hid1)
    ids_ref, _ = oracle.GPT(CFG, w).generate_greedy(prompt, 96, want_logits=True)
    assert np.array_equal(ids_ref, ids1[0])


def test_dual_decode_timed_out_hand_over_fails_the_call(zg, monkeypatch):
    """A poll bound of zero: the first kernel that finds its producer still running gives up, raises the fault word, and the
    call that drains the streams must fail instead of returning tokens."""
    with pytest.raises(ZgError):
        _run(monkeypatch, True, steps=48, spin=0)
