"""What ranks 1..N-1 of the multi-GPU configuration do, on one GPU: a handle that NEVER loads weights receives the
bytes of another handle's `zg_gpt_weight_arena` by a device copy — exactly what `shard.broadcast_weights` leaves on a
receiving rank (bench.py) — and must then generate the sender's tokens, position for position.  Also: a receiver
that has already run with other weights (its folded-LayerNorm vectors belong to those), a sender that has not run
yet (its folded vectors must be valid before its bytes leave), fp32-weight handles (whose bf16 planes for the
whole-prompt GEMMs travel inside the region), and the fp32-weight mode at the BASELINE size — the mode that keeps
real, non-bf16-representable GPT-2 checkpoints (reference src/main.zig:210-314, load_gpt) inside north_star's 1e-3.
"""
import numpy as np
import pytest
import torch

import oracle
from golden_io import assert_greedy_ids_match, assert_model_close
from zig_gpt2_amd import gpt as zgpt
from zig_gpt2_amd import synth

pytestmark = pytest.mark.gpu


class _DevMem:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def arena_tensor(m):
    p, n = m.weight_arena()
    return torch.as_tensor(_DevMem(p, n), device="cuda")


def transfer(src, dst):
    """The broadcast, as seen by one receiving rank: the sender's weight region lands in the receiver's."""
    a, b = arena_tensor(src), arena_tensor(dst)
    assert a.numel() == b.numel()
    torch.cuda.synchronize()
    b.copy_(a)
    torch.cuda.synchronize()


def fast_weights(cfg, seed, bf16=True):
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape, mean, _ in synth.tensor_specs(cfg):
        v = rng.standard_normal(int(np.prod(shape)), dtype=np.float32)
        v *= np.float32(0.02)
        v += np.float32(mean)
        w[name] = (synth.round_bf16(v) if bf16 else v).reshape(shape)
    return w


@pytest.mark.parametrize("name,batch,kw", [("tiny", 1, {}), ("tiny", 3, {}), ("tiny3", 1, {"weights_f32": True}),
                                           ("tiny3", 2, {"weights_f32": True}), ("124M", 1, {}), ("124M", 8, {}),
                                           ("tiny", 1, {"use_graph": False, "prefill": False})])
def test_receiver_generates_the_senders_tokens(zg, name, batch, kw):
    cfg = synth.CONFIGS[name]
    w = fast_weights(cfg, 41, bf16=not kw.get("weights_f32"))
    steps = min(cfg.context_size, 192)
    prompts = [synth.rand_tokens(410 + b, 1 + (b % 3) * 2, cfg.vocab_size) for b in range(batch)]
    a = zgpt.GPT(cfg, batch=batch, **kw)
    a.load_weights(w)               # sender: loaded, has not run (its folded LayerNorm vectors are made on demand)
    b = zgpt.GPT(cfg, batch=batch, **kw)  # receiver: created, never loaded
    transfer(a, b)
    ids_b = b.generate(prompts, steps)
    ids_a = a.generate(prompts, steps)
    assert np.array_equal(ids_a, ids_b), f"{name} x{batch}: receiver differs from sender"
    lg_a = a.forward(1, [int(p[0]) for p in prompts])
    lg_b = b.forward(1, [int(p[0]) for p in prompts])
    assert np.array_equal(lg_a, lg_b)
    # whole-prompt pass on the receiver (fp32 handles: the planes inside the region)
    if kw.get("prefill", True):
        toks = np.stack([synth.rand_tokens(420 + i, 12, cfg.vocab_size) for i in range(batch)])
        assert np.array_equal(a.prefill(toks), b.prefill(toks))
    a.close()
    b.close()


@pytest.mark.parametrize("kw", [{}, {"weights_f32": True}, {"batch": 4}])
def test_receiver_that_already_ran_with_other_weights(zg, kw):
    """the folded c2 / c3 vectors live inside the region; a receiver that has generated before must
    re-derive them from what arrives, and a sender that never ran must fold before its bytes are read."""
    cfg = synth.CONFIGS["tiny3"]
    batch = kw.get("batch", 1)
    bf16 = not kw.get("weights_f32")
    w1, w2 = fast_weights(cfg, 51, bf16), fast_weights(cfg, 52, bf16)
    prompts = [synth.rand_tokens(510 + b, 2, cfg.vocab_size) for b in range(batch)]
    b = zgpt.GPT(cfg, **kw)
    b.load_weights(w2)
    other = b.generate(prompts, cfg.context_size)       # the receiver has run: ln_folded is set for w2
    a = zgpt.GPT(cfg, **kw)
    a.load_weights(w1)                                   # the sender has NOT run
    transfer(a, b)
    got = b.generate(prompts, cfg.context_size)
    exp = a.generate(prompts, cfg.context_size)
    assert np.array_equal(got, exp)
    assert not np.array_equal(other, exp)                # the two weight sets really differ in their tokens
    ref, lg = oracle.GPT(cfg, w1).generate_greedy(prompts[0], cfg.context_size, want_logits=True)
    top = np.sort(lg, axis=1)
    assert_greedy_ids_match(ref[2:], got[0, 2:], top[:, -1], top[:, -2], "receiver vs oracle")
    # and back again: the sender receives the other set
    c = zgpt.GPT(cfg, **kw)
    c.load_weights(w2)
    transfer(c, a)
    assert np.array_equal(a.generate(prompts, cfg.context_size), other)
    for m in (a, b, c):
        m.close()


@pytest.mark.parametrize("batch", [1, 8])
def test_fp32_weights_at_124m(zg, batch):
    """ZG_GPT_WEIGHTS_F32 at the BASELINE size on weights that are NOT bf16-representable: all 1024 positions,
    hipGraph replay == eager launches, the first 64 steps and the position-1 logits against the oracle."""
    cfg = synth.CONFIGS["124M"]
    w = fast_weights(cfg, 61, bf16=False)
    assert not np.array_equal(w["h0.c_fc_w"], synth.round_bf16(w["h0.c_fc_w"].ravel()).reshape(w["h0.c_fc_w"].shape))
    ctx = cfg.context_size
    p0 = synth.rand_tokens(611, 1, cfg.vocab_size)
    prompts = [p0] * (batch - 1) + [synth.rand_tokens(612, 3, cfg.vocab_size)] if batch > 1 else [p0]
    m = zgpt.GPT(cfg, batch=batch, weights_f32=True)
    m.load_weights(w)
    ids = m.generate(prompts, ctx)
    lg_dev = m.forward(1, [int(p[0]) for p in prompts])
    m.close()
    assert ids.shape == (batch, ctx) and int(ids.max()) < cfg.vocab_size
    for b in range(1, batch - 1):
        assert np.array_equal(ids[0], ids[b])
    e = zgpt.GPT(cfg, batch=batch, weights_f32=True, use_graph=False)
    e.load_weights(w)
    ids_e = e.generate(prompts, ctx)
    e.close()
    assert np.array_equal(ids, ids_e), "graph and eager runs differ"
    n = 64
    for b in sorted({0, batch - 1}):
        ref = oracle.GPT(cfg, w)
        ids_ref, lg = ref.generate_greedy(prompts[b], n, want_logits=True)
        top = np.sort(lg, axis=1)
        k = len(prompts[b])
        assert_greedy_ids_match(ids_ref[k:], ids[b, k:n], top[:, -1], top[:, -2], f"124M fp32 weights row {b}")
        assert_model_close(oracle.GPT(cfg, w).forward(1, int(prompts[b][0])), lg_dev[b], f"124M fp32 logits row {b}")


@pytest.mark.parametrize("batch", [2, 5, 8])
def test_medium_width_batched(zg, batch):
    """E = 1024 (GPT-2 medium's width) with 2..8 sequences: the wave-per-tile lm_head at K = 1024 needs more than
    64 KiB of dynamic LDS."""
    cfg = synth.CONFIGS["medium-slice"]
    w = fast_weights(cfg, 71)
    prompts = [synth.rand_tokens(710 + b, 1 + b % 4, cfg.vocab_size) for b in range(batch)]
    m = zgpt.GPT(cfg, batch=batch)
    m.load_weights(w)
    ids = m.generate(prompts, cfg.context_size)
    m.close()
    for b in (0, batch - 1):
        ref, lg = oracle.GPT(cfg, w).generate_greedy(prompts[b], cfg.context_size, want_logits=True)
        top = np.sort(lg, axis=1)
        n = len(prompts[b])
        assert_greedy_ids_match(ref[n:], ids[b, n:], top[:, -1], top[:, -2], f"medium-slice x{batch} row {b}")
