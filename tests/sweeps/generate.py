#!/usr/bin/env python3
"""Random generate() configurations against the oracle's greedy loop: model, batch, ragged prompt lengths, steps, KV cache type
(fp32 / 24-bit: ids must match; fp16 is outside the bound and not swept), weight type, graph on / off, whole-prompt pass on / off,
L2 prefetcher on / off.  A differing id is accepted only at a numerical tie of the oracle's own top two logits (golden_io).
python tests/sweeps/generate.py [first_seed] [count] [max_steps = 96]"""
import os, sys, traceback
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "oracle"))
import numpy as np
import oracle
from golden_io import assert_greedy_ids_match
from zig_gpt2_amd import _lib, gpt as zgpt, synth

zg = _lib.load(); _lib.check(zg.zg_init(0))
first, count, max_steps = (int(v) for v in (sys.argv[1:4] + ["0", "60", "96"][len(sys.argv) - 1:]))
bad = []
for seed in range(first, first + count):
    rng = np.random.default_rng(5000 + seed)
    name = ["tiny", "tiny3", "nano-char", "xl-slice", "medium-slice"][int(rng.integers(0, 5))]
    cfg = synth.CONFIGS[name]
    batch = int(rng.integers(1, 9))
    f32 = bool(rng.integers(0, 3) == 0)
    kw = dict(weights_f32=f32, use_graph=bool(rng.integers(0, 4)), kv_b24=bool(rng.integers(0, 3) == 0), prefill=bool(rng.integers(0, 2)),
              prefetch=bool(rng.integers(0, 2)))
    if not kw["kv_b24"] and rng.integers(0, 5) == 0: kw["kv_f16"] = True  # (outside the parity bound: ids in range and mostly equal, no more)
    n_steps = int(rng.integers(2, min(cfg.context_size, max_steps) + 1))
    lens = [int(rng.integers(1, max(2, min(n_steps, 40)))) for _ in range(batch)]
    prompts = [synth.rand_tokens(5100 + 17 * seed + b, lens[b], cfg.vocab_size) for b in range(batch)]
    # a third of the configurations run as co-running prompt GROUPS (one handle per group, own streams, shared weights) instead of
    # one lock-step handle: same tokens expected (a separate generator, so that the configurations above keep their seeds)
    rng_g = np.random.default_rng(91000 + seed)
    divisors = [g for g in (2, 4, 8) if batch % g == 0]
    groups = int(rng_g.choice(divisors)) if divisors and rng_g.integers(0, 3) == 0 else 1
    what = f"seed {seed}: {name} batch {batch} groups {groups} steps {n_steps} lens {lens} {kw}"
    try:
        w = synth.make_weights(cfg, seed=200 + seed, bf16=not f32)
        m = zgpt.GPTGroups(cfg, batch, groups, **kw) if groups > 1 else zgpt.GPT(cfg, batch=batch, **kw)
        m.load_weights(w)
        ids = m.generate(prompts, n_steps)
        m.close()
        ref = oracle.GPT(cfg, w)
        for b in range(batch):
            ids_ref, lgs = ref.generate_greedy(prompts[b], n_steps, want_logits=True)
            top = np.sort(lgs, axis=1)
            gap_tol = 3e-3 if kw["kv_b24"] else 1e-4  # (24-bit cache: 4.7e-5 of the logit scale; ties are wider)
            assert np.array_equal(ids[b][:lens[b]], prompts[b])
            if kw.get("kv_f16"):
                assert (ids[b] < cfg.vocab_size).all() and np.mean(ids_ref[lens[b]:] == ids[b][lens[b]:]) > 0.5, what
                continue
            assert_greedy_ids_match(ids_ref[lens[b]:], ids[b][lens[b]:], top[:, -1], top[:, -2], what + f" row {b}", gap_tol=gap_tol)
    except Exception:
        bad.append(seed)
        print(what)
        traceback.print_exc(limit=2)
print(f"{count} configurations from seed {first}: {len(bad)} failed {bad}")
sys.exit(1 if bad else 0)
