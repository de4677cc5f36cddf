#!/usr/bin/env python3
"""GPT.sample (src/main.zig:198-207: forward, logits / temp, softmax, an index drawn against the running sum) over random models,
batches, temperatures and uniforms against the oracle: probabilities agree, the pick is the oracle's except where u * total
lands within 1e-6 of a boundary of the running sum; both sides are fed the device's picks.  Then the device loop
(zg_gpt_generate_sample) against the host loop over the per-token call with the same seed: identical tokens.  python tests/sweeps/sample.py [first_seed] [count]"""
import os, sys, traceback
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "oracle"))
import numpy as np
import oracle
from golden_io import assert_model_close
from zig_gpt2_amd import _lib, gpt as zgpt, synth

zg = _lib.load(); _lib.check(zg.zg_init(0))
first, count = (int(v) for v in (sys.argv[1:3] + ["0", "60"][len(sys.argv) - 1:]))
bad = []
for seed in range(first, first + count):
    rng = np.random.default_rng(2000 + seed)
    name = ["tiny", "tiny3", "nano-char", "medium-slice"][int(rng.integers(0, 4))]
    cfg = synth.CONFIGS[name]
    batch = int(rng.integers(1, 9))
    temp = float(rng.choice([0.3, 0.8, 1.0, 1.7, 5.0]))
    steps = int(rng.integers(1, min(cfg.context_size, 50)))
    what = f"seed {seed}: {name} batch {batch} temp {temp} steps {steps}"
    try:
        w = synth.make_weights(cfg, seed=500 + seed, bf16=True)
        m = zgpt.GPT(cfg, batch=batch)
        m.load_weights(w)
        ref = [oracle.GPT(cfg, w) for _ in range(batch)]
        toks = [int(t) for t in rng.integers(0, cfg.vocab_size, batch)]
        near = 0
        for s in range(steps):
            us = rng.random(batch).astype(np.float32)
            got, probs = m.sample(s + 1, toks, temp, uniforms=us, want_probs=True)
            for b in range(batch):
                exp_tok, exp_probs = ref[b].sample(s + 1, toks[b], np.float32(temp), float(us[b]))
                assert_model_close(exp_probs, probs[b], what + f" probs step {s} row {b}")
                assert abs(float(probs[b].sum(dtype=np.float64)) - 1.0) < 1e-5, what
                if int(got[b]) != exp_tok:
                    cdf = np.cumsum(exp_probs.astype(np.float64))
                    assert np.abs(cdf - us[b] * cdf[-1]).min() < 1e-6, (what, s, b, int(got[b]), exp_tok)
                    near += 1
            toks = [int(t) for t in got]
        assert near <= max(2, steps * batch // 20), (what, near)
        # the same model through the DEVICE loop (zg_gpt_generate_sample: sampler node in the captured step, uniforms from the counter
        # PRNG of the seed) against the host loop over the per-token call with that seed: identical tokens
        n_gen = min(cfg.context_size, steps + 8)
        prompts = [synth.rand_tokens(9000 + 31 * seed + b, 1 + (seed + b) % 3, cfg.vocab_size) for b in range(batch)]
        got = m.generate_sample(prompts, n_gen, temp, seed=seed)
        want = np.zeros_like(got)
        draws = [0] * batch
        min_np = min(len(p) for p in prompts)
        for s in range(n_gen):
            fed = [int(p[s]) if s < len(p) else (int(p[-1]) if s == len(p) else int(draws[b])) for b, p in enumerate(prompts)]
            if s >= min_np:
                draws = m.sample(s + 1, fed, temp, seed=seed)
            else:
                m.forward(s + 1, fed, compute_logits=False)
            for b, p in enumerate(prompts):
                want[b, s] = fed[b] if s < len(p) else draws[b]
        assert np.array_equal(got, want), (what, "device loop vs host loop", np.argwhere(got != want)[:3].tolist())
        m.close()
    except Exception:
        bad.append(seed)
        print(what)
        traceback.print_exc(limit=1)
print(f"{count} runs from seed {first}: {len(bad)} failed {bad}")
sys.exit(1 if bad else 0)
