#!/usr/bin/env python3
"""GPT.sample (src/main.zig:198-207: forward, logits / temp, softmax, an index drawn against the running sum) over random models,
batches, temperatures and uniforms against the oracle: probabilities agree, the pick is the oracle's except where u * total
lands within 1e-6 of a boundary of the running sum; both sides are fed the device's picks.  python tests/sweeps/sample.py [first_seed] [count]"""
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
        m.close()
    except Exception:
        bad.append(seed)
        print(what)
        traceback.print_exc(limit=1)
print(f"{count} runs from seed {first}: {len(bad)} failed {bad}")
sys.exit(1 if bad else 0)
