#!/usr/bin/env python3
"""Stateful fuzz of ONE model handle: a random sequence of generate() calls, token-at-a-time forward loops and whole-prompt passes
followed by decode steps — different prompt lengths, rows, step counts, each starting a new sequence on the same handle (graphs
per 64-position bucket, epochs of the tagged hand-overs, the side-stream prefetcher and the caches all carry state between calls) —
every result against a fresh oracle.  python tests/sweeps/session.py [first_seed] [count]"""
import os, sys, traceback
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "oracle"))
import numpy as np
import oracle
from golden_io import assert_greedy_ids_match, assert_model_close
from zig_gpt2_amd import _lib, gpt as zgpt, synth

zg = _lib.load(); _lib.check(zg.zg_init(0))
first, count = (int(v) for v in (sys.argv[1:3] + ["0", "40"][len(sys.argv) - 1:]))
bad = []
for seed in range(first, first + count):
    rng = np.random.default_rng(8000 + seed)
    name = ["tiny", "tiny3", "nano-char", "xl-slice", "medium-slice"][int(rng.integers(0, 5))]
    cfg = synth.CONFIGS[name]
    batch = int(rng.integers(1, 9))
    f32 = bool(rng.integers(0, 4) == 0)
    kw = dict(weights_f32=f32, use_graph=bool(rng.integers(0, 5)), prefetch=bool(rng.integers(0, 2)))
    w = synth.make_weights(cfg, seed=300 + seed, bf16=not f32)
    m = zgpt.GPT(cfg, batch=batch, **kw)
    m.load_weights(w)
    log = []
    try:
        for op_i in range(int(rng.integers(3, 9))):
            op = int(rng.integers(0, 3))
            cmax = min(cfg.context_size, 150)
            if op == 0:  # generate
                n_steps = int(rng.integers(2, cmax + 1))
                lens = [int(rng.integers(1, max(2, min(n_steps, 60)))) for _ in range(batch)]
                prompts = [synth.rand_tokens(int(rng.integers(1, 1 << 30)), lens[b], cfg.vocab_size) for b in range(batch)]
                log.append(f"generate steps {n_steps} lens {lens}")
                ids = m.generate(prompts, n_steps)
                b = int(rng.integers(0, batch))
                ids_ref, lgs = oracle.GPT(cfg, w).generate_greedy(prompts[b], n_steps, want_logits=True)
                top = np.sort(lgs, axis=1)
                assert_greedy_ids_match(ids_ref[lens[b]:], ids[b][lens[b]:], top[:, -1], top[:, -2], f"seed {seed} op {op_i} row {b}")
            elif op == 1:  # token-at-a-time loop with forced tokens
                n = int(rng.integers(1, cmax + 1))
                toks = np.stack([synth.rand_tokens(int(rng.integers(1, 1 << 30)), n, cfg.vocab_size) for _ in range(batch)])
                log.append(f"forward loop {n}")
                check_at = set(int(v) for v in rng.integers(0, n, 3)) | {n - 1}
                b = int(rng.integers(0, batch))
                ref = oracle.GPT(cfg, w).forced_logits(toks[b], 0)
                for s in range(n):
                    want = s in check_at
                    lg = m.forward(s + 1, toks[:, s], compute_logits=want)
                    if want:
                        assert_model_close(ref[s], lg[b], f"seed {seed} op {op_i} loop step {s} row {b}")
            else:  # whole-prompt pass, then forced decode steps on top
                n = int(rng.integers(1, cmax)); k = int(rng.integers(1, min(6, cfg.context_size - n) + 1))
                toks = np.stack([synth.rand_tokens(int(rng.integers(1, 1 << 30)), n + k, cfg.vocab_size) for _ in range(batch)])
                log.append(f"prefill {n} + {k} steps")
                b = int(rng.integers(0, batch))
                ref = oracle.GPT(cfg, w).forced_logits(toks[b], n - 1)
                lg = m.prefill(toks[:, :n])
                assert_model_close(ref[0], lg[b], f"seed {seed} op {op_i} prefill row {b}")
                for j in range(k):
                    lg = m.forward(n + 1 + j, toks[:, n + j])
                    assert_model_close(ref[1 + j], lg[b], f"seed {seed} op {op_i} step {j} after prefill row {b}")
    except Exception:
        bad.append(seed)
        print(f"seed {seed}: {name} batch {batch} {kw}: {log}")
        traceback.print_exc(limit=2)
    m.close()
print(f"{count} sessions from seed {first}: {len(bad)} failed {bad}")
sys.exit(1 if bad else 0)
