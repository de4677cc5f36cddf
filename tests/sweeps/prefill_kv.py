#!/usr/bin/env python3
"""The whole-prompt pass with the narrow KV caches (24-bit: inside the parity bound; fp16: outside, opt-in) over random models,
batches and prompt lengths: the decode steps ON TOP of the prefilled cache against the oracle at 1e-3 / 1e-2 of the logit scale —
what the c_attn epilogue's cache append (fp32 -> 24-bit / fp16, every GEMM family) and the attention's cache reads must agree on.
python tests/sweeps/prefill_kv.py [first_seed] [count]"""
import os, sys, traceback
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "oracle"))
import numpy as np
import oracle
from zig_gpt2_amd import _lib, gpt as zgpt, synth

zg = _lib.load(); _lib.check(zg.zg_init(0))
first, count = (int(v) for v in (sys.argv[1:3] + ["0", "100"][len(sys.argv) - 1:]))
bad = []
for seed in range(first, first + count):
    rng = np.random.default_rng(6000 + seed)
    name = ["tiny", "tiny3", "nano-char", "xl-slice", "medium-slice"][int(rng.integers(0, 5))]
    cfg = synth.CONFIGS[name]
    batch = int(rng.integers(1, 9))
    n = int(rng.integers(1, min(cfg.context_size, 200) - 3))
    mode = "b24" if rng.integers(0, 2) else "f16"
    route = int(rng.integers(0, 3))
    f32 = bool(rng.integers(0, 4) == 0)
    what = f"seed {seed}: {name} batch {batch} n {n} kv {mode} route {route} f32w {f32}"
    try:
        w = synth.make_weights(cfg, seed=400 + seed, bf16=not f32)
        m = zgpt.GPT(cfg, batch=batch, weights_f32=f32, kv_b24=mode == "b24", kv_f16=mode == "f16")
        m.load_weights(w)
        toks = np.stack([synth.rand_tokens(6100 + 13 * seed + b, n + 3, cfg.vocab_size) for b in range(batch)])
        _lib.check(zg.zg_debug_prefill_route(route, 0))
        try:
            lg = m.prefill(toks[:, :n]).copy()
        finally:
            _lib.check(zg.zg_debug_prefill_route(0, 0))
        b = int(rng.integers(0, batch))
        ref = oracle.GPT(cfg, w).forced_logits(toks[b], n - 1)
        tol = 1e-3 if mode == "b24" else 1e-2
        steps = [lg] + [m.forward(n + 1 + j, toks[:, n + j]).copy() for j in range(3)]
        for j, got in enumerate(steps):
            assert np.isfinite(got).all(), what
            err = np.abs(got[b] - ref[j]).max() / np.abs(ref[j]).max()
            assert err < tol, (what, j, float(err))
        m.close()
    except Exception:
        bad.append(seed)
        print(what)
        traceback.print_exc(limit=1)
print(f"{count} cases from seed {first}: {len(bad)} failed {bad}")
sys.exit(1 if bad else 0)
