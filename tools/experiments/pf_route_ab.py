#!/usr/bin/env python3
"""Whole-prompt pass with the library's routing threshold (gemm_s4 from 192 tiles x K slices) against lower ones
(zg_debug_prefill_route) — where the threshold of prefill.hip (s4_route) should sit.
usage: python tools/experiments/pf_route_ab.py [--weights-f32] batch [batch ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from zig_gpt2_amd import _lib, gpt as zgpt, synth

f32 = "--weights-f32" in sys.argv
batches = [int(a) for a in sys.argv[1:] if not a.startswith("--")]
cfg = synth.CONFIGS["124M"]
w = synth.make_weights(cfg, seed=0, bf16=not f32)
lib = _lib.load(); _lib.check(lib.zg_init(0))
for B in batches:
    m = zgpt.GPT(cfg, batch=B, weights_f32=f32)
    m.load_weights(w)
    toks = np.stack([synth.rand_tokens(100 + b, 1023, cfg.vocab_size) for b in range(B)])
    row = {"batch": B, "weights": "f32" if f32 else "bf16"}
    for name, force in (("library", 0), ("from_48", 48), ("from_96", 96), ("from_128", 128), ("library2", 0)):
        _lib.check(lib.zg_debug_prefill_route(force, 0))
        for _ in range(3): m.prefill(toks)
        t0 = time.perf_counter()
        for _ in range(10): m.prefill(toks)
        row[name + "_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
    _lib.check(lib.zg_debug_prefill_route(0, 0))
    print(json.dumps(row), flush=True)
    m.close()
