#!/usr/bin/env python3
"""Warm-cache per-kernel-class cost (graph chain of the same kernel) for a model config."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from zig_gpt2_amd import _lib, gpt, synth

name = sys.argv[1] if len(sys.argv) > 1 else "124M"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = _lib.load(); _lib.check(lib.zg_init(0))
cfg = synth.CONFIGS[name]
m = gpt.GPT(cfg, batch=batch)
w = synth.make_weights(cfg, seed=0, bf16=True); m.load_weights(w)
m.generate([synth.rand_tokens(b, 1, cfg.vocab_size) for b in range(batch)], cfg.context_size)
for cls, nm in enumerate(gpt.GPT.PROFILE_CLASSES[:7]):
    us, nb = m.time_kernel(cls, 1024)
    print(f"{nm:26s} {us:7.2f} us/launch   {nb/1e6:8.2f} MB  {nb/us/1e3 if us else 0:8.1f} GB/s")
