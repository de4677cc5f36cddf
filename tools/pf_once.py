#!/usr/bin/env python3
"""Two full-context greedy generations of one handle (for rocprofv3 --kernel-trace --stats A/B of the prefetcher:
ZGPT2_PF_MODE=2 is 'off').    python tools/pf_once.py 124M[:B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zig_gpt2_amd import _lib, gpt, synth
lib = _lib.load(); _lib.check(lib.zg_init(0))
stream = torch.cuda.Stream(); _lib.check(lib.zg_set_stream(stream.cuda_stream))
name, _, b = (sys.argv[1] if len(sys.argv) > 1 else "124M").partition(":")
B = int(b or 1)
cfg = synth.CONFIGS[name]
prompts = [synth.rand_tokens(900 + i, 1, cfg.vocab_size) for i in range(B)]
m = gpt.GPT(cfg, batch=B)
for _ in range(2):
    t0 = time.perf_counter()
    m.generate(prompts, cfg.context_size)
    print(name, B, "us/token", round((time.perf_counter() - t0) / (cfg.context_size - 1) * 1e6, 2), flush=True)
m.close()
