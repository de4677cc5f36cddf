#!/usr/bin/env python3
"""Two-stream decode at GPT-2 XL (no side-stream prefetcher there): us per token with ZGPT2_DUAL=0 / 1, tokens must agree."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
import torch
import bench
from zig_gpt2_amd import _lib, gpt, synth
lib = _lib.load(); _lib.check(lib.zg_init(0))
stream = torch.cuda.Stream(); _lib.check(lib.zg_set_stream(stream.cuda_stream))
name = sys.argv[1] if len(sys.argv) > 1 else "xl"
cfg = synth.CONFIGS[name]
n = cfg.context_size
res = {}
for dual in (0, 1, 0, 1):
    os.environ["ZGPT2_DUAL"] = str(dual)
    m = gpt.GPT(cfg, batch=1)
    m.load_weights(bench.device_weights(cfg, 5))
    prompts = [synth.rand_tokens(11, 1, cfg.vocab_size)]
    try:
        m.generate_enqueue(prompts, n); ids = m.generate_fetch(n)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.generate_enqueue(prompts, n); m.generate_fetch(n)
        dt = time.perf_counter() - t0
        res.setdefault(dual, ids)
        print(f"{name} dual={dual}: {1e6*dt/n:.1f} us/token, {n/dt:.0f} tok/s  same ids as first run of this mode: {np.array_equal(res[dual], ids)}", flush=True)
    except Exception as e:
        print(f"{name} dual={dual}: FAILED {e}", flush=True)
    m.close()
if 0 in res and 1 in res: print("ids equal across modes:", np.array_equal(res[0], res[1]))
