#!/usr/bin/env python3
"""A/B of the two-stream decode (zg_gpt.dual_on) against the single-stream step: same tokens, logits difference, us per token.
  python tools/dual_ab.py [model] [steps]       (ZGPT2_TAG_SPIN_LIMIT bounds every poll: keep it small while debugging)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zig_gpt2_amd import _lib, gpt, synth
lib = _lib.load(); _lib.check(lib.zg_init(0))
stream = torch.cuda.Stream(); _lib.check(lib.zg_set_stream(stream.cuda_stream))
name = sys.argv[1] if len(sys.argv) > 1 else "124M"
cfg = synth.CONFIGS[name]
n = int(sys.argv[2]) if len(sys.argv) > 2 else cfg.context_size
w = synth.make_weights(cfg, seed=3, bf16=True)
res = {}
for dual in (0, 1):
    os.environ["ZGPT2_DUAL"] = "1" if dual else "0"
    m = gpt.GPT(cfg, batch=1)
    m.load_weights(w)
    prompts = [synth.rand_tokens(11, 3, cfg.vocab_size)]
    try:
        lg = m.forward(1, [int(prompts[0][0])])
        ids = m.generate(prompts, n)
        ts = []
        for r in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m.generate_enqueue(prompts, n); m.generate_fetch(n)
            ts.append(time.perf_counter() - t0)
        res[dual] = (ids, lg, min(ts))
        print(f"dual={dual}: {1e6*min(ts)/n:.1f} us/token, {n/min(ts):.0f} tok/s, prefetcher {m.prefetch_stats()['on']}", flush=True)
    except Exception as e:
        print(f"dual={dual}: FAILED {e}", flush=True)
    m.close()
if 0 in res and 1 in res:
    print("ids equal:", np.array_equal(res[0][0], res[1][0]), " max logit diff:", float(np.abs(res[0][1] - res[1][1]).max()), "of scale", float(np.abs(res[0][1]).max()))
