#!/usr/bin/env python3
"""Generations of growing length with the prefetcher on, printing how it ended (robustness check, e.g. under
rocprofv3).    python tools/pf_short.py [steps ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zig_gpt2_amd import _lib, gpt, synth
lib = _lib.load(); _lib.check(lib.zg_init(0))
stream = torch.cuda.Stream(); _lib.check(lib.zg_set_stream(stream.cuda_stream))
cfg = synth.CONFIGS["124M"]
m = gpt.GPT(cfg, batch=1)
for n in [int(v) for v in sys.argv[1:]] or [8, 64, 64]:
    t0 = time.perf_counter()
    m.generate([synth.rand_tokens(1, 1, cfg.vocab_size)], n)
    dt = time.perf_counter() - t0
    st = m.prefetch_stats() if not os.environ.get("PF_NOSTATS") else {"exit": [], "jobs": []}
    print(json.dumps({"steps": n, "wall_ms": round(dt * 1e3, 2), "exit": st["exit"], "jobs": st["jobs"]}), flush=True)
m.close()
