#!/usr/bin/env python3
"""us per token of a full-context greedy generation against the number of decode steps captured per hipGraph
(ZGPT2_GRAPH_STEPS, read at zg_gpt_create), identical ids required.    python tools/graph_steps_ab.py [model[:B]]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from zig_gpt2_amd import _lib, gpt, synth
lib = _lib.load(); _lib.check(lib.zg_init(0))
stream = torch.cuda.Stream(); _lib.check(lib.zg_set_stream(stream.cuda_stream))
name, _, b = (sys.argv[1] if len(sys.argv) > 1 else "124M").partition(":")
B = int(b or 1)
cfg = synth.CONFIGS[name]
rng = np.random.default_rng(5)
w = {t: synth.round_bf16(rng.standard_normal(int(np.prod(sh)), dtype=np.float32) * np.float32(0.02) + np.float32(mu)).reshape(sh)
     for t, sh, mu, _ in synth.tensor_specs(cfg)}
prompts = [synth.rand_tokens(900 + i, 3, cfg.vocab_size) for i in range(B)]
ref = None
for k in (1, 2, 4, 8, 16, 32, 1):
    os.environ["ZGPT2_GRAPH_STEPS"] = str(k)
    t0 = time.perf_counter()
    m = gpt.GPT(cfg, batch=B)
    create_s = time.perf_counter() - t0
    m.load_weights(w)
    m.generate(prompts, cfg.context_size)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        ids = m.generate(prompts, cfg.context_size)
        best = min(best, time.perf_counter() - t0)
    m.close()
    ref = ids if ref is None else ref
    print(json.dumps({"model": name, "batch": B, "steps_per_graph": k, "us_per_token": round(best / (cfg.context_size - 1) * 1e6, 2),
                      "create_s": round(create_s, 2), "same_ids": bool(np.array_equal(ref, ids))}), flush=True)
