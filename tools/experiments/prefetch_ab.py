#!/usr/bin/env python3
"""A/B of the side-stream L2 prefetcher (zig_gpt2_amd/csrc/prefetch.hip): the same greedy generation with and
without it — identical ids required, wall time per token of the enqueue + fetch, and how the prefetcher ended.
    python tools/prefetch_ab.py [124M|124M:8|xl ...]      (env ZGPT2_PF_LEAD / ZGPT2_PF_NSUB / ZGPT2_PF_SLEEP)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from zig_gpt2_amd import _lib, gpt, synth

lib = _lib.load(); _lib.check(lib.zg_init(0))
stream = torch.cuda.Stream(); _lib.check(lib.zg_set_stream(stream.cuda_stream))

for spec in (sys.argv[1:] or ["124M", "124M:8", "xl"]):
    name, _, b = spec.partition(":")
    B = int(b or 1)
    cfg = synth.CONFIGS[name]
    rng = np.random.default_rng(5)
    w = {}
    for tname, shape, mean, _ in synth.tensor_specs(cfg):
        w[tname] = synth.round_bf16((rng.standard_normal(int(np.prod(shape)), dtype=np.float32) * np.float32(0.02) + np.float32(mean))).reshape(shape)
    prompts = [synth.rand_tokens(900 + i, 1, cfg.vocab_size) for i in range(B)]
    ids, res = {}, {}
    for on in (False, True, False, True):
        m = gpt.GPT(cfg, batch=B, prefetch=on)
        m.load_weights(w)
        m.generate(prompts, 64)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            out = m.generate(prompts, cfg.context_size)
            best = min(best, time.perf_counter() - t0)
        st = m.prefetch_stats()
        m.close()
        ids.setdefault(on, out)
        assert np.array_equal(ids[on], out)
        res.setdefault(on, []).append(round(best / (cfg.context_size - 1) * 1e6, 2))
        if on and os.environ.get("PF_VERBOSE"):
            print(json.dumps({"model": name, "batch": B, "stats": st}), flush=True)
    print(json.dumps({"model": name, "batch": B, "env": {k: v for k, v in os.environ.items() if k.startswith("ZGPT2_PF")}, "us_per_token_off": res[False], "us_per_token_on": res[True],
                      "identical_ids": bool(np.array_equal(ids[False], ids[True]))}), flush=True)
