#!/usr/bin/env python3
"""Sweep of the prefetcher's knobs (read at every generate call) on one handle: us per token of a full-context
greedy generation.    python tools/pf_sweep.py 124M[:B] "LEAD=1,2,3 NSUB=2,4,8 SLEEP=1,2" """
import itertools, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from zig_gpt2_amd import _lib, gpt, synth

lib = _lib.load(); _lib.check(lib.zg_init(0))
stream = torch.cuda.Stream(); _lib.check(lib.zg_set_stream(stream.cuda_stream))
name, _, b = sys.argv[1].partition(":")
B = int(b or 1)
cfg = synth.CONFIGS[name]
axes = {}
for tok in (sys.argv[2] if len(sys.argv) > 2 else "").split():
    k, _, v = tok.partition("=")
    axes["ZGPT2_PF_" + k] = v.split(",")
prompts = [synth.rand_tokens(900 + i, 1, cfg.vocab_size) for i in range(B)]
m = gpt.GPT(cfg, batch=B)
m.generate(prompts, cfg.context_size)

def run():
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        m.generate(prompts, cfg.context_size)
        best = min(best, time.perf_counter() - t0)
    return round(best / (cfg.context_size - 1) * 1e6, 2)

os.environ["ZGPT2_PF_MODE"] = "2"
print(json.dumps({"off": run()}), flush=True)
os.environ["ZGPT2_PF_MODE"] = "1"
print(json.dumps({"dry": run()}), flush=True)
os.environ["ZGPT2_PF_MODE"] = "0"
keys = list(axes)
for combo in itertools.product(*[axes[k] for k in keys]):
    for k, v in zip(keys, combo):
        os.environ[k] = v
    print(json.dumps({**{k[9:]: v for k, v in zip(keys, combo)}, "us": run(), "exit": m.prefetch_stats()["exit"][0]}), flush=True)
os.environ["ZGPT2_PF_MODE"] = "2"
print(json.dumps({"off": run()}), flush=True)
m.close()
