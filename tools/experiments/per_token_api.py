#!/usr/bin/env python3
"""Tokens/s of the per-token entry points of the model tier at GPT-2 124M — the loop a caller with its own sampler runs
(src/main.zig:336-338: token = gpt.sample(s + 1, temp, token, state)) — against zg_gpt_generate_greedy (no host round trip):
zg_gpt_sample (forward + device sampler, one token id back), zg_gpt_forward with the logits copied to the host, zg_gpt_forward +
zg_gpt_argmax."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from zig_gpt2_amd import _lib, gpt, synth
lib = _lib.load(); _lib.check(lib.zg_init(0))
cfg = synth.CONFIGS["124M"]
m = gpt.GPT(cfg)
gen = torch.Generator(device="cuda"); gen.manual_seed(3)
w = {}
for name, shape, mean, _ in synth.tensor_specs(cfg):
    w[name] = ((torch.randn(shape, generator=gen, device="cuda") * 0.02 + mean).to(torch.bfloat16).to(torch.float32)).contiguous()
m.load_weights(w)
n = 512
res = {}
def loop(fn):
    tok = 11
    fn(1, tok)
    t0 = time.perf_counter()
    for s in range(n):
        tok = fn(s + 1, tok)
    return round(n / (time.perf_counter() - t0), 1)
res["sample_tok_s"] = loop(lambda T, tok: int(m.sample(T, [tok], 0.8, seed=5)[0]))
res["forward_logits_to_host_tok_s"] = loop(lambda T, tok: int(np.argmax(m.forward(T, [tok])[0])))
def fwd_argmax(T, tok):
    m.forward(T, [tok], want_logits=False)
    return int(m.argmax()[0])
res["forward_plus_argmax_call_tok_s"] = loop(fwd_argmax)
def fwd_only(T, tok):
    m.forward(T, [tok], want_logits=False)
    return tok
res["forward_no_copy_tok_s"] = loop(fwd_only)
def fwd_nologits(T, tok):
    m.forward(T, [tok], compute_logits=False)
    return tok
res["forward_without_lm_head_tok_s"] = loop(fwd_nologits)
t0 = time.perf_counter(); m.generate([[11]], n); res["generate_greedy_tok_s"] = round(n / (time.perf_counter() - t0), 1)
print(json.dumps(res))
