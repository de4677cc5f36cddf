#!/usr/bin/env python3
"""Microseconds per op-tier call (host buffers, as an unchanged src/main.zig makes them) for the ops of one GPT-2 124M decode
step, 300 calls each after 30 warm-ups.  ctypes adds ~2 us per call.  usage: python tools/experiments/op_tier_calls.py [T]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from zig_gpt2_amd import _lib, ops, synth

T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
E, H, V = 768, 12, 50257
lib = _lib.load(); _lib.check(lib.zg_init(0))
z = lambda n: np.zeros(n, np.float32)
rnd = lambda seed, n, std=0.02: synth.fill_normal(seed, n, 0, std)


def reg(a):
    _lib.check(lib.zg_register_tensor(_lib.ptr(a), a.size))
    return a


def timed(fn, n=300, warm=30):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return round((time.perf_counter() - t0) / n * 1e6, 2)


res = {"T": T}
x = rnd(1, E, 1.0)
ln = ops.LayerNorm(E, reg(rnd(2, E) + 1), reg(rnd(3, E)))
res["layernorm_768"] = timed(lambda: ln.forward(x))
for name, k, n in (("c_attn_768x2304", E, 3 * E), ("c_fc_768x3072", E, 4 * E), ("mlp_proj_3072x768", 4 * E, E), ("lm_head_768x50257", E, V)):
    lin = ops.Linear(k, n, reg(rnd(10 + n, k * n)), reg(rnd(11 + n, n)) if n != V else None)
    xi, yo = rnd(5, k, 1.0), z(n)
    res["linear_" + name] = timed(lambda: lin.forward(xi, yo), n=100 if n == V else 300)
h4 = rnd(6, 4 * E, 1.0)
res["gelu_3072"] = timed(lambda: ops.gelu(h4))
lg = rnd(7, V, 1.0)
res["softmax_50257"] = timed(lambda: ops.softmax(lg), n=100)
emb = ops.Embedding(E, reg(rnd(8, 1024 * E)))
idx, eo = np.array([5], np.uint64), z(E)
res["embedding_1x768"] = timed(lambda: emb.forward(idx, eo))
attn = ops.CausalSelfAttention(H, E, ops.Linear(E, 3 * E, reg(rnd(20, 3 * E * E)), reg(rnd(21, 3 * E))), ops.Linear(E, E, reg(rnd(22, E * E)), reg(rnd(23, E))))
kc, vc = rnd(24, 1024 * E, 0.5), rnd(25, 1024 * E, 0.5)
out, _qkv, _q, _k, _v, _a = z(E), z(3 * E), z(E), z(1024 * E), z(1024 * E), z(1024)
state = {"t": 0}


def attn_step():  # positions in order (the mirror is extended), wrapping at T
    t = state["t"] % T + 1
    state["t"] += 1
    attn.forward(t, x, kc[: t * E], vc[: t * E], out, _qkv, _q, _k[: t * E], _v[: t * E], _a[:t])


res[f"attn_forward_walk_to_{T}"] = timed(attn_step, n=T, warm=T)
res["attn_forward_repeat_T (re-upload)"] = timed(lambda: attn.forward(T, x, kc[: T * E], vc[: T * E], out, _qkv, _q, _k[: T * E], _v[: T * E], _a[:T]), n=50, warm=5)
per_token = 2 * res["embedding_1x768"] + 12 * (2 * res["layernorm_768"] + res[f"attn_forward_walk_to_{T}"] + res["linear_c_fc_768x3072"] + res["gelu_3072"] +
                                               res["linear_mlp_proj_3072x768"]) + res["layernorm_768"] + res["linear_lm_head_768x50257"]
res["sum_for_one_token_us"] = round(per_token, 1)
print(json.dumps(res))
