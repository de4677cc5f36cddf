#!/usr/bin/env python3
"""Random shapes through the op tier against float64 numpy and the oracle: Linear.forward (any in_features incl. > 8192, ragged out_features,
batch 1..40: GEMV kernels, K chunks, the matrix-core path from batch 16), LayerNorm, gelu, softmax, incremental CausalSelfAttention.forward,
scaled_dot_product_attention, split_qkv / transpose, Embedding.
python tests/sweeps/ops.py [first_seed] [count]"""
import os, sys, traceback
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'oracle'))
import numpy as np
import oracle
from zig_gpt2_amd import _lib, ops, synth

import torch
zg = _lib.load(); _lib.check(zg.zg_init(0))


def place(a, rng, may_register=False):
    """The array as the caller might hold it: host numpy, a device tensor, or (weights) a host array registered with the library."""
    k = int(rng.integers(0, 3))
    if a is None: return None
    if k == 1:
        t = torch.from_numpy(np.ascontiguousarray(a)).cuda(); torch.cuda.synchronize(); return t
    if k == 2 and may_register:
        _lib.check(zg.zg_register_tensor(a.ctypes.data, a.size))
    return a


def host(a):
    return a.cpu().numpy() if hasattr(a, "cpu") else a


first, count = (int(v) for v in (sys.argv[1:3] + ["0", "60"][len(sys.argv) - 1:]))
bad = []
for seed in range(first, first + count):
    rng = np.random.default_rng(3000 + seed)
    kind = int(rng.integers(0, 10))
    try:
        if kind <= 2:  # Linear
            in_f = int(rng.choice([int(rng.integers(1, 300)), int(rng.integers(300, 4200)), 64 * int(rng.integers(2, 60)), int(rng.integers(8193, 20000))]))
            out_f = int(rng.choice([int(rng.integers(1, 70)), int(rng.integers(70, 2500)), 64 * int(rng.integers(1, 50))]))
            batch = int(rng.choice([1, int(rng.integers(2, 9)), int(rng.integers(9, 41))]))
            if in_f * out_f > 40_000_000: out_f = max(1, 40_000_000 // in_f)
            w = synth.fill_normal(seed * 7 + 1, in_f * out_f, 0, 0.05).reshape(out_f, in_f)
            bias = synth.fill_normal(seed * 7 + 2, out_f, 0, 0.1) if rng.integers(0, 4) else None
            x = synth.fill_normal(seed * 7 + 3, batch * in_f, 0, 1.0)
            y = place(np.zeros(batch * out_f, np.float32), rng)
            ops.Linear(in_f, out_f, place(w, rng, True), place(bias, rng, True)).forward(place(x, rng), y)
            torch.cuda.synchronize(); y = host(y); _lib.check(zg.zg_unregister_all())
            ref = x.reshape(batch, in_f).astype(np.float64) @ w.astype(np.float64).T + (0 if bias is None else bias.astype(np.float64))
            scale = np.abs(ref).max() + 1e-30
            err = np.abs(y.reshape(batch, out_f) - ref).max() / scale
            what = f"Linear in {in_f} out {out_f} batch {batch} bias {bias is not None}"
            assert err < 3e-6, (what, err)
        elif kind == 3:
            n = int(rng.integers(1, 5000)); rows = int(rng.integers(1, 6))
            g, b = synth.fill_normal(seed + 5, n, 1, 0.2), synth.fill_normal(seed + 6, n, 0, 0.2)
            x = synth.fill_normal(seed + 7, rows * n, 0.3, 2.0)
            ref = x.reshape(rows, n).astype(np.float64)
            ref = (ref - ref.mean(1, keepdims=True)) / np.sqrt(ref.var(1, keepdims=True) + 1e-5) * g + b
            xd = place(x, rng)
            ops.LayerNorm(n, place(g, rng, True), place(b, rng, True)).forward(xd)
            torch.cuda.synchronize(); x = host(xd); _lib.check(zg.zg_unregister_all())
            what = f"LayerNorm n {n} rows {rows}"
            assert np.abs(x.reshape(rows, n) - ref).max() < 2e-5, what
        elif kind == 4:
            n = int(rng.integers(1, 100000))
            x = synth.fill_normal(seed + 8, n, 0, 3.0)
            r = x.astype(np.float64)
            ref = 0.5 * r * (1 + np.tanh(np.sqrt(2 / np.pi) * (r + 0.044715 * r ** 3)))
            xd = place(x, rng); ops.gelu(xd); torch.cuda.synchronize(); x = host(xd)
            what = f"gelu n {n}"
            assert np.abs(x - ref).max() < 2e-6, what
        elif kind == 6:  # CausalSelfAttention.forward over T incremental steps against the oracle's
            hds = int(rng.integers(1, 7)); hd = 64 if rng.integers(0, 2) else int(rng.choice([8, 24, 32, 40, 96, 128])); e = hd * hds; T = int(rng.integers(1, 330))
            caw = synth.fill_normal(seed + 10, 3 * e * e, 0, 0.08).reshape(3 * e, e); cab = synth.fill_normal(seed + 11, 3 * e, 0, 0.05)
            cpw = synth.fill_normal(seed + 12, e * e, 0, 0.08).reshape(e, e); cpb = synth.fill_normal(seed + 13, e, 0, 0.05)
            xs = synth.fill_normal(seed + 14, T * e, 0, 1.0).reshape(T, e)
            ref = oracle.CausalSelfAttention(hds, e, caw, cab, cpw, cpb, T)
            attn = ops.CausalSelfAttention(hds, e, ops.Linear(e, 3 * e, caw, cab), ops.Linear(e, e, cpw, cpb))
            z = lambda *sh: np.zeros(sh, np.float32)
            k_cache, v_cache = z(T * e), z(T * e)
            _qkv, _q, _k, _v, _attn = z(3 * e), z(e), z(T * e), z(T * e), z(T)
            what = f"attn heads {hds} head_dim {hd} T {T}"
            for st in range(T):
                out = z(e)
                attn.forward(st + 1, xs[st], k_cache[: (st + 1) * e], v_cache[: (st + 1) * e], out, _qkv, _q, _k[: (st + 1) * e], _v[: (st + 1) * e], _attn[: st + 1])
                exp = ref.forward(st + 1, xs[st])
                assert np.abs(out - exp).max() <= 2e-5 * max(1.0, np.abs(exp).max()), (what, st)
            assert np.abs(k_cache - ref.k_cache).max() < 1e-5 and np.abs(v_cache - ref.v_cache).max() < 1e-5, what
        elif kind == 7:  # scaled_dot_product_attention on head-major q / k / v
            hds = int(rng.integers(1, 13)); T = int(rng.integers(1, 700)); hd = 64 if rng.integers(0, 2) else int(rng.integers(1, 200))
            q = synth.fill_normal(seed + 15, hds * hd, 0, 1.0); k = synth.fill_normal(seed + 16, hds * T * hd, 0, 1.0); v = synth.fill_normal(seed + 17, hds * T * hd, 0, 1.0)
            out, _attn = np.zeros(hds * hd, np.float32), np.zeros(T, np.float32)
            ops.scaled_dot_product_attention(q, k, v, hds, T, hd, out, _attn)
            exp = oracle.sdpa(q, k, v, hds, T, hd)
            what = f"sdpa heads {hds} T {T} head_dim {hd}"
            assert np.abs(out - exp).max() <= 5e-6 * max(1.0, np.abs(exp).max()), what
        elif kind == 8:  # split_qkv + transpose
            hds = int(rng.integers(1, 13)); e = 64 * hds; T = int(rng.integers(1, 200)); idx = int(rng.integers(0, 3))
            x = synth.fill_normal(seed + 18, T * 3 * e, 0, 1.0)
            out = np.zeros(T * e, np.float32)
            ops.CausalSelfAttention(hds, e, None, None).split_qkv(T, x, idx, out)
            what = f"split_qkv / transpose heads {hds} T {T} idx {idx}"
            assert np.array_equal(out, oracle.split_qkv(e, T, x, idx)), what
            tr = np.zeros(T * e, np.float32)
            ops.CausalSelfAttention.transpose((T, hds, 64), out, tr)
            assert np.array_equal(tr, oracle.transpose(T, hds, 64, out)), what
        elif kind == 9:  # Embedding
            dim = int(rng.integers(1, 2000)); rows = int(rng.integers(1, 3000)); n = int(rng.integers(1, 40))
            w = synth.fill_normal(seed + 19, rows * dim, 0, 1.0).reshape(rows, dim)
            idxs = rng.integers(0, rows, n).astype(np.uint64)
            out = np.zeros(n * dim, np.float32)
            ops.Embedding(dim, w).forward(idxs, out)
            what = f"Embedding dim {dim} rows {rows} n {n}"
            assert np.array_equal(out.reshape(n, dim), w[idxs.astype(np.int64)]), what
        else:
            n = int(rng.integers(1, 60000))
            x = synth.fill_normal(seed + 9, n, 0, 4.0)
            r = x.astype(np.float64); r = np.exp(r - r.max()); ref = r / r.sum()
            xd = place(x, rng); ops.softmax(xd); torch.cuda.synchronize(); x = host(xd)
            what = f"softmax n {n}"
            assert np.abs(x - ref).max() < 1e-6 * max(1.0, ref.max() * 10), what
    except Exception:
        bad.append(seed)
        traceback.print_exc(limit=1)
print(f"{count} cases from seed {first}: {len(bad)} failed {bad}")
sys.exit(1 if bad else 0)
