#!/usr/bin/env python3
"""Random shapes through the op tier against float64 numpy: Linear.forward (any in_features incl. > 8192, ragged out_features,
batch 1..40: GEMV kernels, K chunks, the matrix-core path from batch 16), LayerNorm, gelu, softmax.
python tools/fuzz_ops.py [first_seed] [count]"""
import os, sys, traceback
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
from zig_gpt2_amd import _lib, ops, synth

zg = _lib.load(); _lib.check(zg.zg_init(0))
first, count = (int(v) for v in (sys.argv[1:3] + ["0", "60"][len(sys.argv) - 1:]))
bad = []
for seed in range(first, first + count):
    rng = np.random.default_rng(3000 + seed)
    kind = int(rng.integers(0, 6))
    try:
        if kind <= 2:  # Linear
            in_f = int(rng.choice([int(rng.integers(1, 300)), int(rng.integers(300, 4200)), 64 * int(rng.integers(2, 60)), int(rng.integers(8193, 20000))]))
            out_f = int(rng.choice([int(rng.integers(1, 70)), int(rng.integers(70, 2500)), 64 * int(rng.integers(1, 50))]))
            batch = int(rng.choice([1, int(rng.integers(2, 9)), int(rng.integers(9, 41))]))
            if in_f * out_f > 40_000_000: out_f = max(1, 40_000_000 // in_f)
            w = synth.fill_normal(seed * 7 + 1, in_f * out_f, 0, 0.05).reshape(out_f, in_f)
            bias = synth.fill_normal(seed * 7 + 2, out_f, 0, 0.1) if rng.integers(0, 4) else None
            x = synth.fill_normal(seed * 7 + 3, batch * in_f, 0, 1.0)
            y = np.zeros(batch * out_f, np.float32)
            ops.Linear(in_f, out_f, w, bias).forward(x, y)
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
            ops.LayerNorm(n, g, b).forward(x)
            what = f"LayerNorm n {n} rows {rows}"
            assert np.abs(x.reshape(rows, n) - ref).max() < 2e-5, what
        elif kind == 4:
            n = int(rng.integers(1, 100000))
            x = synth.fill_normal(seed + 8, n, 0, 3.0)
            r = x.astype(np.float64)
            ref = 0.5 * r * (1 + np.tanh(np.sqrt(2 / np.pi) * (r + 0.044715 * r ** 3)))
            ops.gelu(x)
            what = f"gelu n {n}"
            assert np.abs(x - ref).max() < 2e-6, what
        else:
            n = int(rng.integers(1, 60000))
            x = synth.fill_normal(seed + 9, n, 0, 4.0)
            r = x.astype(np.float64); r = np.exp(r - r.max()); ref = r / r.sum()
            ops.softmax(x)
            what = f"softmax n {n}"
            assert np.abs(x - ref).max() < 1e-6 * max(1.0, ref.max() * 10), what
    except Exception:
        bad.append(seed)
        traceback.print_exc(limit=1)
print(f"{count} cases from seed {first}: {len(bad)} failed {bad}")
