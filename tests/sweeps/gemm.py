#!/usr/bin/env python3
"""Random shapes through zg_gemm_bf16_nt (C = A B^T + bias, optional GELU, bf16 or fp32 out) against float64: any M, ragged N,
K a multiple of 64 from 128 up (both persistent kernels: beyond 16320 the eight-wave one).  python tests/sweeps/gemm.py [first_seed] [count]"""
import os, sys, traceback
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np
import torch
from zig_gpt2_amd import _lib, synth

zg = _lib.load(); _lib.check(zg.zg_init(0))
first, count = (int(v) for v in (sys.argv[1:3] + ["0", "100"][len(sys.argv) - 1:]))
bad = []
bits = lambda a: synth.to_bf16_bits(a)
f64 = lambda b: (b.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
for seed in range(first, first + count):
    rng = np.random.default_rng(1000 + seed)
    M = int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(40, 600)), int(rng.integers(600, 5000))]))
    K = 64 * int(rng.choice([int(rng.integers(2, 30)), int(rng.integers(30, 120)), int(rng.integers(250, 270))]))
    out_bf16, gelu = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    N = int(rng.choice([int(rng.integers(1, 100)), int(rng.integers(100, 1500)), 64 * int(rng.integers(1, 40))]))
    if out_bf16: N = max(8, N // 8 * 8)
    if N % 4 and K > 16320: N = N // 4 * 4 + 4  # (documented limit of the entry: a ragged N is stored by the four-wave kernel only)
    while M * K + N * K > 60_000_000: M = max(1, M // 2)
    what = f"seed {seed}: M {M} N {N} K {K} gelu {gelu} bf16 out {out_bf16}"
    try:
        a = bits(synth.fill_normal(seed * 5 + 1, M * K, 0, 1.0)).reshape(M, K)
        b = bits(synth.fill_normal(seed * 5 + 2, N * K, 0, 0.05)).reshape(N, K)
        bias = synth.fill_normal(seed * 5 + 3, N, 0, 0.2) if rng.integers(0, 3) else None
        ad, bd = torch.from_numpy(a.view(np.int16)).cuda(), torch.from_numpy(b.view(np.int16)).cuda()
        biasd = torch.from_numpy(bias).cuda() if bias is not None else None
        c = torch.zeros((M, N), dtype=torch.int16 if out_bf16 else torch.float32, device="cuda")
        torch.cuda.synchronize()
        _lib.check(zg.zg_gemm_bf16_nt(ad.data_ptr(), bd.data_ptr(), biasd.data_ptr() if bias is not None else None, c.data_ptr(), M, N, K, int(gelu), int(out_bf16)))
        _lib.check(zg.zg_sync()) if hasattr(zg, "zg_sync") else torch.cuda.synchronize()
        torch.cuda.synchronize()
        ref = f64(a) @ f64(b).T + (0 if bias is None else bias.astype(np.float64))
        if gelu: ref = 0.5 * ref * (1 + np.tanh(np.sqrt(2 / np.pi) * (ref + 0.044715 * ref ** 3)))
        got = f64(c.cpu().numpy().view(np.uint16)) if out_bf16 else c.cpu().numpy().astype(np.float64)
        tol = (2.0 ** -8) * np.abs(ref) + 1e-6 * np.abs(ref).max() if out_bf16 else 3e-6 * np.abs(ref).max()
        assert np.all(np.abs(got - ref) <= tol + 1e-30), (what, float(np.abs(got - ref).max()), float(np.abs(ref).max()))
    except Exception:
        bad.append(seed)
        print(what)
        traceback.print_exc(limit=1)
print(f"{count} shapes from seed {first}: {len(bad)} failed {bad}")
sys.exit(1 if bad else 0)
