#!/usr/bin/env python3
"""Random cases of tests/test_attn_prefill_gpu.py::test_attn_prefill_matches_float64 (batch, prompt length, heads, forced key
tiles per workgroup, cache / qkv rows as the K / V source, a dominant key): python tests/sweeps/attn.py [first_seed] [count]."""
import os, sys, traceback
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from zig_gpt2_amd import _lib
import test_attn_prefill_gpu as T

zg = _lib.load(); _lib.check(zg.zg_init(0))
first, count = (int(v) for v in (sys.argv[1:3] + ["0", "100"][len(sys.argv) - 1:]))
bad = []
for seed in range(first, first + count):
    rng = np.random.default_rng(7000 + seed)
    B, H = int(rng.integers(1, 5)), int(rng.integers(1, 13))
    P = int(rng.integers(1, 700)) if rng.integers(0, 4) else int(rng.integers(1, 70))
    tiles = int(rng.integers(0, 9)) if rng.integers(0, 2) else 0
    case = (B, P, H, tiles, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)) and P > 4)
    try:
        T.test_attn_prefill_matches_float64(zg, None, *case)
    except Exception:
        bad.append((seed, case))
        traceback.print_exc(limit=1)
print(f"{count} cases from seed {first}: {len(bad)} failed {bad}")
sys.exit(1 if bad else 0)
