#!/usr/bin/env python3
"""More seeds of tests/test_prefill_gpu.py::test_prefill_shape_sweep_against_the_oracle than the suite carries:
python tests/sweeps/prefill.py [first_seed] [count] — prints the failing seeds (none expected)."""
import os, sys, traceback
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from zig_gpt2_amd import _lib
import test_prefill_gpu as T

zg = _lib.load(); _lib.check(zg.zg_init(0))
first, count = (int(v) for v in (sys.argv[1:3] + ["12", "100"][len(sys.argv) - 1:]))
bad = []
for seed in range(first, first + count):
    try:
        T.test_prefill_shape_sweep_against_the_oracle(zg, seed)
    except Exception:
        bad.append(seed)
        traceback.print_exc(limit=2)
print(f"{count} seeds from {first}: {len(bad)} failed {bad}")
sys.exit(1 if bad else 0)
