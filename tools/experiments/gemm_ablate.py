#!/usr/bin/env python3
"""A/B timings of the persistent GEMM (ZGPT2_GEMM_DBG bits: 1 no global stores, 4 no epilogue, 8 drain the stores,
16 non-temporal stores)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_check
from zig_gpt2_amd import _lib
lib = _lib.load(); _lib.check(lib.zg_init(0))
for dbg in [int(a) for a in sys.argv[1:]] or (0, 16, 1, 4):
    os.environ["ZGPT2_GEMM_DBG"] = str(dbg)
    print("dbg", dbg, flush=True)
    gemm_check.time_it(lib, 8192)
    gemm_check.time_it(lib, 8192, gelu=False)
    gemm_check.time_it(lib, 16384)
