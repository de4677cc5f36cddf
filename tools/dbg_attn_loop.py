import sys, numpy as np, torch
sys.path.insert(0, "tests")
from zig_gpt2_amd import _lib, synth
import test_attn_prefill_gpu as T
zg = _lib.load(); _lib.check(zg.zg_init(0))
class MP:
    def setenv(self, k, v):
        import os; os.environ[k] = v
B, P, H, tiles, cache, spike = [int(x) for x in sys.argv[1:7]]
bad = 0
for it in range(int(sys.argv[7])):
    try:
        T.test_attn_prefill_matches_float64(zg, MP(), B, P, H, tiles, bool(cache), bool(spike))
    except AssertionError as e:
        bad += 1; print("fail", it, str(e)[:80])
print("case", sys.argv[1:7], "failures", bad, "of", sys.argv[7])
