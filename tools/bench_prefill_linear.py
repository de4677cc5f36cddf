#!/usr/bin/env python3
"""Time ONE whole-prompt Linear (zg_debug_prefill_linear) on either GEMM family: tools/bench_prefill_linear.py M N K epilogue kernel [slices]
epilogue 1 = residual add (partial slabs + reduce), 2 = GELU + three-plane split; kernel 1 = gemm_s4 (persistent, planes in one K loop),
2 = the 128-row prompt GEMM.  Prints us per call and the matrix-core rate of the three plane products."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from zig_gpt2_amd import _lib, synth

M, N, K, epi, kern = (int(v) for v in sys.argv[1:6])
slices = int(sys.argv[6]) if len(sys.argv) > 6 else 0
lib = _lib.load(); _lib.check(lib.zg_init(0))
a = torch.from_numpy(synth.to_bf16_bits(synth.fill_normal(1, M * 3 * K, 0, 1.0)).view(np.int16)).cuda()
w = torch.from_numpy(synth.to_bf16_bits(synth.fill_normal(2, N * K, 0, 0.02)).view(np.int16)).cuda()
bias = torch.from_numpy(synth.fill_normal(3, N, 0, 0.02)).cuda()
c = torch.zeros(M * 3 * N, dtype=torch.int16, device="cuda") if epi == 2 else torch.zeros(M * N, dtype=torch.float32, device="cuda")
ws = torch.zeros(16 << 20, dtype=torch.float32, device="cuda")
stream = torch.cuda.Stream(); _lib.check(lib.zg_set_stream(stream.cuda_stream))
run = lambda: _lib.check(lib.zg_debug_prefill_linear(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, epi, kern, slices, ws.data_ptr(), ws.numel()))
for _ in range(200): run()
torch.cuda.synchronize()
best, tot = 1e9, 0.0
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(30): run()
    e1.record(stream); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30; best = min(best, us); tot += us
us = tot / 5
print(json.dumps({"M": M, "N": N, "K": K, "epilogue": epi, "kernel": "gemm_s4" if kern == 1 else "prefill_gemm", "slices": slices, "us": round(us, 2), "us_min": round(best, 2),
                  "mfma_tflops_3planes": round(6.0 * M * N * K / us / 1e6, 1), "frac_of_2.5PF": round(6.0 * M * N * K / us / 1e6 / 2500, 4)}))
