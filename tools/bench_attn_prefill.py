#!/usr/bin/env python3
"""Time the causal prompt attention kernel alone (zg_debug_attn_prefill): tools/bench_attn_prefill.py [batch] [tokens] [heads] [key_tiles]
K / V from head-major fp32 caches, as zg_gpt_prefill runs it.  Prints us per call (merge kernel included when key ranges are split)
and the matrix-core rate of the six plane products over the causal half."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from zig_gpt2_amd import _lib, synth

B, P, H, tiles = (int(v) for v in (sys.argv[1:5] + ["8", "1023", "12", "0"][len(sys.argv) - 1:]))
E, ctx = 64 * H, 1024 if P <= 1024 else P
lib = _lib.load(); _lib.check(lib.zg_init(0))
qkv = synth.fill_normal(5, B * P * 3 * E, 0, 1.0).reshape(B * P, 3 * E)
kc = np.zeros((B, H, ctx, 64), np.float32); vc = np.zeros((B, H, ctx, 64), np.float32)
kc[:, :, :P] = qkv[:, E:2 * E].reshape(B, P, H, 64).transpose(0, 2, 1, 3)
vc[:, :, :P] = qkv[:, 2 * E:].reshape(B, P, H, 64).transpose(0, 2, 1, 3)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
qkv_d, kc_d, vc_d = dev(qkv), dev(kc), dev(vc)
out = torch.zeros((B * P, 3 * E), dtype=torch.int16, device="cuda")
ws = torch.zeros(16 << 20, dtype=torch.float32, device="cuda")
stream = torch.cuda.Stream(); _lib.check(lib.zg_set_stream(stream.cuda_stream))
torch.cuda.synchronize()
run = lambda: _lib.check(lib.zg_debug_attn_prefill(qkv_d.data_ptr(), out.data_ptr(), B, P, E, H, kc_d.data_ptr(), vc_d.data_ptr(), ctx, ws.data_ptr(), ws.numel(), tiles))
reps = int(os.environ.get("REPS", "50"))
for _ in range(reps): run()
torch.cuda.synchronize()
best, tot = 1e9, 0.0
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps): run()
    e1.record(stream); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps; best = min(best, us); tot += us
us = tot / 5
flops = 6 * 2 * 2 * 64 * B * H * P * (P + 1) / 2  # six plane products, two matrix products, causal half
print(json.dumps({"batch": B, "tokens": P, "heads": H, "key_tiles": tiles, "us": round(us, 2), "us_min": round(best, 2),
                  "mfma_tflops_causal": round(flops / us / 1e6, 1), "frac_of_2.5PF": round(flops / us / 1e6 / 2500, 4)}))
