#!/usr/bin/env python3
"""Edge and nonsense arguments through the op tier of the C ABI: zero lengths, zero dimensions, lengths that do not divide,
null pointers, a sequence length of 0, head counts that do not divide — every call must RETURN (ZG_OK or an error code with a
message), never crash or hang, and the library must keep working afterwards.  python tests/sweeps/errors.py [seed] [count]"""
import ctypes as C, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np
from zig_gpt2_amd import _lib, ops, synth

zg = _lib.load(); _lib.check(zg.zg_init(0))
seed, count = (int(v) for v in (sys.argv[1:3] + ["0", "400"][len(sys.argv) - 1:]))
rng = np.random.default_rng(seed)
buf = [np.zeros(1 << 21, np.float32) for _ in range(8)]  # (large enough for every weight shape below: lengths the ABI cannot check must not lie)
idx = np.zeros(64, np.uint64)
import torch
dbuf = [torch.zeros(1 << 21, dtype=torch.float32, device="cuda") for _ in range(5)]
P = lambda a: a.ctypes.data
def size():
    return int(rng.choice([0, 0, 1, 2, 3, 7, 8, 63, 64, 65, 100, 128, 192, 255, 256, 1000, 4096, int(rng.integers(0, 1 << 16))]))
def ptr(i):
    return None if rng.integers(0, 12) == 0 else P(buf[i])
codes = {}
hip_errors = []  # a nonsense argument must be refused by a check, not by the HIP runtime
only = int(os.environ.get("ONLY", "-1"))
for it in range(count):
    k = int(rng.integers(0, 12))
    if only >= 0: k = only
    if os.environ.get("TRACE"): print("call", it, "kind", k, flush=True)
    if k == 0:
        i_f, o_f = size(), size()
        if i_f * o_f > (1 << 21): o_f = (1 << 21) // max(i_f, 1)
        r = zg.zg_linear_forward(i_f, o_f, ptr(0), ptr(1), ptr(2), size(), ptr(3), size())
    elif k == 1: r = zg.zg_embedding_forward(size(), ptr(0), size(), P(idx) if rng.integers(0, 10) else None, int(rng.integers(0, 65)), ptr(1), size())
    elif k == 2: r = zg.zg_layernorm_forward(size(), ptr(0), ptr(1), 1e-5, ptr(2), size())
    elif k == 3:
        h, e = int(rng.choice([0, 1, 2, 3, 12, 16])), int(rng.choice([0, 64, 128, 100, 192, 768]))
        t = int(rng.choice([0, 1, 2, 5, 40]))
        r = zg.zg_attn_forward(h, e, ptr(0), ptr(1), ptr(2), ptr(3), t, ptr(4), size(), ptr(5), size(), ptr(6), size(), ptr(7), size(),
                               ptr(4), size(), ptr(5), size(), ptr(6), size(), ptr(7), size(), ptr(4), size())
    elif k == 4: r = zg.zg_split_qkv(size(), int(rng.choice([0, 1, 5])), ptr(0), size(), int(rng.integers(0, 5)), ptr(1), size())
    elif k == 5: r = zg.zg_transpose(int(rng.choice([0, 1, 5])), int(rng.choice([0, 1, 12])), int(rng.choice([0, 1, 64])), ptr(0), size(), ptr(1), size())
    elif k == 6: r = zg.zg_scaled_dot_product_attention(ptr(0), size(), ptr(1), size(), ptr(2), size(), int(rng.choice([0, 1, 12])), int(rng.choice([0, 1, 7])),
                                                         int(rng.choice([0, 32, 64])), ptr(3), size(), ptr(4), size())
    elif k == 9:   # matrix-core GEMM entry: device pointers required, K a multiple of 64 >= 128, N % 8 for bf16 out
        dp = lambda i: None if rng.integers(0, 10) == 0 else (dbuf[i].data_ptr() if rng.integers(0, 8) else P(buf[i]))
        M, N, K = int(rng.choice([0, 1, 17, 256, 300])), int(rng.choice([0, 1, 8, 12, 64, 200])), int(rng.choice([0, 64, 100, 128, 192, 1000]))
        r = zg.zg_gemm_bf16_nt(dp(0), dp(1), dp(2) if rng.integers(0, 2) else None, dp(3), M, N, K, int(rng.integers(0, 2)), int(rng.integers(0, 2)))
    elif k == 10:  # one whole-prompt Linear (debug entry)
        dp = lambda i: None if rng.integers(0, 10) == 0 else dbuf[i].data_ptr()
        M, N, K = int(rng.choice([0, 1, 100, 300])), int(rng.choice([0, 32, 64, 192, 200])), int(rng.choice([0, 64, 100, 128, 256]))
        r = zg.zg_debug_prefill_linear(dp(0), dp(1), dp(2), dp(3), M, N, K, int(rng.integers(-1, 4)), int(rng.integers(-1, 4)), int(rng.integers(0, 6)),
                                       dp(4) if rng.integers(0, 3) else None, int(rng.choice([0, 1000, 1 << 20])))
    elif k == 11:  # the prompt attention alone (debug entry)
        dp = lambda i: None if rng.integers(0, 10) == 0 else dbuf[i].data_ptr()
        B, Pn, H = int(rng.choice([0, 1, 2])), int(rng.choice([0, 1, 33, 200])), int(rng.choice([0, 1, 2, 3]))
        E = int(rng.choice([64 * H, 64 * H + 64, 100]))
        cache = bool(rng.integers(0, 2))
        r = zg.zg_debug_attn_prefill(dp(0), dp(1), B, Pn, E, H, dp(2) if cache else None, dp(3) if cache else None, int(rng.choice([0, Pn, 256])),
                                     dp(4) if rng.integers(0, 3) else None, int(rng.choice([0, 100, 1 << 20])), int(rng.choice([0, 1, 3, 300, -1])))
    elif k == 7: r = zg.zg_gelu(ptr(0), size())
    else: r = zg.zg_softmax(ptr(0), size())
    codes[r] = codes.get(r, 0) + 1
    if r != 0: assert len(zg.zg_last_error()) > 0
    if r == -3:
        hip_errors.append((it, k))
        print(f"call {it} kind {k}: HIP error: {zg.zg_last_error().decode(errors='replace')}", flush=True)
# the library still computes
w = synth.fill_normal(1, 64 * 32, 0, 0.1).reshape(32, 64); x = synth.fill_normal(2, 64, 0, 1.0); y = np.zeros(32, np.float32)
ops.Linear(64, 32, w, None).forward(x, y)
assert np.abs(y - w @ x).max() < 1e-5
print(f"{count} calls returned; status histogram {dict(sorted(codes.items()))}; the library still computes")
sys.exit(1 if hip_errors else 0)
