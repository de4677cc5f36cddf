#!/usr/bin/env python3
"""Timeline of the prompt attention's workgroups from the diagnostic build (make -C zig_gpt2_amd/csrc stamps;
ZGPT2_LIB=zig_gpt2_amd/lib/libzgpt2_hip_stamps.so python3 tools/attn_timeline.py [batch] [tokens] [heads]): start / end of every
workgroup by s_memrealtime, the CU it ran on — residency per CU over time, per-tile pace by co-residency, the tail."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from zig_gpt2_amd import _lib, synth

B, P, H = (int(v) for v in (sys.argv[1:4] + ["8", "1023", "12"][len(sys.argv) - 1:]))
E, ctx = 64 * H, 1024
lib = _lib.load(); _lib.check(lib.zg_init(0))
qkv = synth.fill_normal(5, B * P * 3 * E, 0, 1.0).reshape(B * P, 3 * E)
kc = np.zeros((B, H, ctx, 64), np.float32); vc = np.zeros((B, H, ctx, 64), np.float32)
kc[:, :, :P] = qkv[:, E:2 * E].reshape(B, P, H, 64).transpose(0, 2, 1, 3)
vc[:, :, :P] = qkv[:, 2 * E:].reshape(B, P, H, 64).transpose(0, 2, 1, 3)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
qkv_d, kc_d, vc_d = dev(qkv), dev(kc), dev(vc)
out = torch.zeros((B * P, 3 * E), dtype=torch.int16, device="cuda")
ws = torch.zeros(16 << 20, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
run = lambda: _lib.check(lib.zg_debug_attn_prefill(qkv_d.data_ptr(), out.data_ptr(), B, P, E, H, kc_d.data_ptr(), vc_d.data_ptr(), ctx, ws.data_ptr(), ws.numel(), 0))
for _ in range(5): run()
torch.cuda.synchronize()
ng = ((P + 31) // 32 + 3) // 4
n_wg = H * B * ng
st = ws.cpu().numpy().view(np.uint64)[:4 * n_wg].reshape(n_wg, 4)
t0 = st[:, 0].min()
start = (st[:, 0] - t0).astype(np.float64) / 100.0  # us
end = (st[:, 1] - t0).astype(np.float64) / 100.0
hw = st[:, 2] & 0xffffffff
xcc = (st[:, 2] >> 32) & 0xf
cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4)  # CU_ID | SE_ID << 4 (gfx9 HW_ID layout)
tiles = st[:, 3].astype(np.int64)
key = xcc.astype(np.int64) * 1000 + cu.astype(np.int64)
print(f"{n_wg} workgroups on {len(set(key.tolist()))} distinct (XCC, SE, CU); kernel span {end.max():.1f} us; first start spread {start[:512].max():.1f} us")
for n in sorted(set(tiles.tolist()), reverse=True):
    m = tiles == n
    print(f"  {n:2d} tiles: {m.sum():3d} WGs, start {start[m].min():6.1f} .. {start[m].max():6.1f} us, duration {np.mean(end[m] - start[m]):6.1f} us avg ({np.min(end[m] - start[m]):.1f} .. {np.max(end[m] - start[m]):.1f}),"
          f" {np.mean((end[m] - start[m]) / n) * 1000:.0f} ns per tile")
# residency over time
grid = np.linspace(0, end.max(), 41)
print("  time us : resident workgroups")
for t in grid[:-1]:
    print(f"  {t:7.1f} : {int(((start <= t) & (end > t)).sum())}")
per_cu = {}
for k, s_, e_ in zip(key.tolist(), start.tolist(), end.tolist()): per_cu.setdefault(k, []).append((s_, e_))
busy = [sum(e - s for s, e in v) for v in per_cu.values()]
print(f"  per CU: workgroups {min(len(v) for v in per_cu.values())} .. {max(len(v) for v in per_cu.values())}; resident WG-time {min(busy):.1f} .. {max(busy):.1f} us (2 x span = {2 * end.max():.1f})")
