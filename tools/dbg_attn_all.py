import sys, os, numpy as np, torch
sys.path.insert(0, "tests")
from zig_gpt2_amd import _lib, synth
import test_attn_prefill_gpu as T
zg = _lib.load(); _lib.check(zg.zg_init(0))
cases = [(1, 1, 2, 0, False, False), (2, 31, 2, 0, False, False), (1, 32, 3, 0, True, False), (2, 33, 2, 1, True, False),
    (1, 64, 12, 0, True, False), (1, 80, 12, 0, True, False), (1, 95, 12, 0, True, False), (1, 96, 12, 0, True, False),
    (1, 97, 12, 0, False, False), (3, 129, 2, 2, True, True), (1, 300, 4, 3, True, True), (2, 257, 3, 0, False, True),
    (1, 1023, 2, 0, True, False), (1, 1023, 2, 5, False, True)]
def run(B, P, H, tiles, cache, spike):
    E, ctx = 64 * H, ((P + 63) // 64) * 64 + 64
    qkv = synth.fill_normal(21 + P, B * P * 3 * E, 0, 1.0).reshape(B * P, 3 * E)
    if spike:
        qkv[P // 2, E:E + 64] *= 9.0
        qkv[P // 2 + 1:, :64] += 3.0 * np.sign(qkv[P // 2, E:E + 64])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    qkv_d = dev(qkv)
    out_d = torch.zeros((B * P, 3 * E), dtype=torch.int16, device="cuda")
    ws = torch.zeros(B * H * P * 40 * 66 + 16, dtype=torch.float32, device="cuda") if P <= 320 or tiles else torch.zeros(16 << 20, dtype=torch.float32, device="cuda")
    kc = vc = None
    if cache:
        x = qkv.reshape(B, P, 3, H, 64)
        full = np.full((2, B, H, ctx, 64), np.nan, np.float32)
        full[0, :, :, :P] = x[:, :, 1].transpose(0, 2, 1, 3)
        full[1, :, :, :P] = x[:, :, 2].transpose(0, 2, 1, 3)
        kc, vc = dev(full[0]), dev(full[1])
        qkv_d[:, E:] = float("nan")
    torch.cuda.synchronize()
    _lib.check(zg.zg_debug_attn_prefill(qkv_d.data_ptr(), out_d.data_ptr(), B, P, E, H, kc.data_ptr() if cache else None,
                                        vc.data_ptr() if cache else None, ctx, ws.data_ptr(), ws.numel(), tiles))
    torch.cuda.synchronize()
    got = T.planes_to_f64(out_d.cpu().numpy().view(np.uint16), E)
    ref = T.ref_attention(qkv, B, P, E, H)
    err = np.abs(got - ref) / np.abs(ref).max()
    bad = np.argwhere(~(err < 5e-6))
    if len(bad):
        rows = sorted(set(int(r) for r in bad[:, 0])); heads = sorted(set(int(c) // 64 for c in bad[:, 1]))
        r0 = rows[0]; c0 = int(bad[0, 1]) // 64 * 64
        np.set_printoptions(precision=4, linewidth=250)
        print(" got", got[r0, c0:c0 + 40]); print(" ref", ref[r0, c0:c0 + 40]); rr = rows[len(rows) // 2]; print(" row", rr, "got", got[rr, c0:c0 + 12], "ref", ref[rr, c0:c0 + 12])
        print("FAIL", (B, P, H, tiles, cache, spike), "bad elems", len(bad), "rows", rows[:12], "..", rows[-3:], "heads", heads, "dcols", sorted(set(int(c) % 64 for c in bad[:, 1]))[:16])
    else:
        print("ok  ", (B, P, H, tiles, cache, spike), float(err.max()))
for rep in range(1):
    for c in cases: run(*c)
