#!/usr/bin/env python3
"""Development check of the persistent MFMA GEMM on the GPU box: correctness against torch fp32 matmul of the
same bf16 operands over tile-multiple and ragged shapes (1 tile and many tiles per workgroup), a repeat-run race
screen (bitwise identical outputs), and timing of the BASELINE 768x3072 point.
    python tools/gemm_check.py [quick]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from zig_gpt2_amd import _lib, synth


def gelu_ref(x):
    u = x * 0.7978845608 * (1.0 + 0.044715 * x * x)
    return 0.5 * x * (1.0 + torch.tanh(u))


def run(lib, a, b, bias, gelu, out_bf16):
    m, k = a.shape
    n = b.shape[0]
    c = torch.full((m, n), float("nan"), dtype=torch.bfloat16 if out_bf16 else torch.float32, device="cuda")
    torch.cuda.synchronize()  # the library launches on its own (non-blocking) stream
    _lib.check(lib.zg_gemm_bf16_nt(a.data_ptr(), b.data_ptr(), 0 if bias is None else bias.data_ptr(), c.data_ptr(), m, n, k,
                                  int(gelu), int(out_bf16)))
    _lib.check(lib.zg_synchronize())
    return c


def check(lib, m, n, k, gelu, out_bf16, bn=0, wgs=0, reps=3):
    os.environ["ZGPT2_GEMM_BN"] = str(bn)
    os.environ["ZGPT2_GEMM_WGS"] = str(wgs)
    g = torch.Generator(device="cuda").manual_seed(m * 7 + n * 3 + k)
    a = (torch.rand((m, k), device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.randn((n, k), device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn((n,), device="cuda", generator=g) * 0.5
    ref = a.float() @ b.float().T + bias
    if gelu:
        ref = gelu_ref(ref)
    outs = [run(lib, a.view(torch.int16), b.view(torch.int16), bias, gelu, out_bf16) for _ in range(reps)]
    same = all(torch.equal(outs[0].view(torch.int16 if out_bf16 else torch.int32), o.view(torch.int16 if out_bf16 else torch.int32)) for o in outs[1:])
    got = outs[0].float()
    err = (got - ref).abs()
    tol = (0.01 if out_bf16 else 2e-4) * ref.abs() + (2e-3 if out_bf16 else 2e-4)
    bad = int((~(err <= tol)).sum().item())
    print(json.dumps({"m": m, "n": n, "k": k, "gelu": gelu, "bf16": out_bf16, "bn": bn, "wgs": wgs, "max_err": float(err.max()),
                      "bad": bad, "nan": int(torch.isnan(got).sum()), "repeatable": same}), flush=True)
    return bad == 0 and same


def time_it(lib, m, n=3072, k=768, gelu=True, out_bf16=True, bn=0, iters=50):
    os.environ["ZGPT2_GEMM_BN"] = str(bn)
    os.environ["ZGPT2_GEMM_WGS"] = "0"
    g = torch.Generator(device="cuda").manual_seed(1)
    a = (torch.rand((m, k), device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.randn((n, k), device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    bias = torch.randn((n,), device="cuda", generator=g) * 0.02
    c = torch.empty((m, n), dtype=torch.bfloat16 if out_bf16 else torch.float32, device="cuda")
    stream = torch.cuda.Stream()
    _lib.check(lib.zg_set_stream(stream.cuda_stream))
    f = lambda: _lib.check(lib.zg_gemm_bf16_nt(a.data_ptr(), b.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, int(gelu), int(out_bf16)))
    for _ in range(400):  # the chip's clocks take tens of milliseconds of load to settle
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(iters):
            f()
        e1.record(stream)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    tf = 2.0 * m * n * k / best / 1e6
    print(json.dumps({"time": [m, n, k], "gelu": gelu, "bf16": out_bf16, "bn": bn, "us": round(best, 2), "tflops": round(tf, 1),
                      "frac_2.5PF": round(tf / 2500, 4)}), flush=True)


if __name__ == "__main__":
    lib = _lib.load()
    _lib.check(lib.zg_init(0))
    ok = True
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    shapes = [(256, 256, 128), (256, 192, 128), (128, 128, 128), (512, 768, 768), (100, 136, 192), (1000, 200, 128), (1023, 2304, 768),
              (384, 128, 3072), (2048, 3072, 768)]
    for (m, n, k) in shapes:
        for bn in (192, 256):
            ok &= check(lib, m, n, k, gelu=False, out_bf16=False, bn=bn)
    ok &= check(lib, 1024, 768, 768, True, True, bn=192)
    ok &= check(lib, 1024, 768, 768, True, True, bn=256)
    # many tiles per workgroup (persistence, tile hand-over, odd K-step counts)
    for bn in (192, 256):
        ok &= check(lib, 2048, 1536, 768, True, True, bn=bn, wgs=8)
        ok &= check(lib, 1100, 776, 320, False, False, bn=bn, wgs=3)
        ok &= check(lib, 1024, 1024, 1600, True, False, bn=bn, wgs=5)
    ok &= check(lib, 8192, 3072, 768, True, True, reps=5)
    print("ALL OK" if ok else "FAILURES", flush=True)
    if not quick:
        for bn in (192, 256):
            time_it(lib, 8192, bn=bn)
        time_it(lib, 16384)
        time_it(lib, 1024)
        time_it(lib, 8192, gelu=False)
        time_it(lib, 8192, gelu=False, out_bf16=False)
        time_it(lib, 8192, 4096, 4096, gelu=False)
