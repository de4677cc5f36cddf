#!/usr/bin/env python3
"""MFMA utilisation of the 768x3072 (c_fc) GEMM: C[M,3072] = A[M,768] * W[3072,768]^T + bias, GELU, bf16 out.
Reports TFLOP/s against the 2.5 PFLOP/s dense bf16 peak for M in {1024, 8192} (BASELINE headline M = 8192)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from zig_gpt2_amd import _lib, synth

PEAK_TF = 2500.0


def measure(lib, m, n=3072, k=768, gelu=True, out_bf16=True, iters=50, batches=5, warm=400):
    a = torch.from_numpy(synth.to_bf16_bits(synth.fill_uniform(1, m * k, -1, 1)).view(np.int16)).cuda()
    b = torch.from_numpy(synth.to_bf16_bits(synth.fill_normal(2, n * k, 0, 0.02)).view(np.int16)).cuda()
    bias = torch.from_numpy(synth.fill_normal(3, n, 0, 0.02)).cuda()
    c = torch.empty((m, n), dtype=torch.bfloat16 if out_bf16 else torch.float32, device="cuda")
    stream = torch.cuda.Stream()
    _lib.check(lib.zg_set_stream(stream.cuda_stream))
    run = lambda: _lib.check(lib.zg_gemm_bf16_nt(a.data_ptr(), b.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, int(gelu), int(out_bf16)))
    for _ in range(warm):  # the chip needs tens of milliseconds under load before its clocks settle
        run()
    torch.cuda.synchronize()
    n_batches, batches = batches, []
    for _ in range(n_batches):  # five timed batches: the MEAN is the headline, the minimum is printed beside it
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(iters):
            run()
        e1.record(stream)
        torch.cuda.synchronize()
        batches.append(e0.elapsed_time(e1) * 1e3 / iters)
    us, us_min = sum(batches) / len(batches), min(batches)
    flops = 2.0 * m * n * k
    return {"M": m, "N": n, "K": k, "us": round(us, 2), "us_min": round(us_min, 2), "tflops": round(flops / us / 1e6, 1),
            "mfma_frac_of_2.5PF": round(flops / us / 1e6 / PEAK_TF, 4), "mfma_frac_of_2.5PF_best_batch": round(flops / us_min / 1e6 / PEAK_TF, 4),
            "timing": "mean of %d batches of %d back-to-back launches (HIP events on the launch stream) after %d warm-up launches; us_min = best batch" % (n_batches, iters, warm),
            "kernel": os.environ.get("ZGPT2_GEMM_KERNEL", "default (gemm_s4_kernel for 192-wide tiles, gemm_p8_kernel for 256-wide)"), "stores": "write-through (sc1)",
            "gelu": gelu, "out": "bf16" if out_bf16 else "f32",
            "bytes_min": (m * k + n * k) * 2 + m * n * (2 if out_bf16 else 4)}


if __name__ == "__main__":
    lib = _lib.load(); _lib.check(lib.zg_init(0))
    if len(sys.argv) > 1:  # e.g. `bench_gemm.py 8192` or `bench_gemm.py 8192 4096 4096` (M N K), `nogelu` / `f32` anywhere
        gelu, bf16 = "nogelu" not in sys.argv, "f32" not in sys.argv
        a = [int(v) for v in sys.argv[1:] if v.isdigit()]
        print(json.dumps(measure(lib, a[0], *(a[1:3] if len(a) >= 3 else ()), gelu=gelu, out_bf16=bf16)))
        sys.exit(0)
    for m in (1024, 8192, 16384):
        print(json.dumps(measure(lib, m)))
    print(json.dumps(measure(lib, 8192, gelu=False, out_bf16=False)))
    print(json.dumps(measure(lib, 8192, n=4096, k=4096, gelu=False)))
