#!/usr/bin/env python3
"""N independent prompts on one GPU as G co-running groups of N / G sequences (gpt.GPTGroups: one handle per group on its own
stream, one shared weight region) against the lock-step batch (G = 1): tokens/s of full greedy generations, ids compared row
for row with G = 1.
usage: python tools/experiments/corun_ab.py [--model 124M] [--prompts 8] [--gens 2] [--ctx 0] [--prio cycle|normal|...] [--prefetch] G [G ...]
Every G runs in this process one after another; GPU_MAX_HW_QUEUES etc. are the caller's environment."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from zig_gpt2_amd import _lib, gpt as zgpt, synth

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="124M")
ap.add_argument("--prompts", type=int, default=8)
ap.add_argument("--gens", type=int, default=2)
ap.add_argument("--ctx", type=int, default=0)
ap.add_argument("--prio", default="cycle", help="cycle (0, +1, -1, ...), normal (all 0), or a comma list of priorities")
ap.add_argument("--prefetch", action="store_true", help="leave the side-stream L2 prefetcher to the library's default rule")
ap.add_argument("--kv-b24", action="store_true")
ap.add_argument("--like-bench", default="", help="comma list of what bench.py has done to the process before its groups table: "
                "stream (torch stream as the library stream), rccl (one-rank communicator made and destroyed), handle (a one-prompt handle with "
                "its prefetcher alive, one generation run), gemm (the GEMM bench)")
ap.add_argument("groups", type=int, nargs="+")
a = ap.parse_args()

cfg = synth.CONFIGS[a.model]
ctx = a.ctx or cfg.context_size
lib = _lib.load(); _lib.check(lib.zg_init(0))
gen = torch.Generator(device="cuda"); gen.manual_seed(7)
w = {}
for name, shape, mean, _ in synth.tensor_specs(cfg):
    t = torch.randn(shape, generator=gen, device="cuda", dtype=torch.float32) * 0.02 + mean
    w[name] = t.to(torch.bfloat16).to(torch.float32).contiguous()
prompts = [synth.rand_tokens(2000 + b, 1, cfg.vocab_size) for b in range(a.prompts)]
like = set(x for x in a.like_bench.split(",") if x)
keep = []
if "stream" in like:
    st = torch.cuda.Stream(); _lib.check(lib.zg_set_stream(st.cuda_stream)); keep.append(st)
if "rccl" in like:
    import ctypes as C
    uid = (C.c_ubyte * 128)()
    _lib.check(lib.zg_dist_unique_id(uid, 128)); _lib.check(lib.zg_dist_init(uid, 128, 0, 1)); _lib.check(lib.zg_dist_finalize())
if "handle" in like:
    h = zgpt.GPT(cfg, batch=1); h.load_weights(w); h.generate_enqueue(prompts[:1], ctx); torch.cuda.synchronize(); keep.append(h)
if "gemm" in like:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench_gemm
    bench_gemm.measure(lib, 8192)
ref_ids = None
for G in a.groups:
    if a.prio == "cycle":
        pr = None
    elif a.prio == "normal":
        pr = [0] * G
    else:
        pl = [int(x) for x in a.prio.split(",")]
        pr = [pl[i % len(pl)] for i in range(G)]
    m = zgpt.GPTGroups(cfg, a.prompts, G, priorities=pr, prefetch=a.prefetch, kv_b24=a.kv_b24)
    m.load_weights(w)
    m.generate_enqueue(prompts, ctx)
    torch.cuda.synchronize()
    # how fast the host enqueues when no queue is full: 64 steps = 8 graph launches per handle
    short = min(64, ctx)
    m.generate_enqueue(prompts, short); torch.cuda.synchronize()
    ts = time.perf_counter()
    m.generate_enqueue(prompts, short)
    t_host_short = time.perf_counter() - ts
    torch.cuda.synchronize()
    t_all_short = time.perf_counter() - ts
    t0 = time.perf_counter()
    t_enq = 0.0
    for _ in range(a.gens):
        te = time.perf_counter()
        m.generate_enqueue(prompts, ctx)
        t_enq += time.perf_counter() - te
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ids = m.generate_fetch(ctx)
    if ref_ids is None:
        ref_ids = ids
    row = {"model": a.model, "prompts": a.prompts, "groups": G, "per_group": a.prompts // G, "priorities": a.prio, "prefetch": a.prefetch, "like_bench": a.like_bench,
           "hw_queues_env": os.environ.get("GPU_MAX_HW_QUEUES"), "tokens_per_s": round(a.prompts * (ctx - 1) * a.gens / wall, 1),
           "ms_per_generation": round(1e3 * wall / a.gens, 2), "us_per_step": round(1e6 * wall / a.gens / ctx, 2),
           "host_enqueue_ms_per_generation": round(1e3 * t_enq / a.gens, 2), "feeder_threads": os.environ.get("ZGPT2_MANY_THREADS", "1"),
           "short64_host_ms": round(1e3 * t_host_short, 2), "short64_total_ms": round(1e3 * t_all_short, 2),
           "ids_equal_first": bool(np.array_equal(ids, ref_ids)), "first_tokens": [int(t) for t in ids[0, :4]]}
    print(json.dumps(row), flush=True)
    m.close()
