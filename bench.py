#!/usr/bin/env python3
"""Headline benchmark: tokens/s of GPT-2-124M greedy decode to a 1024-token context on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one complete greedy generation of the reference decode loop (src/main.zig:322-342 with
argmax instead of the sampler): context_size forward passes per prompt, all enqueued on the GPU
without a host round trip per token.  N=1 runs BASELINE.json configs[1] (one prompt, 1024 ctx);
N>1 runs configs[2] (independent prompts sharded over the ranks, 8 per GPU, weights broadcast once
from rank 0 over RCCL, no collective in the timed region).  Weights/prompts are synthetic (seeded
PRNG, bf16-representable) — there are no checkpoints offline.  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--model", default="124M", choices=["124M", "xl", "nano-char", "tiny"])
    p.add_argument("--prompts-per-gpu", type=int, default=0, help="0 = 1 prompt at N=1, 8 per GPU at N>1")
    p.add_argument("--ctx", type=int, default=0, help="decode steps per generation (default: context_size)")
    p.add_argument("--kv-f16", action="store_true")
    p.add_argument("--weights-f32", action="store_true")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for the CPU baseline sample")
    p.add_argument("--seed", type=int, default=0)
    return p.parse_args()


class _DevMem:
    """Expose a raw device allocation to torch (for the RCCL broadcast of the weight arena)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def cpu_baseline(cfg, weights, prompt, budget_s):
    """The oracle (C restatement of the reference CPU path, fp32, all host cores) timed on a bounded
    sample of the same workload: a window of early positions and a window of late positions; the
    per-token cost is affine in T (KV re-transpose + attention, src/ops.zig:153-160), so the whole
    1..ctx run is priced from the two windows."""
    import oracle

    cores = oracle.host_cores()  # affinity mask clipped by the cgroup CPU quota: the threads actually usable
    oracle.set_num_threads(cores)
    blas = "built-in OpenMP sgemm"
    found = oracle.find_cblas()
    m = oracle.GPT(cfg, weights)
    ctx = cfg.context_size

    def window(t0, n):
        tok = int(prompt[0])
        m.forward(t0, tok, True)  # warm
        t = time.perf_counter()
        for i in range(n):
            m.forward(t0 + i, tok, True)
        return (time.perf_counter() - t) / n

    # choose the faster of the built-in SGEMM and a real CBLAS (if one can be dlopen'ed) on a short probe
    t_builtin = window(1, 4)
    if found and oracle.use_cblas(found[0], found[1]) == 0:
        t_blas = window(1, 4)
        if t_blas < t_builtin:
            blas = found[2]
        else:
            oracle.use_cblas(None)
    n_win = max(8, int(budget_s / 2 / max(min(t_builtin, 1.0), 1e-3) / 1.5))
    n_win = min(n_win, 96)
    lo0, hi0 = 1, max(1, ctx - n_win)
    t_lo = window(lo0, n_win)
    t_hi = window(hi0, n_win)
    # affine model: cost(T) = a + b*T, fitted at the window centres, averaged over T = 1..ctx
    c_lo, c_hi = lo0 + (n_win - 1) / 2, hi0 + (n_win - 1) / 2
    b = (t_hi - t_lo) / max(c_hi - c_lo, 1.0)
    a = t_lo - b * c_lo
    mean_cost = a + b * (ctx + 1) / 2
    oracle.use_cblas(None)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    return {
        "value": round(1.0 / mean_cost, 2), "unit": "tokens/s", "cores": cores, "kind": "port",
        "cpu_model": cpu_model, "visible_cpus": os.cpu_count(), "blas": blas,
        "sample": f"oracle GPT.forward fp32, {blas}: {n_win} tokens at T={lo0}.. ({1e3 * t_lo:.1f} ms/tok) and "
                  f"{n_win} at T={hi0}.. ({1e3 * t_hi:.1f} ms/tok); whole 1..{ctx} run priced by the affine fit",
    }


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    import torch

    from zig_gpt2_amd import _lib, gpt, shard, synth

    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dist = None
    # ZGPT2_FORCE_DIST=1 runs the RCCL path (process group, weight broadcast, barriers, max-reduce) with a
    # single rank: the only way to exercise it on a one-GPU box
    use_dist = world > 1 or os.environ.get("ZGPT2_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    lib = _lib.load()  # no fallback: raises if libzgpt2_hip.so is missing
    _lib.check(lib.zg_init(local_rank))
    stream = torch.cuda.Stream()
    _lib.check(lib.zg_set_stream(stream.cuda_stream))

    cfg = synth.CONFIGS[a.model]
    ctx = a.ctx or cfg.context_size
    ppg = a.prompts_per_gpu or (1 if world == 1 else 8)
    model = gpt.GPT(cfg, batch=ppg, weights_f32=a.weights_f32, use_graph=not a.no_graph, kv_f16=a.kv_f16)

    # ---- weights: generated and uploaded on rank 0, broadcast to the other GPUs over RCCL/xGMI
    weights = None
    t0 = time.perf_counter()
    if rank == 0:
        weights = synth.make_weights(cfg, seed=a.seed, bf16=not a.weights_f32)
        model.load_weights(weights)
    bcast_ms = None
    if use_dist:
        ptr, nbytes = model.weight_arena()
        arena = torch.as_tensor(_DevMem(ptr, nbytes), device=torch.device("cuda", local_rank))
        torch.cuda.synchronize()
        dist.barrier()
        tb = time.perf_counter()
        shard.broadcast_weights(arena, dist, src=0)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - tb) * 1e3
    setup_s = time.perf_counter() - t0

    # ---- prompts: one token each, distinct per global prompt index
    mine = shard.shard_prompts(ppg * world, world, rank)
    prompts = [synth.rand_tokens(1000 + a.seed * 131 + gi, 1, cfg.vocab_size) for gi in mine]

    def one_generation():
        model.generate_enqueue(prompts, ctx)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        one_generation()
    sync_all()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_wall = time.perf_counter()
    e0.record(stream)
    for _ in range(a.steps):
        one_generation()
    e1.record(stream)
    sync_all()
    wall_s = time.perf_counter() - t_wall
    dev_s = e0.elapsed_time(e1) / 1e3
    elapsed = wall_s
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ids = model.generate_fetch(ctx)

    # tokens produced per generation: every position after the prompt is one generated token
    gen_tokens = ppg * (ctx - 1)
    value = world * gen_tokens * a.steps / elapsed

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel: ln_f + lm_head GEMV + argmax (31 % of the bytes of a step)
    wbytes, _ = model.step_bytes(1)
    n_prof = min(64, ctx)
    prof_lo = model.profile_step(1, n_prof)
    prof_hi = model.profile_step(ctx - n_prof + 1, n_prof)
    lm_interval_us = 0.5 * (prof_lo["lnf_lm_head_argmax"] + prof_hi["lnf_lm_head_argmax"])
    null_us = 0.5 * (prof_lo["null_kernel_interval"] + prof_hi["null_kernel_interval"])
    # Event pairs around single kernels carry +-3 us of event/boundary overhead at this granularity (the
    # null-kernel interval is itself ~6 us), so the roofline duration is the per-launch time of 256
    # back-to-back launches between ONE pair of HIP events (includes the ~1.6 us launch boundary, i.e. it
    # under-states the rate); the in-situ intervals are reported beside it.
    lm_bytes = cfg.vocab_size * cfg.n_embed * (4 if a.weights_f32 else 2) * 1.0
    lm_loop_us, _ = model.time_kernel(_lib.TIME_LM_HEAD, 256)
    lm_us = lm_loop_us
    achieved = lm_bytes / (lm_us * 1e-6) / 1e9
    # HBM traffic per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
    # FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM): bench.py itself cannot collect counters.
    traffic, rocprof_us = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "lm_head_traffic.json")) as f:
            tj = json.load(f)
        if a.model == "124M" and ppg == 1 and not a.weights_f32:
            traffic, rocprof_us = tj["traffic_bytes_per_launch"], tj["rocprof_avg_us"]
    except Exception:
        pass
    gemm = None
    if a.model == "124M" and world == 1:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_gemm

            gemm = bench_gemm.measure(lib, 8192)
            _lib.check(lib.zg_set_stream(stream.cuda_stream))
        except Exception as e:  # the headline metric must not die with the secondary one
            gemm = {"error": str(e)}
    # the prompt side of generate (src/main.zig:331-334) as one pass: zg_gpt_prefill of a (ctx - 1)-token prompt
    prefill = None
    if world == 1 and not a.weights_f32 and ctx > 1:
        try:
            n_p = ctx - 1
            ptoks = np.stack([synth.rand_tokens(a.seed + 500 + b, n_p, cfg.vocab_size) for b in range(ppg)])
            model.prefill(ptoks, compute_logits=False)
            t0 = time.perf_counter()
            for _ in range(5):
                model.prefill(ptoks, compute_logits=False)
            p_ms = (time.perf_counter() - t0) / 5 * 1e3
            prefill = {"prompt_tokens": n_p, "prompts": ppg, "ms": round(p_ms, 3),
                       "prompt_tokens_per_s": round(ppg * n_p / p_ms * 1e3, 1),
                       "vs_token_at_a_time": round((1e3 * elapsed / a.steps) * n_p / ctx / p_ms, 1),
                       "how": "synchronous zg_gpt_prefill calls (host wall clock, 5 repetitions), activations split "
                              "3-way into bf16 for the MFMA GEMMs (exact), fp32-MFMA causal attention"}
        except Exception as e:
            prefill = {"error": str(e)}
    # whole-step view: algorithmic bytes of all ctx steps / device time
    kv_elem = 2 if a.kv_f16 else 4
    kv_total = sum(kv_elem * 2 * t * cfg.n_embed * cfg.n_layer * ppg for t in range(1, ctx + 1))
    step_bytes_total = wbytes * ctx + kv_total
    out = {
        "metric": "tokens/sec GPT-2-124M greedy 1024-ctx" if a.model == "124M" else f"tokens/sec {a.model} greedy {ctx}-ctx",
        "value": round(value, 1),
        "unit": "tokens/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(1e3 * elapsed / a.steps, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 weights+activations" if a.weights_f32 else "bf16 weights, f32 activations/accumulate",
        "data": "synthetic",
        "config": {
            "workload": f"GPT-2 {a.model} greedy decode, {ppg} prompt(s)/GPU x {world} GPU(s), 1-token prompts, "
                        f"{ctx} decode steps per prompt (reference generate loop, src/main.zig:322-342)",
            "vocab": cfg.vocab_size, "context": cfg.context_size, "n_layer": cfg.n_layer, "n_heads": cfg.n_heads,
            "n_embed": cfg.n_embed, "prompts_per_gpu": ppg, "global_prompts": ppg * world,
            "kv_cache": "f16" if a.kv_f16 else "f32", "hip_graph": not a.no_graph,
            "parallelism": f"replicated weights, prompts sharded x{world}, RCCL broadcast at start-up only",
            "tokens_counted": "generated tokens (context - prompt) per prompt",
        },
        "roofline": {
            "kernel": ("gemv_kernel<bf16,M=1,LPR16,CPL6,ARGMAX>" if ppg == 1 else "gemv_mfma_kernel<KS=6,NW=4,ARGMAX>") +
                      " (ln_f + lm_head + argmax)",
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "algorithmic_bytes_per_launch": int(lm_bytes), "avg_launch_us": round(lm_us, 2),
            "rocprof_avg_us_committed_profile": rocprof_us,
            "in_situ_event_interval_us": round(lm_interval_us, 2), "null_kernel_event_interval_us": round(null_us, 2),
            "how": "256 launches of the kernel replayed back to back from a hipGraph between one pair of HIP events "
                   "on the launch stream, right after the timed region (launch boundary included); in_situ_* are "
                   f"event-to-event intervals inside {2 * n_prof} complete decode steps (T=1.. and T={ctx - n_prof + 1}..)",
        },
        "step_roofline": {
            "bound": "hbm", "algorithmic_bytes_per_generation": int(step_bytes_total),
            "achieved": round(step_bytes_total * a.steps / dev_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(step_bytes_total * a.steps / dev_s / 1e9 / HBM_PEAK_GBS, 4),
            "us_per_token_device": round(1e6 * dev_s / (a.steps * ctx), 2),
            "per_kernel_class_us_eager_T_low": {k: round(v, 2) for k, v in prof_lo.items()},
            "per_kernel_class_us_eager_T_high": {k: round(v, 2) for k, v in prof_hi.items()},
        },
        "mfma_gemm_768x3072": gemm,
        "prefill": prefill,
        "device_time_s": round(dev_s, 4),
        "setup_s": round(setup_s, 2),
        "weight_broadcast_ms": None if bcast_ms is None else round(bcast_ms, 2),
        "first_tokens": [int(t) for t in ids[0, :8]],
    }
    if not a.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(cfg, weights, prompts[0], a.cpu_seconds)
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
