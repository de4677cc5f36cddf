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
import ctypes as C
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
    p.add_argument("--kv-b24", action="store_true", help="24-bit KV cache (bf16 plane + byte plane), ZG_GPT_KV_B24")
    p.add_argument("--weights-f32", action="store_true")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--no-prefetch", action="store_true", help="no side-stream L2 prefetcher beside the decode chain (A/B)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-op-tier", action="store_true", help="skip the timing of the literal drop-in (zgpt2_main over the op tier)")
    p.add_argument("--no-other-configs", action="store_true", help="skip the secondary block (8 prompts, XL, nano-char)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for the CPU baseline sample")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--dry-run", action="store_true",
                   help="CPU rehearsal of the launch path (gloo, no GPU, no kernels): rendezvous, prompt sharding, "
                        "weight-arena broadcast, barriers and the max-reduce; prints a line with dry_run=true")
    return p.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a fresh child
    process BEFORE anything in this process touches the GPU, forward rank 0's JSON line, propagate failure."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    for l in out.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if out.returncode != 0 or len(lines) != 1:
        print(f"bench.py: the {a.gpus}-rank launch failed (rc {out.returncode}, {len(lines)} result lines)", file=sys.stderr)
        sys.exit(out.returncode or 1)
    d = json.loads(lines[0])
    if d.get("n_gpus") != a.gpus:
        print(f"bench.py: asked for {a.gpus} GPUs, the ranks report {d.get('n_gpus')}", file=sys.stderr)
        sys.exit(1)
    print(lines[0], flush=True)
    sys.exit(0)


def scaling_fields(dist, torch, device, world, value, ref_value, own_rate, bcast_ms, bcast_bytes):
    """What makes an N > 1 line self-contained: the one-GPU figure with the SAME per-GPU work (rank 0 alone, timed before the
    collective region), the efficiency it implies, the spread over the ranks, and the rate of the one collective."""
    lo = torch.tensor([own_rate], dtype=torch.float64, device=device)
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    ref = torch.tensor([ref_value if ref_value else 0.0], dtype=torch.float64, device=device)
    dist.broadcast(ref, src=0)
    ref_value = float(ref.item())
    return {
        "scaling_reference": {"n_gpus": 1, "value": round(ref_value, 1), "unit": "tokens/s",
                              "how": "rank 0 alone, one generation of its per-GPU load after one warm-up generation, the other ranks "
                                     "waiting at a barrier; same handle, same prompts as in the timed region"},
        "scaling_efficiency": round(value / (world * ref_value), 4) if ref_value > 0 else None,
        "per_rank_tokens_per_s": {"min": round(float(lo.item()), 1), "max": round(float(hi.item()), 1)},
        "weight_broadcast_GBps": round(bcast_bytes / bcast_ms / 1e6, 2) if bcast_ms and bcast_ms > 0 else None,
    }


def dry_run(a, rank, world):
    """The N-rank control flow of main() on CPU: same sharding helpers, same collectives, same line (the scaling fields
    included), gloo instead of RCCL; a "generation" is a fixed amount of numpy arithmetic."""
    import torch
    import torch.distributed as dist

    from zig_gpt2_amd import shard, synth

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = synth.CONFIGS["tiny"]
    ppg = a.prompts_per_gpu or (1 if world == 1 else 8)
    ctx = a.ctx or cfg.context_size
    n = 1 << 20
    arena = torch.full((n,), 7, dtype=torch.uint8) if rank == 0 else torch.zeros(n, dtype=torch.uint8)
    dist.barrier()
    t0 = time.perf_counter()
    shard.broadcast_weights(arena, dist, src=0)
    bcast_ms = (time.perf_counter() - t0) * 1e3
    assert int(arena[-1]) == 7
    mine = shard.shard_prompts(ppg * world, world, rank)
    assert len(mine) == ppg
    work = np.ones((64, 64))

    def one_generation():
        x = work
        for _ in range(40):
            x = x @ work * 1e-2
        return x

    ref_value = None
    if rank == 0:  # the one-GPU reference: rank 0 alone
        one_generation()
        t = time.perf_counter()
        one_generation()
        ref_value = ppg * (ctx - 1) / (time.perf_counter() - t)
    dist.barrier()
    for _ in range(a.warmup):
        one_generation()
    dist.barrier()
    t_wall = time.perf_counter()
    for _ in range(a.steps):
        one_generation()
    own_s = time.perf_counter() - t_wall
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t_wall], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    value = world * ppg * (ctx - 1) * a.steps / elapsed
    extra = scaling_fields(dist, torch, torch.device("cpu"), world, value, ref_value, ppg * (ctx - 1) * a.steps / own_s, bcast_ms, n)
    if rank == 0:
        print(json.dumps({"metric": "tokens/sec GPT-2-124M greedy 1024-ctx", "value": round(value, 1), "unit": "tokens/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dry_run": True, "data": "none (CPU rehearsal of the launch path: a generation is numpy arithmetic)",
                          "config": {"workload": f"dry run, {ppg} prompt(s)/rank x {world} rank(s), model {cfg.n_embed}-wide",
                                     "prompts_per_gpu": ppg, "global_prompts": ppg * world},
                          "weight_broadcast_ms": round(bcast_ms, 3), "max_over_ranks_s": elapsed, **extra}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


class _DevMem:
    """Expose a raw device allocation to torch (for the RCCL broadcast of the weight arena)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def cpu_baseline(cfg, weights, prompt, budget_s):
    """The oracle (C restatement of the reference CPU path, fp32, all host cores) timed on the same workload: the whole
    1..ctx run when that fits about 2.5 x the budget (GPT-2 124M on the GPU box: ~5 s), else a window of early positions and
    a window of late positions — the per-token cost is affine in T (KV re-transpose + attention, src/ops.zig:153-160), so the
    whole run is priced from the two windows."""
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

    # BASELINE configs[0]: 1 prompt, 64 tokens (positions 1..64), with each SGEMM available: the built-in OpenMP one and a
    # real CBLAS if one can be dlopen'ed on this box (the reference links Accelerate / OpenBLAS, build.zig:28-32).  The whole
    # 1..ctx run (wall clock over the full run) when it fits about 2.5 x the budget — on the GPU box's 16 cores
    # 124M takes about five seconds — otherwise two windows priced by an affine fit (the per-token cost is affine in T).
    def whole_or_windows(name, t_first):
        if t_first * ctx * 1.6 <= 2.5 * budget_s:
            tok = int(prompt[0])
            t = time.perf_counter()
            for T in range(1, ctx + 1):
                m.forward(T, tok, True)
            total = time.perf_counter() - t
            return total / ctx, f"oracle GPT.forward fp32, {name}: the whole run, positions 1..{ctx} with lm_head at every position, {total:.2f} s wall clock"
        n_win = min(max(8, int(budget_s / 2 / max(min(t_first, 1.0), 1e-3) / 1.5)), 96)
        lo0, hi0 = 1, max(1, ctx - n_win)
        t_lo, t_hi = window(lo0, n_win), window(hi0, n_win)
        # affine model: cost(T) = a + b*T, fitted at the window centres, averaged over T = 1..ctx
        c_lo, c_hi = lo0 + (n_win - 1) / 2, hi0 + (n_win - 1) / 2
        b = (t_hi - t_lo) / max(c_hi - c_lo, 1.0)
        a = t_lo - b * c_lo
        return a + b * (ctx + 1) / 2, (f"oracle GPT.forward fp32, {name}: {n_win} tokens at T={lo0}.. ({1e3 * t_lo:.1f} ms/tok) and "
                                       f"{n_win} at T={hi0}.. ({1e3 * t_hi:.1f} ms/tok); whole 1..{ctx} run priced by the affine fit")

    t64 = {}
    window(1, 4)  # warm
    t_builtin = t64["built-in OpenMP sgemm"] = window(1, min(64, ctx))
    mean_cost, sample = whole_or_windows(blas, t_builtin)  # (before a CBLAS is loaded: its thread pool would spin beside OpenMP's)
    blas_found = None
    if found and oracle.use_cblas(found[0], found[1]) == 0:
        blas_found = found[2]
        window(1, 4)
        t_blas = t64[blas_found] = window(1, min(64, ctx))
        if t_blas < t_builtin:  # the faster of the two over the 64-token window is the baseline
            c2, s2 = whole_or_windows(blas_found, t_blas)
            if c2 < mean_cost:
                blas, mean_cost, sample = blas_found, c2, s2
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
        "cblas_found": blas_found if blas_found else "not found (no libopenblas / libcblas to dlopen on this box)",
        "first_64_tokens_tokens_per_s": {k: round(1.0 / v, 2) for k, v in t64.items()},
        "sample": sample,
    }


def op_tier(model_name, seed, prompt, ctx, cpu_value):
    """The LITERAL drop-in timed: src/main.zig's dataflow (State, Block.forward, GPT.forward, generate with greedy argmax) over
    the op tier — host buffers, one synchronous FFI call per op, 77 per token at 124M — run by the compiled C++ caller
    zig_gpt2_amd/bin/zgpt2_main as a child process (what an unchanged main.zig over zig/ops.zig would do; fp32 weights
    mirrored once at Linear.init, caller-owned KV caches mirrored on the device).  The whole 1..ctx run, no extrapolation."""
    import subprocess

    exe = os.path.join(ROOT, "zig_gpt2_amd", "bin", "zgpt2_main")
    if not os.path.exists(exe):
        return {"error": f"{exe} is not built"}
    args = [exe, model_name, str(seed), ",".join(str(int(t)) for t in prompt), str(ctx)]
    t0 = time.perf_counter()
    out = subprocess.run(args, capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stderr.splitlines() if l.startswith("{")]
    if out.returncode != 0 or not lines:
        return {"error": f"zgpt2_main rc {out.returncode}: {out.stderr[-300:]}"}
    d = json.loads(lines[-1])
    ids = [int(t) for t in out.stdout.split()]
    return {"value": d["tokens_per_s"], "unit": "tokens/s", "steps": d["steps"], "seconds": d["seconds"],
            "first_64_tokens_per_s": d["first_64_tokens_per_s"], "calls_per_token": 2 + 6 * synth_layers(model_name) + 2,
            "vs_cpu_baseline": round(d["tokens_per_s"] / cpu_value, 2) if cpu_value else None,
            "first_tokens": ids[:8], "process_seconds": round(time.perf_counter() - t0, 1),
            "how": "zgpt2_main <model> <seed> <prompt> <ctx> (op tier: the default), timing of its generate loop from its stderr; "
                   "weights = the same synthetic seed as the headline run, kept fp32 as the reference keeps them"}


def synth_layers(model_name):
    from zig_gpt2_amd import synth

    return synth.CONFIGS[model_name].n_layer


def kernel_table(model, lib, cfg, ppg, wsz, kv_elem):
    """Every kernel class of the decode step timed live: HIP events on the launch stream around a hipGraph chain of 64
    launches of that one kernel at a mid-context control block (launch boundary included), priced per token by its launch
    count.  Returns (table, dominant class, lm_head row)."""
    from zig_gpt2_amd import _lib

    E, L, V = cfg.n_embed, cfg.n_layer, cfg.vocab_size
    t_mid = max(cfg.context_size // 2, 1)
    classes = [  # (time_kernel id, name, launches per token, algorithmic bytes per launch)
        (1, "ln_1 + c_attn + KV append", L, 3 * E * E * wsz),
        (2, "attention (split-KV decode)", L, 2 * t_mid * E * kv_elem * ppg),
        (3, "head merge + attn c_proj + residual", L, E * E * wsz),
        (4, "ln_2 + c_fc + GELU", L, 4 * E * E * wsz),
        (5, "mlp c_proj + residual", L, 4 * E * E * wsz),
        (6, "ln_f + lm_head + argmax", 1, V * E * wsz),
    ]
    table = []
    for which, name, n_launch, nbytes in classes:
        us, _ = model.time_kernel(which, 256)
        us_cold, _ = model.time_kernel(which, 256, walk_layers=True)  # weights / KV from the memory side, as in the real step
        sym = C.create_string_buffer(160)
        _lib.check(lib.zg_debug_last_kernel(sym, 160))
        table.append({"class": name, "kernel_symbol": sym.value.decode(), "launches_per_token": n_launch, "avg_launch_us": round(us, 3),
                      "avg_launch_us_layers_walked": round(us_cold, 3),
                      "algorithmic_bytes_per_launch": int(nbytes), "GBps": round(nbytes / us / 1e3, 1),
                      "frac_of_8TBps": round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4), "us_per_token": round(us * n_launch, 2)})
    tot_us = sum(r["us_per_token"] for r in table)
    for r in table:
        r["share_of_token_time"] = round(r["us_per_token"] / tot_us, 4)
    return table, max(table, key=lambda r: r["us_per_token"]), table[-1]


def device_weights(cfg, seed):
    """bf16-representable N(mean, 0.02^2) weights generated ON the GPU (torch): the secondary configurations are timed, not
    compared with the oracle, and numpy needs half a minute for GPT-2 XL's 1.5 G parameters."""
    import torch

    from zig_gpt2_amd import synth

    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    w = {}
    for name, shape, mean, _ in synth.tensor_specs(cfg):
        t = torch.randn(shape, generator=gen, device="cuda", dtype=torch.float32) * 0.02 + mean
        w[name] = t.to(torch.bfloat16).to(torch.float32).contiguous()
    return w


def other_config(lib, stream, model_name, ppg, gens, seed, kv_b24=False):
    """One of BASELINE.json's other single-GPU workloads, driver-timed in the same run: tokens/s of `gens` full greedy
    generations (after one warm-up generation), device time per forward, whole-step roofline fraction, dominant kernel class."""
    import torch

    from zig_gpt2_amd import gpt, synth

    cfg = synth.CONFIGS[model_name]
    ctx = cfg.context_size
    model = gpt.GPT(cfg, batch=ppg, kv_b24=kv_b24)
    kvb = 3 if kv_b24 else 4
    try:
        w = device_weights(cfg, seed)
        model.load_weights(w)
        del w
        prompts = [synth.rand_tokens(2000 + seed * 131 + b, 1, cfg.vocab_size) for b in range(ppg)]
        model.generate_enqueue(prompts, ctx)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        for _ in range(gens):
            model.generate_enqueue(prompts, ctx)
        e1.record(stream)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        dev_s = e0.elapsed_time(e1) / 1e3
        ids = model.generate_fetch(ctx)
        pf = model.prefetch_stats()
        wbytes, _ = model.step_bytes(1)
        kv_total = sum(kvb * 2 * t * cfg.n_embed * cfg.n_layer * ppg for t in range(1, ctx + 1))
        step_bytes_total = wbytes * ctx + kv_total
        table, dom, lm = kernel_table(model, lib, cfg, ppg, 2, kvb)
        # the whole-prompt pass of this configuration: ctx - 1 tokens per prompt (synchronous calls, host wall clock)
        ptoks = np.stack([synth.rand_tokens(3000 + seed * 17 + b, ctx - 1, cfg.vocab_size) for b in range(ppg)])
        for _ in range(2):
            model.prefill(ptoks, compute_logits=False)
        t0 = time.perf_counter()
        for _ in range(3):
            model.prefill(ptoks, compute_logits=False)
        pf_ms = (time.perf_counter() - t0) / 3 * 1e3
        return {
            "workload": f"GPT-2 {model_name} greedy decode, {ppg} prompt(s) on one GPU, 1-token prompts, {ctx} decode steps per prompt",
            "value": round(ppg * (ctx - 1) * gens / wall, 1), "unit": "tokens/s", "generations": gens, "ms_per_generation": round(1e3 * wall / gens, 3),
            "us_per_forward_device": round(1e6 * dev_s / (gens * ctx), 2),
            "kv_cache": "b24 (ZG_GPT_KV_B24: 24-bit elements, 4.7e-5 of the logit scale at full context)" if kv_b24 else "f32",
            "l2_prefetcher": ("stalled" if pf["stalled"] else "on") if pf["on"] else "off",
            "step_roofline": {"bound": "hbm", "algorithmic_bytes_per_generation": int(step_bytes_total),
                              "achieved": round(step_bytes_total * gens / dev_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(step_bytes_total * gens / dev_s / 1e9 / HBM_PEAK_GBS, 4)},
            "dominant_class": {k: dom[k] for k in ("class", "kernel_symbol", "avg_launch_us", "avg_launch_us_layers_walked", "algorithmic_bytes_per_launch",
                                                   "GBps", "frac_of_8TBps", "share_of_token_time")},
            "lm_head": {k: lm[k] for k in ("kernel_symbol", "avg_launch_us", "GBps", "frac_of_8TBps")},
            "prefill": {"prompt_tokens": ctx - 1, "prompts": ppg, "ms": round(pf_ms, 3), "prompt_tokens_per_s": round(ppg * (ctx - 1) / pf_ms * 1e3, 1),
                        **prefill_flops(cfg, ppg, ctx - 1, pf_ms)},
            # timed only.  bench.py may use the oracle in its cpu_baseline leg and nowhere else; the same shapes are held to the oracle
            # at full context by pytest (tests/test_full_configs_gpu.py, test_prefill_gpu.py)
            "checked": False,
            "data": "synthetic (torch.randn on the GPU, bf16-representable)", "first_tokens": [int(t) for t in ids[0, :4]],
        }
    finally:
        model.close()


def groups_table(seed, n_prompts=8, group_counts=(1, 2, 4, 8)):
    """BASELINE configs[2]'s per-GPU load (8 independent prompts at 124M) as G co-running groups of 8 / G sequences — one handle
    per group on its own stream, one shared weight region, a feeder thread per handle (gpt.GPTGroups, zg_gpt_generate_enqueue_many)
    — against the lock-step batch (G = 1): one row per G, ids compared row for row with G = 1.  Measured in every run because the
    answer decides the per-GPU load of --gpus N: co-running loses at every G (the chip overlaps two chains by a factor 1.6-2.4,
    the lock-step batch packs eight into 1.4 x the time of one), so the load stays one lock-step handle.
    Runs tools/experiments/corun_ab.py as a CHILD process, BEFORE this process touches the GPU: how well hardware queues overlap
    depends on who else holds queues on the device — inside this program, after its own earlier work, the same table has read
    2.2 ms per step round at G = 2 instead of 0.35 (profiles/NOTEBOOK.md §5.1) — so the table is taken where a user of GPTGroups
    would take it: on a device that does nothing else."""
    import subprocess

    exe = os.path.join(ROOT, "tools", "experiments", "corun_ab.py")
    best = {}
    # both ways of dealing stream priorities to the groups (cycle: normal / high / low, which puts three groups on three different
    # hardware queues; normal: all alike) — which one is better depends on G, the table keeps the better one per G
    for prio in ("cycle", "normal"):
        out = subprocess.run([sys.executable, exe, "--prompts", str(n_prompts), "--gens", "1", "--prio", prio] + [str(g) for g in group_counts],
                             capture_output=True, text=True, timeout=600)
        rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or len(rows) != len(group_counts):
            raise RuntimeError(f"corun_ab.py rc {out.returncode}: {out.stderr[-300:]}")
        for d in rows:
            row = {"workload": f"124M, {n_prompts} prompts as {d['groups']} x {d['per_group']}", "groups": d["groups"], "sequences_per_group": d["per_group"],
                   "value": d["tokens_per_s"], "unit": "tokens/s", "us_per_step_round": d["us_per_step"], "ids_equal_lock_step": d["ids_equal_first"],
                   "stream_priorities": d["priorities"] if d["groups"] > 1 else "(one handle on the library stream)"}
            if d["groups"] not in best or row["value"] > best[d["groups"]]["value"]:
                best[d["groups"]] = row
    return [best[g] for g in group_counts]


def prefill_flops(cfg, prompts, n, ms, planes=3):
    """Matrix-core work of one whole-prompt pass: the useful Linear FLOPs (2 M K N each), and what the MFMAs execute — `planes`
    bf16 plane products per Linear (exact split of the fp32 activations), six per attention product (both operands split),
    causal attention counted over the 32 x 32 tiles it visits."""
    E, L, H = cfg.n_embed, cfg.n_layer, cfg.n_heads
    lin = 2.0 * prompts * n * 12 * E * E * L
    nqb = (n + 31) // 32
    tiles = nqb * (nqb + 1) // 2
    attn = 2.0 * 2 * tiles * 32 * 32 * 64 * H * prompts * L  # S = q k^T and P v per visited tile
    return {"linear_tflops_useful": round(lin / ms / 1e9, 1), "mfma_tflops": round((planes * lin + 6 * attn) / ms / 1e9, 1),
            "mfma_frac_of_2.5PF": round((planes * lin + 6 * attn) / ms / 1e9 / 2500.0, 4)}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    if a.dry_run:
        return dry_run(a, rank, world)
    # The two measurements that run as CHILD processes are taken first, while this process holds no GPU queue: both are sensitive
    # to who else has queues on the device (the co-running groups: DESIGN §3.3; the op tier polls a completion word behind every
    # call — under a parent with a dozen idle queues it has read 600 tok/s where it reads 770-850 alone).
    early = {}
    if world == 1 and a.gpus == 1:
        from zig_gpt2_amd import synth as _synth

        cfg0 = _synth.CONFIGS[a.model]
        ctx0 = a.ctx or cfg0.context_size
        ppg0 = a.prompts_per_gpu or 1
        if not a.no_op_tier and a.model in ("124M", "nano-char", "tiny") and not a.weights_f32:
            try:
                early["op_tier"] = op_tier(a.model, a.seed, _synth.rand_tokens(1000 + a.seed * 131, 1, cfg0.vocab_size), ctx0, None)
            except Exception as e:
                early["op_tier"] = {"error": str(e)}
        if a.model == "124M" and ppg0 == 1 and not a.weights_f32 and not a.kv_f16 and not a.kv_b24 and not a.no_other_configs:
            try:
                t_o = time.perf_counter()
                early["groups"] = {"workload": "124M, 8 prompts on one GPU as G co-running groups (table)", "groups_on_one_gpu": groups_table(a.seed + 7),
                                   "seconds_spent": round(time.perf_counter() - t_o, 1)}
            except Exception as e:
                early["groups"] = {"workload": "124M, 8 prompts as G groups", "error": str(e)}
    import torch

    from zig_gpt2_amd import _lib, gpt, shard, synth

    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # Test hooks (tests/test_bench_gpu.py): ZGPT2_ALL_RANKS_ON_DEVICE=d puts every rank on device d, ZGPT2_RCCL_LIB names a stand-in
    # for RCCL (tests/stub_rccl) — then torch's own collectives run over gloo on CPU tensors, since RCCL refuses two ranks on one
    # GPU.  Everything else of the N > 1 path is the code a multi-GPU node runs.
    one_dev = os.environ.get("ZGPT2_ALL_RANKS_ON_DEVICE")
    dev_index = int(one_dev) if one_dev is not None else local_rank
    backend = "gloo" if os.environ.get("ZGPT2_RCCL_LIB") else "nccl"
    coll_dev = torch.device("cuda", dev_index) if backend == "nccl" else torch.device("cpu")
    torch.cuda.set_device(dev_index)
    dist = None
    # ZGPT2_FORCE_DIST=1 runs the RCCL path (process group, weight broadcast, barriers, max-reduce) with a
    # single rank: the only way to exercise it on a one-GPU box
    use_dist = world > 1 or os.environ.get("ZGPT2_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")

    lib = _lib.load()  # no fallback: raises if libzgpt2_hip.so is missing
    _lib.check(lib.zg_init(dev_index))
    stream = torch.cuda.Stream()
    _lib.check(lib.zg_set_stream(stream.cuda_stream))

    cfg = synth.CONFIGS[a.model]
    ctx = a.ctx or cfg.context_size
    ppg = a.prompts_per_gpu or (1 if world == 1 else 8)
    model = gpt.GPT(cfg, batch=ppg, weights_f32=a.weights_f32, use_graph=not a.no_graph, kv_f16=a.kv_f16, kv_b24=a.kv_b24, prefetch=not a.no_prefetch)

    # ---- weights: generated and uploaded on rank 0, broadcast to the other GPUs over RCCL/xGMI
    weights = None
    t0 = time.perf_counter()
    if rank == 0:
        weights = synth.make_weights(cfg, seed=a.seed, bf16=not a.weights_f32)
        model.load_weights(weights)
    # The one collective of the multi-GPU case goes through the library's own RCCL path (zg_dist_* / zg_gpt_broadcast_weights,
    # include/zgpt2.h): rank 0 makes the communicator id, torch.distributed only ships its 128 bytes.  With one rank the same calls
    # run on a one-rank communicator, so the line always carries the device time of the broadcast.
    import ctypes as C

    bcast_ms = None
    bcast_note = None
    # (RCCL prints a version banner through C stdio when a communicator is made; this program's stdout carries ONE JSON line, so
    # file descriptor 1 points at stderr while the library's RCCL calls run, and C's buffers are flushed before it is restored)
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    def all_ranks_ok(ok):  # every rank takes the same branch: the native path only if it worked everywhere
        if not use_dist:
            return ok
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    class NativeDistError(RuntimeError):
        pass

    native_ok = False
    try:
        # zg_dist_init is a rendezvous (ncclCommInitRank): every rank enters it or none does.  So the ranks first agree that
        # RCCL can be bound everywhere (zg_dist_available is not a collective), and only rank 0's id failing is left to ship.
        ok, err = True, None
        try:
            _lib.check(lib.zg_dist_available())
        except _lib.ZgError as e:
            ok, err = False, str(e)
        if not all_ranks_ok(ok):
            raise NativeDistError(err or "RCCL cannot be bound on another rank")
        uid = (C.c_ubyte * 128)()
        if rank == 0:
            try:
                _lib.check(lib.zg_dist_unique_id(uid, 128))
            except _lib.ZgError as e:
                err = str(e)
        if use_dist:
            box = [None if err else bytes(uid), err]
            dist.broadcast_object_list(box, src=0)
            err = box[1]
            if err is None:
                uid = (C.c_ubyte * 128).from_buffer_copy(box[0])
        if err is not None:
            raise NativeDistError(err)
        try:
            _lib.check(lib.zg_dist_init(uid, 128, rank, world))
        except _lib.ZgError as e:
            ok, err = False, str(e)
        if not all_ranks_ok(ok):
            raise NativeDistError(err or "zg_dist_init failed on another rank")
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        ms = C.c_float(0.0)
        try:
            _lib.check(lib.zg_gpt_broadcast_weights(model.h, 0, C.byref(ms)))
        except _lib.ZgError as e:
            ok, err = False, str(e)
        if not all_ranks_ok(ok):
            raise NativeDistError(err or "zg_gpt_broadcast_weights failed on another rank")
        bcast_ms = float(ms.value)
        _lib.check(lib.zg_dist_finalize())
        native_ok = True
    except (NativeDistError, _lib.ZgError) as e:
        lib.zg_dist_finalize()  # a communicator made before the failure must not outlive it (no-op without one)
        bcast_note = f"native RCCL path unavailable ({e})"  # (a one-GPU box without librccl: nothing to broadcast to)
    finally:
        C.CDLL(None).fflush(None)
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if not native_ok and world > 1:
        # the run must still measure the step: the same weight region through torch.distributed's broadcast (RCCL as well), and
        # the line says so
        ptr, nbytes = model.weight_arena()
        arena = torch.as_tensor(_DevMem(ptr, nbytes), device=torch.device("cuda", dev_index))
        torch.cuda.synchronize()
        dist.barrier()
        tb = time.perf_counter()
        shard.broadcast_weights(arena, dist, src=0)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - tb) * 1e3
        bcast_note += "; weights broadcast by torch.distributed.broadcast over the same arena region (host wall clock)"
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

    # N > 1: the one-GPU figure with the same per-GPU work, taken by rank 0 alone before the collective region
    ref_value = None
    if dist is not None:
        if rank == 0:
            one_generation()
            torch.cuda.synchronize()
            t_ref = time.perf_counter()
            one_generation()
            torch.cuda.synchronize()
            ref_value = ppg * (ctx - 1) / (time.perf_counter() - t_ref)
        dist.barrier()
    for _ in range(a.warmup):
        one_generation()
    sync_all()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_wall = time.perf_counter()
    e0.record(stream)
    for _ in range(a.steps):
        one_generation()
    e1.record(stream)
    torch.cuda.synchronize()
    own_s = time.perf_counter() - t_wall  # this rank's own K generations (before it waits for the others)
    sync_all()
    wall_s = time.perf_counter() - t_wall
    dev_s = e0.elapsed_time(e1) / 1e3
    elapsed = wall_s
    if dist is not None:
        t = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ids = model.generate_fetch(ctx)
    pf = model.prefetch_stats()  # how the side-stream prefetcher of the last generation ended (after the timed region)

    # tokens produced per generation: every position after the prompt is one generated token
    gen_tokens = ppg * (ctx - 1)
    value = world * gen_tokens * a.steps / elapsed

    scaling_extra = {}
    if dist is not None:
        _, wbytes_region = model.weight_arena()
        scaling_extra = scaling_fields(dist, torch, coll_dev, world, value, ref_value,
                                       gen_tokens * a.steps / own_s, bcast_ms, wbytes_region)
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline: every kernel class of the decode step timed live (HIP events on the launch stream around a
    # hipGraph chain of 64 launches of that one kernel, mid-context control block), priced per token by its launch
    # count; `roofline` describes the class with the LARGEST share of the token time, lm_head and the whole step
    # are reported beside it.
    wbytes, _ = model.step_bytes(1)
    wsz = 4 if a.weights_f32 else 2
    kv_elem = 2 if a.kv_f16 else 3 if a.kv_b24 else 4
    table, dom, lm = kernel_table(model, lib, cfg, ppg, wsz, kv_elem)
    n_prof = min(64, ctx)
    prof_lo = model.profile_step(1, n_prof)
    prof_hi = model.profile_step(ctx - n_prof + 1, n_prof)
    # HBM traffic per launch: PMC counters cannot be read from inside this process; the committed rocprofv3 passes
    # (profiles/<round>_traffic.json: FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE, per launch, taken at the
    # same shapes) are quoted with their provenance, null when the file has no entry for this kernel / config.
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tj = json.load(f)
        key = f"{a.model}/{ppg}/{'f32' if a.weights_f32 else 'bf16'}"
        ent = tj.get(key, {}).get(dom["class"])
        if ent:
            traffic, traffic_src = ent["bytes_per_launch"], ent["source"]
    except Exception:
        pass
    gemm = None
    if a.model == "124M" and world == 1:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_gemm

            gemm = bench_gemm.measure(lib, 8192)  # BASELINE's point; the other heights beside it (same warm-up: the clocks need it)
            keep = ("M", "N", "K", "gelu", "out", "us", "us_min", "tflops", "mfma_frac_of_2.5PF")
            gemm["other_M"] = [{k: v for k, v in bench_gemm.measure(lib, m).items() if k in keep} for m in (64, 1024, 16384)]  # M in {64, 1024, 8192}
            # the account beside the headline (same kernel, same launch path): the epilogue's share (bias only, no GELU), and the
            # MLP's OTHER 768 x 3072 Linear — mlp c_proj, N = 768, K = 3072 (src/main.zig:81) — whose tiles see 4 x the K per
            # epilogue: whole K at M = 16384 (256 tiles of 256 x 192 = one per CU); at M = 8192 its 128 tiles fill half the chip
            gemm["account"] = {
                "c_fc_bias_only_M8192": {k: v for k, v in bench_gemm.measure(lib, 8192, gelu=False).items() if k in keep},
                "mlp_c_proj_N768_K3072_M16384": {k: v for k, v in bench_gemm.measure(lib, 16384, n=768, k=3072, gelu=False).items() if k in keep},
                "mlp_c_proj_N768_K3072_M8192_half_chip": {k: v for k, v in bench_gemm.measure(lib, 8192, n=768, k=3072, gelu=False).items() if k in keep},
                "note": "FLOP-based fractions of 2.5 PF; the counter-based MFMA-busy of the same launches is in profiles/round6_gemm_account_pmc.md",
            }
            _lib.check(lib.zg_set_stream(stream.cuda_stream))
        except Exception as e:  # the headline metric must not die with the secondary one
            gemm = {"error": str(e)}
    # the prompt side of generate (src/main.zig:331-334) as one pass: zg_gpt_prefill of a (ctx - 1)-token prompt
    prefill = None
    if world == 1 and ctx > 1:
        try:
            n_p = ctx - 1
            ptoks = np.stack([synth.rand_tokens(a.seed + 500 + b, n_p, cfg.vocab_size) for b in range(ppg)])
            lin_flops = 2.0 * ppg * n_p * (12 * cfg.n_embed * cfg.n_embed) * cfg.n_layer  # the four Linears of every Block

            def time_prefill(mdl):
                for _ in range(3):
                    mdl.prefill(ptoks, compute_logits=False)
                t0 = time.perf_counter()
                for _ in range(5):
                    mdl.prefill(ptoks, compute_logits=False)
                return (time.perf_counter() - t0) / 5 * 1e3

            p_ms = time_prefill(model)
            prefill = {"prompt_tokens": n_p, "prompts": ppg, "ms": round(p_ms, 3),
                       "prompt_tokens_per_s": round(ppg * n_p / p_ms * 1e3, 1),
                       **prefill_flops(cfg, ppg, n_p, p_ms),
                       "vs_token_at_a_time": round((1e3 * elapsed / a.steps) * n_p / ctx / p_ms, 1),
                       "how": "synchronous zg_gpt_prefill calls (host wall clock, 5 repetitions after 3 warm-ups); "
                              + ("fp32 weights: both GEMM operands as exact bf16 plane triples, the six plane products as three "
                                 "passes of one launch of the 128-row prompt GEMM per Linear" if a.weights_f32 else
                                 "activations split exactly 3-way into bf16 for the MFMA GEMMs") + ", causal attention on the bf16 matrix cores with exact three-plane splits of q, k, v and the probabilities (six plane products per matrix product)"}
            if not a.weights_f32:  # the two-plane mode (inside north_star's 1e-3, outside the tests' near-zero floor)
                m2 = gpt.GPT(cfg, batch=ppg, use_graph=False, kv_f16=a.kv_f16, kv_b24=a.kv_b24, prefill_planes=2)
                m2.load_weights(weights)
                p2_ms = time_prefill(m2)
                m2.close()
                prefill["two_plane"] = {"ms": round(p2_ms, 3), "prompt_tokens_per_s": round(ppg * n_p / p2_ms * 1e3, 1),
                                        "linear_tflops_useful": round(lin_flops / p2_ms / 1e9, 1)}
        except Exception as e:
            prefill = {"error": str(e)}
    # generate as the reference runs it (src/main.zig:322-342: every token drawn by GPT.sample, temp 0.8 in main): the device loop
    # with the sampler as a node of the captured step, and the per-token entry point zg_gpt_sample it replaces
    sampled = None
    if world == 1:
        try:
            model.generate_sample(prompts, ctx, 0.8, seed=a.seed)  # (captures the sampled graphs)
            torch.cuda.synchronize()
            t_s = time.perf_counter()
            ids_s = model.generate_sample(prompts, ctx, 0.8, seed=a.seed)
            s_wall = time.perf_counter() - t_s
            n_pt = min(ctx, 128)
            toks = [int(p[0]) for p in prompts]
            model.sample(1, toks, 0.8, seed=a.seed)
            t_s = time.perf_counter()
            for T in range(1, n_pt + 1):
                toks = [int(t) for t in model.sample(T, toks, 0.8, seed=a.seed)]
            pt_wall = time.perf_counter() - t_s
            sampled = {"temperature": 0.8, "device_loop_tokens_per_s": round(ppg * (ctx - 1) / s_wall, 1), "ms_per_generation": round(1e3 * s_wall, 3),
                       "per_token_call_tokens_per_s": round(ppg * n_pt / pt_wall, 1), "per_token_positions": n_pt,
                       "distinct_tokens_in_row_0": int(len(set(int(t) for t in ids_s[0]))),
                       "how": "zg_gpt_generate_sample (sampler node in the captured decode step, uniforms from the library's counter PRNG) against a "
                              "host loop over zg_gpt_sample; identical tokens for identical seeds (tests/test_sampled_generate_gpu.py)"}
        except Exception as e:
            sampled = {"error": str(e)}
    # BASELINE.json configs[2..4] on this GPU, one handle after another (the headline handle stays: its numbers are above)
    others = None
    if a.model == "124M" and world == 1 and ppg == 1 and not a.weights_f32 and not a.kv_f16 and not a.kv_b24 and not a.no_other_configs:
        others = []
        for mname, mp, gens, b24 in (("124M", 8, 2, False), ("124M", 8, 2, True), ("xl", 1, 1, False), ("nano-char", 1, 3, False)):
            try:
                t_o = time.perf_counter()
                o = other_config(lib, stream, mname, mp, gens, a.seed + 7, kv_b24=b24)
                o["seconds_spent"] = round(time.perf_counter() - t_o, 1)
                others.append(o)
            except Exception as e:
                others.append({"workload": f"{mname} x {mp}", "error": str(e)})
        if "groups" in early:
            others.append(early["groups"])
        _lib.check(lib.zg_set_stream(stream.cuda_stream))
    # whole-step view: algorithmic bytes of all ctx steps / device time
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
            "kv_cache": "f16" if a.kv_f16 else "b24" if a.kv_b24 else "f32", "hip_graph": not a.no_graph,
            "l2_prefetcher": ("stalled" if pf["stalled"] else "on") if pf["on"] else "off",
            "parallelism": f"replicated weights, prompts sharded x{world}, RCCL broadcast at start-up only",
            "tokens_counted": "generated tokens (context - prompt) per prompt",
            "scaling_note": ("N = 1 runs BASELINE configs[1] (one prompt: the headline metric); N > 1 runs configs[2] (8 prompts per GPU, "
                             "fixed per-GPU work for every N >= 2).  The one-GPU figure with THAT per-GPU work is other_configs[0].value of the "
                             "N = 1 line (8 prompts on one GPU), which is what an N-GPU value divides by for a scaling efficiency; "
                             "--prompts-per-gpu pins the per-GPU work for all N") if a.prompts_per_gpu == 0 else
                            f"per-GPU work pinned by --prompts-per-gpu {ppg} for every N",
        },
        "roofline": {
            "kernel": dom["class"], "kernel_symbol": dom["kernel_symbol"], "bound": "hbm", "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": dom["frac_of_8TBps"], "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"], "avg_launch_us": dom["avg_launch_us"],
            "avg_launch_us_layers_walked": dom["avg_launch_us_layers_walked"],
            "achieved_layers_walked": round(dom["algorithmic_bytes_per_launch"] / dom["avg_launch_us_layers_walked"] / 1e3, 1),
            "frac_layers_walked": round(dom["algorithmic_bytes_per_launch"] / dom["avg_launch_us_layers_walked"] / 1e3 / HBM_PEAK_GBS, 4),
            "share_of_token_time": dom["share_of_token_time"],
            "how": "the kernel class with the largest share of the decode-step time; duration = HIP events on the launch "
                   "stream around a hipGraph chain of that kernel (launch boundary included), measured in this run; "
                   "kernel_classes also lists avg_launch_us_layers_walked: the same chain walking the layers, i.e. with "
                   "weights / KV coming from the memory side as in the real step (lm_head has one matrix: identical)",
        },
        "kernel_classes": table,
        "roofline_lm_head": {"kernel": lm["class"], "achieved": lm["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": lm["frac_of_8TBps"], "avg_launch_us": lm["avg_launch_us"],
                             "algorithmic_bytes_per_launch": lm["algorithmic_bytes_per_launch"]},
        "step_roofline": {
            "bound": "hbm", "algorithmic_bytes_per_generation": int(step_bytes_total),
            "achieved": round(step_bytes_total * a.steps / dev_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(step_bytes_total * a.steps / dev_s / 1e9 / HBM_PEAK_GBS, 4),
            "us_per_token_device": round(1e6 * dev_s / (a.steps * ctx), 2),
            # two clocks that do not share code: torch events around the whole generations (above) against the sum of the per-class
            # chain timings of kernel_classes (the library's HIP events around hipGraph chains), priced per token with the embed launch
            "sum_of_kernel_class_chains_us": round(sum(r["avg_launch_us_layers_walked"] * r["launches_per_token"] for r in table) + prof_lo.get("embed", 0.0), 2),
            "per_kernel_class_us_eager_T_low": {k: round(v, 2) for k, v in prof_lo.items()},
            "per_kernel_class_us_eager_T_high": {k: round(v, 2) for k, v in prof_hi.items()},
        },
        "mfma_gemm_768x3072": gemm,
        "prefill": prefill,
        "sampled_generation": sampled,
        "other_configs": others,
        "device_time_s": round(dev_s, 4),
        "setup_s": round(setup_s, 2),
        "weight_broadcast_ms": None if bcast_ms is None else round(bcast_ms, 3),
        "weight_broadcast": bcast_note or "zg_gpt_broadcast_weights: one ncclBroadcast of the arena's weight region on the library's stream (device time, HIP events)",
        "first_tokens": [int(t) for t in ids[0, :8]],
        **scaling_extra,
    }
    if not a.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(cfg, weights, prompts[0], a.cpu_seconds)
    if "op_tier" in early:
        out["op_tier"] = early["op_tier"]
        if "first_tokens" in out["op_tier"]:  # same weights, same prompt: the literal drop-in and the model tier agree
            out["op_tier"]["tokens_equal_model_tier"] = out["op_tier"]["first_tokens"] == out["first_tokens"]
            cpu_v = out.get("cpu_baseline", {}).get("value")
            out["op_tier"]["vs_cpu_baseline"] = round(out["op_tier"]["value"] / cpu_v, 2) if cpu_v else None
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
