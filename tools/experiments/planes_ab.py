#!/usr/bin/env python3
"""A/B of the activation planes between the kernels of a Block (lock-step batch, bf16 weights: GemvArgs.pl_in / pl_out,
ZGPT2_NO_PLANES=1 switches them off): the same greedy generation both ways — ids, last-step logits, wall time per step
and the per-class launch times (hipGraph chains).
    python tools/planes_ab.py [124M:8 124M:2 xl:8 ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from zig_gpt2_amd import _lib, gpt, synth

lib = _lib.load(); _lib.check(lib.zg_init(0))
stream = torch.cuda.Stream(); _lib.check(lib.zg_set_stream(stream.cuda_stream))

for spec in (sys.argv[1:] or ["124M:8", "124M:3"]):
    name, _, b = spec.partition(":")
    B = int(b or 8)
    cfg = synth.CONFIGS[name]
    rng = np.random.default_rng(5)
    w = {}
    for tname, shape, mean, _ in synth.tensor_specs(cfg):
        w[tname] = synth.round_bf16((rng.standard_normal(int(np.prod(shape)), dtype=np.float32) * np.float32(0.02) + np.float32(mean))).reshape(shape)
    prompts = [synth.rand_tokens(900 + i, 1 + i % 3, cfg.vocab_size) for i in range(B)]
    n = min(cfg.context_size, int(os.environ.get("AB_STEPS", cfg.context_size)))
    ids, logits, res, cls = {}, {}, {}, {}
    for on in (False, True, False, True):
        if on: os.environ.pop("ZGPT2_NO_PLANES", None)
        else: os.environ["ZGPT2_NO_PLANES"] = "1"
        m = gpt.GPT(cfg, batch=B, kv_f16=bool(os.environ.get("AB_KV_F16")))
        m.load_weights(w)
        m.generate(prompts, min(64, n))
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            out = m.generate(prompts, n)
            best = min(best, time.perf_counter() - t0)
        lg = m.forward(5, [int(p[0]) for p in prompts])
        cls[on] = {gpt.GPT.PROFILE_CLASSES[c]: round(m.time_kernel(c, 200)[0], 2) for c in (1, 2, 3, 4, 5, 6)}
        m.close()
        ids.setdefault(on, out); logits.setdefault(on, np.array(lg))
        assert np.array_equal(ids[on], out)
        res.setdefault(on, []).append(round(best / (n - 1) * 1e6, 2))
    print(json.dumps({"model": name, "batch": B, "steps": n, "us_per_step_off": res[False], "us_per_step_on": res[True],
                      "identical_ids": bool(np.array_equal(ids[False], ids[True])),
                      "max_logit_diff": float(np.abs(logits[False] - logits[True]).max()), "logit_scale": float(np.abs(logits[False]).max()),
                      "class_us_off": cls[False], "class_us_on": cls[True]}), flush=True)
