"""Time zg_gpt_prefill (whole-prompt forward) against the token-at-a-time prompt loop it replaces.

usage: python tools/bench_prefill.py [--model 124M] [--batch 1] [--lengths 128,512,1023]
Prints one JSON line per prompt length: ms per prefill, prompt tokens/s, speed-up over the decode loop.
"""
import argparse
import json
import time

import numpy as np

from zig_gpt2_amd import _lib, gpt as zgpt, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="124M")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--lengths", default="64,128,256,512,1023")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--planes", type=int, default=3, help="activation planes of the GEMMs: 3 (exact) or 2 (ZG_GPT_PREFILL_2PLANE)")
    ap.add_argument("--weights-f32", action="store_true")
    a = ap.parse_args()
    cfg = synth.CONFIGS[a.model]
    w = synth.make_weights(cfg, seed=0, bf16=not a.weights_f32)
    _lib.check(_lib.load().zg_init(0))
    m = zgpt.GPT(cfg, batch=a.batch, prefill_planes=a.planes, weights_f32=a.weights_f32)
    m.load_weights(w)
    for n in (int(x) for x in a.lengths.split(",")):
        n = min(n, cfg.context_size)
        toks = np.stack([synth.rand_tokens(900 + b, n, cfg.vocab_size) for b in range(a.batch)])
        m.prefill(toks, compute_logits=False)  # warm-up (also raises the LDS limit once)
        t0 = time.perf_counter()
        for _ in range(a.reps):
            m.prefill(toks, compute_logits=False)
        ms = (time.perf_counter() - t0) / a.reps * 1e3
        # the loop it replaces: n decode steps without lm_head (main.zig:331-334)
        loop_n = min(n, 128)
        m.forward(1, toks[:, 0], compute_logits=False)
        t0 = time.perf_counter()
        for s in range(loop_n):
            m.forward(s + 1, toks[:, s], compute_logits=False)
        loop_ms = (time.perf_counter() - t0) / loop_n * n * 1e3
        print(json.dumps({"model": a.model, "batch": a.batch, "planes": a.planes, "weights": "f32" if a.weights_f32 else "bf16",
                          "prompt_len": n, "prefill_ms": round(ms, 3),
                          "linear_tflops_useful": round(2.0 * a.batch * n * 12 * cfg.n_embed ** 2 * cfg.n_layer / ms / 1e9, 1),
                          "prompt_tokens_per_s": round(a.batch * n / ms * 1e3, 1),
                          "decode_loop_ms_est": round(loop_ms, 1), "speedup": round(loop_ms / ms, 1)}))
    m.close()


if __name__ == "__main__":
    main()
