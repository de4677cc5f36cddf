#!/usr/bin/env python3
"""Edge and nonsense arguments through the MODEL tier of the C ABI: configurations that cannot be built, sequence lengths 0 and
beyond the context, token ids beyond the vocabulary, wrong token counts, NULL arrays, zero steps, prompts longer than the run —
every call must RETURN a status (no crash, no hang), and a good call on the same handle must still match the oracle afterwards.
python tests/sweeps/errors_gpt.py [seed] [count]"""
import ctypes as C, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "oracle"))
import numpy as np
import oracle
from golden_io import assert_model_close
from zig_gpt2_amd import _lib, gpt as zgpt, synth

zg = _lib.load(); _lib.check(zg.zg_init(0))
seed, count = (int(v) for v in (sys.argv[1:3] + ["0", "300"][len(sys.argv) - 1:]))
rng = np.random.default_rng(seed)
codes = {}
def note(r):
    codes[r] = codes.get(r, 0) + 1
    if r != 0: assert len(zg.zg_last_error()) > 0
# ---- configurations
for _ in range(40):
    cfg = _lib.GptConfig(int(rng.choice([0, 1, 65, 257])), int(rng.choice([0, 1, 48, 64])), int(rng.choice([0, 1, 2])),
                         int(rng.choice([0, 1, 2, 3, 5])), int(rng.choice([0, 64, 100, 128, 192, 320, 4096])))
    h = C.c_void_p()
    r = zg.zg_gpt_create(C.byref(h), C.byref(cfg), int(rng.choice([0, 1, 2, 8, 9, 100])), int(rng.choice([0, 1, 2, 4, 8, 66, 1 << 20])))
    note(r)
    if r == 0: zg.zg_gpt_destroy(h)
note(zg.zg_gpt_create(None, None, 1, 0))
# ---- calls on a good handle
cfg = synth.CONFIGS["tiny"]
w = synth.make_weights(cfg, seed=1, bf16=True)
B = 3
m = zgpt.GPT(cfg, batch=B)
m.load_weights(w)
V, ctx = cfg.vocab_size, cfg.context_size
tok = np.zeros(64 * B, np.uint64); lg = np.zeros(B * V, np.float32); big = np.full(3 * cfg.n_embed * cfg.n_embed + V * cfg.n_embed, np.nan, np.float32)  # (NaN weights must not fault either)
out = np.zeros(B * ctx, np.uint64); lens = np.ones(B, np.uint64)
p = lambda a: a.ctypes.data
def tokens():
    t = tok.copy()
    if rng.integers(0, 4) == 0: t[int(rng.integers(0, B))] = int(rng.choice([V, V + 5, 1 << 40]))
    return t
for it in range(count):
    k = int(rng.integers(0, 7))
    t = tokens()
    if os.environ.get("TRACE"): print("call", it, "kind", k, flush=True)
    sl = int(rng.choice([0, 1, 2, ctx, ctx + 1, 1 << 33]))
    nt = int(rng.choice([0, 1, B - 1, B, B + 1]))
    if k == 0:
        args = (sl, p(t) if rng.integers(0, 8) else None, nt, int(rng.integers(0, 2)), p(lg) if rng.integers(0, 3) else None, int(rng.choice([0, V, B * V])))
        if os.environ.get("TRACE"): print("  forward", args[0], args[1] is not None, args[2:4], args[4] is not None, args[5], t[:B], flush=True)
        r = zg.zg_gpt_forward(m.h, *args)
    elif k == 1:
        args = (p(t) if rng.integers(0, 8) else None, int(rng.choice([0, 1, 20, 64])), int(rng.choice([0, 1, 20, ctx, ctx + 1])), int(rng.integers(0, 2)),
                p(lg) if rng.integers(0, 3) else None, int(rng.choice([0, V, B * V])))
        if os.environ.get("TRACE"): print("  prefill", args[0] is not None, args[1:4], args[4] is not None, args[5], t[:B], flush=True)
        r = zg.zg_gpt_prefill(m.h, *args)
    elif k == 2:
        lens[:] = [int(rng.choice([0, 1, 5, 21, ctx, ctx + 1])) for _ in range(B)]
        args = (p(t) if rng.integers(0, 8) else None, int(rng.choice([0, 1, 21, 64])), p(lens) if rng.integers(0, 8) else None,
                int(rng.choice([0, 1, 7, ctx, ctx + 1])), p(out) if rng.integers(0, 8) else None, int(rng.choice([0, B, B * ctx])))
        if os.environ.get("TRACE"): print("  generate", args[0] is not None, args[1], args[2] is not None, args[3], args[4] is not None, args[5], lens, t[:B], flush=True)
        r = zg.zg_gpt_generate_greedy(m.h, *args)
    elif k == 3: r = zg.zg_gpt_argmax(m.h, p(out) if rng.integers(0, 4) else None, nt)
    elif k == 4: r = zg.zg_gpt_sample(m.h, sl, p(t), nt, float(rng.choice([0.0, -1.0, 1.0, float("nan")])), None, 7, p(out), None, 0)
    elif k == 5: r = zg.zg_gpt_load_tensor(m.h, int(rng.choice([-1, 0, 1, 3, 4, 99])), p(big) if rng.integers(0, 4) else None, int(rng.choice([0, 7, V * cfg.n_embed])))
    else: r = zg.zg_gpt_load_block_tensor(m.h, int(rng.choice([0, 1, 2, 99])), int(rng.choice([-1, 0, 2, 11, 12, 99])), p(big) if rng.integers(0, 4) else None, int(rng.choice([0, 7, 3 * cfg.n_embed * cfg.n_embed])))
    note(r)
# ---- (the load calls above may have replaced tensors with zeros: reload) the handle still computes the oracle's numbers
m.load_weights(w)
toks = synth.rand_tokens(5, 20, V)
ref = oracle.GPT(cfg, w).forced_logits(toks, 18)
got = m.prefill([toks[:19]] * B)
assert_model_close(ref[0], got[1], "after the error sweep")
m.close()
print(f"{count + 41} calls returned; status histogram {dict(sorted(codes.items()))}; the handle still matches the oracle")
