"""Create / use / destroy handles in a loop: device memory must come back (hipMemGetInfo), host RSS must not grow without bound."""
import os, sys, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from zig_gpt2_amd import _lib, gpt as zgpt, synth
zg = _lib.load(); _lib.check(zg.zg_init(0))
cfgs = [synth.CONFIGS[n] for n in ("tiny", "nano-char", "medium-slice")]
ws = [synth.make_weights(c, seed=1, bf16=True) for c in cfgs]
def free(): return torch.cuda.mem_get_info()[0]
rows = []
for it in range(240):
    c, w = cfgs[it % 3], ws[it % 3]
    m = zgpt.GPT(c, batch=1 + it % 8, weights_f32=bool(it % 5 == 0), kv_b24=bool(it % 7 == 0))
    if it % 5 != 0: m.load_weights(w)
    else: m.load_weights(synth.make_weights(c, seed=1, bf16=False))
    m.generate([[1, 2, 3]] * (1 + it % 8), min(c.context_size, 40))
    m.prefill([[1, 2, 3, 4, 5]] * (1 + it % 8))
    m.close()
    if it % 40 == 39:
        torch.cuda.synchronize()
        rows.append((it + 1, free() >> 20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10))
        print("after", *rows[-1], "MiB free / MiB max RSS", flush=True)
assert abs(rows[-1][1] - rows[1][1]) < 64, rows
assert rows[-1][2] - rows[1][2] < 200, rows
print("no leak")
