import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from zig_gpt2_amd import _lib, gpt, synth
lib = _lib.load(); _lib.check(lib.zg_init(0))
stream = torch.cuda.Stream(); _lib.check(lib.zg_set_stream(stream.cuda_stream))
cfg = synth.CONFIGS[sys.argv[1]]
w = synth.make_weights(cfg, seed=3, bf16=True)
os.environ["ZGPT2_DUAL"] = "0"
m0 = gpt.GPT(cfg, batch=1); m0.load_weights(w)
ref = [m0.forward(t, [7]) for t in (1, 2, 3)]
os.environ["ZGPT2_DUAL"] = "1"
m = gpt.GPT(cfg, batch=1); m.load_weights(w)
for i, t in enumerate((1, 2, 3, 1, 2, 3)):
    t0 = time.perf_counter()
    try:
        lg = m.forward(t, [7])
        print(i, "ok", f"{1e3*(time.perf_counter()-t0):.2f} ms", "max diff", float(np.abs(lg - ref[i % 3]).max()), flush=True)
    except Exception as e:
        print(i, "FAIL", f"{1e3*(time.perf_counter()-t0):.2f} ms", str(e)[:80], flush=True)
