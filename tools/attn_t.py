import os, sys
sys.path.insert(0, os.getcwd())
from zig_gpt2_amd import _lib, gpt, synth
import torch
lib = _lib.load(); _lib.check(lib.zg_init(0))
stream = torch.cuda.Stream(); _lib.check(lib.zg_set_stream(stream.cuda_stream))
for name, B in (("124M", 1), ("124M", 8), ("xl", 1)):
    cfg = synth.CONFIGS[name]
    m = gpt.GPT(cfg, batch=B)
    for T in (64, 256, 257, 512, 768, 1024):
        us, _ = m.time_kernel(2, 256, at=T)
        print(name, B, "T", T, "attention us", round(us, 2), flush=True)
    m.close()
