import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
from zig_gpt2_amd import _lib, gpt, synth
lib = _lib.load(); _lib.check(lib.zg_init(0))
for name, batch in (("124M", 1), ("124M", 8), ("xl", 1)):
    cfg = synth.CONFIGS[name]
    m = gpt.GPT(cfg, batch=batch)
    row = {"model": name, "batch": batch}
    for T in (1, 64, 128, 256, 257, 512, 768, 1024):
        us, _ = m.time_kernel(2, 256, walk_layers=True, at=T)
        row[f"T{T}"] = round(us, 2)
    for which, nm in ((1, "c_attn"), (3, "c_proj"), (4, "c_fc"), (5, "mlp_proj")):
        us, _ = m.time_kernel(which, 256, walk_layers=True)
        row[nm] = round(us, 2)
    print(json.dumps(row), flush=True)
    m.close()
