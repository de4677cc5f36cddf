import sys, os
sys.path.insert(0, os.getcwd())
import torch
from zig_gpt2_amd import _lib, gpt, synth
lib = _lib.load(); _lib.check(lib.zg_init(0))
cfg = synth.CONFIGS["124M"]
m = gpt.GPT(cfg)
gen = torch.Generator(device="cuda"); gen.manual_seed(3)
w = {}
for name, shape, mean, _ in synth.tensor_specs(cfg):
    w[name] = ((torch.randn(shape, generator=gen, device="cuda") * 0.02 + mean).to(torch.bfloat16).to(torch.float32)).contiguous()
m.load_weights(w)
for _ in range(2): m.generate_sample([[11]], 1024, 0.8, seed=1)
m.close()
