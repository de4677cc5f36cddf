#!/usr/bin/env python3
"""XCD (HW_REG_XCC_ID) of block 0 of every launch of the last decode step, with the prefetcher running and without
(needs the diagnostic build: make -C zig_gpt2_amd/csrc stamps; ZGPT2_LIB=zig_gpt2_amd/lib/libzgpt2_hip_stamps.so)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zig_gpt2_amd import _lib, gpt, synth
lib = _lib.load(); _lib.check(lib.zg_init(0))
stream = torch.cuda.Stream(); _lib.check(lib.zg_set_stream(stream.cuda_stream))
cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "124M"]
for mode in ("0", "2"):
    os.environ["ZGPT2_PF_MODE"] = mode
    m = gpt.GPT(cfg, batch=1)
    for n in (100, 300, 1024):
        m.generate([synth.rand_tokens(1, 1, cfg.vocab_size)], n)
        st = m.prefetch_stats()
        print(json.dumps({"pf_mode": mode, "steps": n, "xcd_of_block0": st["xcd_of_block0"]}), flush=True)
    m.close()
