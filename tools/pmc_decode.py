#!/usr/bin/env python3
"""The leanest target for rocprofv3 --pmc passes over the decode step: one full-context greedy generation of GPT-2
124M, one prompt, eager launches, no prefetcher (every kernel's own traffic), the stream drained every 16 steps
(rocprofv3 7.2's counter collection crashes behind long queues and over graph replays).  Nothing else runs."""
import os, sys
os.environ.setdefault("ZGPT2_SYNC_EVERY", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zig_gpt2_amd import _lib, gpt, synth
lib = _lib.load(); _lib.check(lib.zg_init(0))
stream = torch.cuda.Stream(); _lib.check(lib.zg_set_stream(stream.cuda_stream))
cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "124M"]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # usage: pmc_decode.py [model [prompts [f32|b24|f16]]]
kv = sys.argv[3] if len(sys.argv) > 3 else "f32"
m = gpt.GPT(cfg, batch=batch, use_graph=False, prefetch=False, prefill=False, kv_b24=kv == "b24", kv_f16=kv == "f16")
ids = m.generate([synth.rand_tokens(1000 + b, 1, cfg.vocab_size) for b in range(batch)], cfg.context_size)
print("generated", ids.shape, flush=True)
m.close()
