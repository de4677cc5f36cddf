#!/usr/bin/env python3
"""Where one wave of a prompt GEMM (prefill.hip) spends its K loop: cycles of workgroup 8 / wave 0 of the LAST prompt GEMM a
whole-prompt pass of a two-Block 124M-shaped model launches (ln_1 + c_attn of the last Block: 18 column tiles, K = 768), from the
stamps of the diagnostic build:  make -C zig_gpt2_amd/csrc stamps [PF_ABL=1|2|3]  then
ZGPT2_LIB=$PWD/zig_gpt2_amd/lib/libzgpt2_hip_stamps.so python tools/pf_stamps.py [prompts [tokens]].
(Each stamp is an s_memtime behind an lgkmcnt(0) wait: only the coarse ones are kept, finer ones disturbed what they measured.)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from zig_gpt2_amd import _lib, gpt, synth
lib = _lib.load(); _lib.check(lib.zg_init(0))
raw = C.CDLL(os.environ["ZGPT2_LIB"])
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1023
base = synth.CONFIGS["124M"]
cfg = synth.GPTConfig(base.vocab_size, base.context_size, 2, base.n_heads, base.n_embed)
m = gpt.GPT(cfg, batch=batch)
m.load_weights(synth.make_weights(cfg, seed=0, bf16=True))
toks = np.stack([synth.rand_tokens(900 + b, n, cfg.vocab_size) for b in range(batch)])
for _ in range(3):
    m.prefill(toks, compute_logits=False)
out = (C.c_ulonglong * 32)()
assert raw.zg_debug_prefill_stamps(out, 32) == 0
ks, wait, bar, loop, epi, e, pro = [int(out[i]) for i in range(7)]
print(f"EPI {e}: entry to loop {pro} cycles; {ks} K-steps, K loop {loop} cycles = {loop / max(ks, 1):.0f} per K-step (48 MFMAs = 1536), of which vmcnt(0) waits {wait} "
      f"({wait / max(ks, 1):.0f} per step), barrier {bar} ({bar / max(ks, 1):.0f} per step); barrier + epilogue {epi}")
m.close()
