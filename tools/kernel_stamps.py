#!/usr/bin/env python3
"""In-kernel timeline of the decode GEMVs (needs `make -C zig_gpt2_amd/csrc stamps`,
ZGPT2_LIB=zig_gpt2_amd/lib/libzgpt2_hip_stamps.so).  Prints, per kernel class, the median
s_memtime deltas between the stamps of wave 0 of the first / last workgroup over a graph chain."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from zig_gpt2_amd import _lib, gpt, synth

lib = _lib.load(); _lib.check(lib.zg_init(0))
raw = C.CDLL(_lib.SO_PATH)
raw.zg_debug_stamps_read.argtypes = [C.c_void_p, C.c_size_t]
cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "124M"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
m = gpt.GPT(cfg, batch=B); m.load_weights(synth.make_weights(cfg, seed=0, bf16=True))
m.generate([synth.rand_tokens(b, 1, cfg.vocab_size) for b in range(B)], min(64, cfg.context_size))
names = ["t0 entry", "t1 W issued+T", "t2 LN stats barrier", "t3 xs ready", "t4 xr in regs", "t5 pass A done", "t6 pass B done", "t7 loop end", "t9 flush"]
for cls in (1, 3, 4, 5, 6):
    assert raw.zg_debug_stamps_begin() == 0
    m.time_kernel(cls, 64)
    buf = np.zeros(16 + 10 * 4096, np.uint64)
    assert raw.zg_debug_stamps_read(buf.ctypes.data, buf.size) == 0
    n = int(buf[0]); rec = buf[16:16 + 10 * n].reshape(n, 10).astype(np.int64)
    print(f"== class {cls} {gpt.GPT.PROFILE_CLASSES[cls]}: {n} records")
    for which, label in ((0, "first WG"), (None, "last WG")):
        r = rec[rec[:, 8] == 0] if which == 0 else rec[rec[:, 8] != 0]
        if len(r) == 0: continue
        r = r[np.argsort(r[:, 0])]
        ts = np.concatenate([r[:, :8], r[:, 9:10]], axis=1)
        d = np.diff(ts, axis=1)
        med = np.median(d[4:], axis=0)
        gap = np.median(np.diff(r[:, 0])[4:]) if len(r) > 6 else -1
        print(f"  {label}: launch-to-launch {gap:.0f} ticks; phases " + " ".join(f"{names[i+1].split()[0]}:{med[i]:.0f}" for i in range(8)) + f"  body {np.median(ts[4:, -1] - ts[4:, 0]):.0f}")
