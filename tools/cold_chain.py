#!/usr/bin/env python3
"""Per-launch chain cost of each decode kernel class with the SAME layer's weights every launch (they stay in the
XCDs' L2s: what bench.py's kernel_classes reports) against walking the layers (time_kernel(walk_layers=True): weights from the
memory side, as in the real step).  Random-init weights are not needed: the timing does not depend on values.
    python tools/cold_chain.py [124M|xl ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import torch
from zig_gpt2_amd import _lib, gpt, synth

lib = _lib.load(); _lib.check(lib.zg_init(0))
stream = torch.cuda.Stream(); _lib.check(lib.zg_set_stream(stream.cuda_stream))
for spec in (sys.argv[1:] or ["124M", "124M:8", "xl"]):
    name, _, b = spec.partition(":")
    m = gpt.GPT(synth.CONFIGS[name], batch=int(b or 1))
    for w in range(1, 6):
        row = {"model": name, "batch": int(b or 1), "class": w}
        for cyc in (0, 1, 0, 1):
            us, _ = m.time_kernel(w, 1024, walk_layers=bool(cyc))
            row.setdefault("cycle" if cyc else "same_layer", []).append(round(us, 3))
        print(json.dumps(row), flush=True)
    m.close()
