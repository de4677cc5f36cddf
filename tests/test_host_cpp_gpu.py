"""The compiled-language caller of the drop-in boundary: zig_gpt2_amd/host/zgpt2_main.cpp is
src/main.zig restated in C++ over include/zgpt2_ops.hpp (the mirror of ops.zig).  Its greedy
tokens must equal the CPU oracle's, through the op tier (one FFI call per op, host buffers, like
main.zig) and through the model tier (one FFI call per generation)."""
import os
import subprocess

import numpy as np
import pytest

import oracle
from zig_gpt2_amd import synth

pytestmark = pytest.mark.gpu
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zig_gpt2_amd", "bin", "zgpt2_main")


@pytest.mark.parametrize("name,seed,n_steps", [("tiny", 7, 40), ("tiny3", 8, 48)])
@pytest.mark.parametrize("tier", ["op", "model"])
def test_cpp_host_generation_matches_oracle(name, seed, n_steps, tier):
    cfg = synth.CONFIGS[name]
    prompt = synth.rand_tokens(seed + 100, 3, cfg.vocab_size)
    args = [BIN, name, str(seed), ",".join(str(int(t)) for t in prompt), str(n_steps)]
    if tier == "model":
        args.append("--model-tier")
    env = dict(os.environ, ZGPT2_STAGING_MB="64")
    out = subprocess.run(args, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr
    ids = np.array([int(t) for t in out.stdout.split()], dtype=np.uint64)
    ref = oracle.GPT(cfg, synth.make_weights(cfg, seed=seed, bf16=True)).generate_greedy(prompt, n_steps)
    assert np.array_equal(ids, ref), (ids, ref)
