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


@pytest.mark.parametrize("tier", ["ops", "model"])
def test_cpp_host_reads_the_raw_weight_directory(tmp_path, tier):
    """load_linear / load_layer_norm / load_embedding (src/main.zig:210-269) in the C++ host: the reference's raw
    directory format written by weights_io.save_raw_dir (download_weights.py:57-64 layout); wrong sizes are errors."""
    from zig_gpt2_amd import weights_io

    cfg = synth.CONFIGS["tiny3"]
    w = synth.make_weights(cfg, seed=41, bf16=True)
    d = str(tmp_path / "raw")
    weights_io.save_raw_dir(d, cfg, w)
    prompt = synth.rand_tokens(411, 3, cfg.vocab_size)
    args = [BIN, "tiny3", d, ",".join(str(int(t)) for t in prompt), "24"] + (["--model-tier"] if tier == "model" else [])
    env = dict(os.environ, ZGPT2_STAGING_MB="64")
    out = subprocess.run(args, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr
    ids = np.array([int(t) for t in out.stdout.split()], dtype=np.uint64)
    assert np.array_equal(ids, oracle.GPT(cfg, w).generate_greedy(prompt, 24))
    # a truncated tensor file is refused (the reference's load_tensor would silently keep zeros)
    f = os.path.join(d, "model-h1-mlp-c_fc-w")
    with open(f, "r+b") as fh:
        fh.truncate(os.path.getsize(f) - 4)
    bad = subprocess.run(args, capture_output=True, text=True, timeout=300, env=env)
    assert bad.returncode != 0 and "size does not match" in bad.stderr



def test_cpp_host_op_tier_at_124m_matches_oracle_and_reports_its_speed():
    """The literal drop-in at full size: main.zig's dataflow over the op tier (host buffers, one FFI call per op, fp32
    weights mirrored once, the KV caches mirrored on the device) for GPT-2 124M x 64 positions — BASELINE configs[0]'s
    workload — token for token the oracle's; the program reports tokens/s on stderr."""
    import json

    cfg = synth.CONFIGS["124M"]
    prompt = synth.rand_tokens(1000, 1, cfg.vocab_size)
    args = [BIN, "124M", "0", ",".join(str(int(t)) for t in prompt), "64"]
    out = subprocess.run(args, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    ids = np.array([int(t) for t in out.stdout.split()], dtype=np.uint64)
    ref = oracle.GPT(cfg, synth.make_weights(cfg, seed=0, bf16=True)).generate_greedy(prompt, 64)
    assert np.array_equal(ids, ref), (ids, ref)
    line = [l for l in out.stderr.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["tier"] == "op" and d["steps"] == 64 and d["tokens_per_s"] > 0
    print(f"op tier, 124M x 64: {d['tokens_per_s']} tokens/s")
