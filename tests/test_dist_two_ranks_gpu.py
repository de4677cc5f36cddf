"""The library's native multi-rank path (zg_dist_* + zg_gpt_broadcast_weights, csrc/dist.hip) with MORE THAN ONE
RANK on the one-GPU box: every rank is a process on device 0 and the six collective entry points are bound from
tests/stub_rccl (ZGPT2_RCCL_LIB; bytes travel through files) because RCCL refuses two ranks on one device.  What this runs for
real: the id handed from rank 0 to the others, zg_dist_init as a rendezvous of N processes, a RECEIVING rank of the weight
broadcast (its folded LayerNorm vectors re-derived from what arrived), sharded generation, the all-gather of the token matrices
in rank order, finalize — and the C++ host's --gpus N plumbing (fork before any GPU call, id and results over pipes).  RCCL itself
runs with one rank in tests/test_dist_gpu.py; a real multi-GPU node is the driver's."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from zig_gpt2_amd import shard, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB_SRC = os.path.join(ROOT, "tests", "stub_rccl", "stub_rccl.cpp")
STUB_SO = os.path.join(ROOT, "tests", "stub_rccl", "libstub_rccl.so")
BIN = os.path.join(ROOT, "zig_gpt2_amd", "bin", "zgpt2_main")


@pytest.fixture(scope="module")
def stub():
    if not os.path.exists(STUB_SO) or os.path.getmtime(STUB_SO) < os.path.getmtime(STUB_SRC):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", STUB_SO, STUB_SRC])
    return STUB_SO


@pytest.mark.parametrize("world", [2, 4])
def test_native_path_with_several_ranks_on_one_gpu(stub, tmp_path, world):
    name, seed, n_prompts, n_steps = "tiny", 57, 4, 24
    env = dict(os.environ, ZGPT2_RCCL_LIB=stub)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_rank_worker.py"), str(r), str(world), str(tmp_path), name, str(seed),
                               str(n_prompts), str(n_steps)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}:\n{se[-2000:]}"
    cfg = synth.CONFIGS[name]
    w = synth.make_weights(cfg, seed=seed, bf16=True)
    want = np.stack([oracle.GPT(cfg, w).generate_greedy(synth.rand_tokens(700 + gi, 1 + gi % 4, cfg.vocab_size), n_steps) for gi in range(n_prompts)])
    for r in range(world):  # every rank gathered every rank's rows, in rank order = prompt order (contiguous shards)
        got = np.load(tmp_path / f"gathered.{r}.npy")
        assert np.array_equal(got, want.astype(np.int64)), f"rank {r}"
    assert [g for r in range(world) for g in shard.shard_prompts(n_prompts, world, r)] == list(range(n_prompts))


def test_cpp_host_gpus_mode_three_ranks_on_one_gpu(stub):
    """zgpt2_main --gpus 3 with every rank on device 0: three forked ranks, the id over pipes, rank 0 loads, ranks 1 and 2 receive
    the weights, five prompts in shards of 2 / 2 / 1, rows back over pipes in prompt order."""
    cfg = synth.CONFIGS["tiny3"]
    seed, n_steps = 23, 28
    prompts = [synth.rand_tokens(230 + i, 1 + i % 3, cfg.vocab_size) for i in range(5)]
    arg = ";".join(",".join(str(int(t)) for t in p) for p in prompts)
    env = dict(os.environ, ZGPT2_RCCL_LIB=stub, ZGPT2_ALL_RANKS_ON_DEVICE="0")
    out = subprocess.run([BIN, "tiny3", str(seed), arg, str(n_steps), "--gpus", "3"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr
    assert "weights broadcast to 3 rank(s)" in out.stderr
    rows = [np.array([int(t) for t in line.split()], dtype=np.uint64) for line in out.stdout.strip().splitlines()]
    assert len(rows) == 5
    w = synth.make_weights(cfg, seed=seed, bf16=True)
    for i, p in enumerate(prompts):
        assert np.array_equal(rows[i], oracle.GPT(cfg, w).generate_greedy(p, n_steps)), i
