"""Native multi-GPU path of the C ABI (include/zgpt2.h zg_dist_*, zg_gpt_broadcast_weights): one process per GPU,
one RCCL broadcast of the weight arena, prompts sharded.  The GPU box has ONE GPU, so what runs here is the one-rank world —
communicator set-up through librccl (dlopen), the broadcast call itself, the receiver-side state handling — and the C++ host's
--gpus mode end to end; more ranks have no hardware here (the partition arithmetic is tested on CPU: test_dist_cpu.py)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle
from zig_gpt2_amd import _lib
from zig_gpt2_amd import gpt as zgpt
from zig_gpt2_amd import synth

pytestmark = pytest.mark.gpu
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zig_gpt2_amd", "bin", "zgpt2_main")


def test_broadcast_weights_one_rank_world_then_generate(zg):
    cfg = synth.CONFIGS["tiny"]
    w = synth.make_weights(cfg, seed=91, bf16=True)
    uid = (C.c_ubyte * 128)()
    _lib.check(zg.zg_dist_unique_id(uid, 128))
    assert any(uid), "ncclGetUniqueId returned an empty id"
    _lib.check(zg.zg_dist_init(uid, 128, 0, 1))
    try:
        r, n = C.c_int(-1), C.c_int(-1)
        _lib.check(zg.zg_dist_world(C.byref(r), C.byref(n)))
        assert (r.value, n.value) == (0, 1)
        m = zgpt.GPT(cfg, batch=2)
        m.load_weights(w)
        ms = C.c_float(-1.0)
        _lib.check(zg.zg_gpt_broadcast_weights(m.h, 0, C.byref(ms)))
        assert ms.value >= 0.0
        with pytest.raises(_lib.ZgError):  # a root outside the world is an error, not a hang
            _lib.check(zg.zg_gpt_broadcast_weights(m.h, 3, None))
        prompts = [synth.rand_tokens(911, 3, cfg.vocab_size), synth.rand_tokens(912, 5, cfg.vocab_size)]
        ids = m.generate(prompts, 40)
        for b, p in enumerate(prompts):
            assert np.array_equal(ids[b], oracle.GPT(cfg, w).generate_greedy(p, 40)), b
        m.close()
    finally:
        _lib.check(zg.zg_dist_finalize())
    with pytest.raises(_lib.ZgError):  # no communicator any more
        m2 = zgpt.GPT(cfg)
        try:
            _lib.check(zg.zg_gpt_broadcast_weights(m2.h, 0, None))
        finally:
            m2.close()


def test_cpp_host_gpus_mode_one_rank():
    """zgpt2_main --gpus 1: the rank is a child process started before any GPU call; it makes the id, initialises RCCL, loads
    the weights, broadcasts, generates its three prompts in lock step; the parent prints the rows in prompt order."""
    cfg = synth.CONFIGS["tiny3"]
    seed, n_steps = 13, 30
    prompts = [synth.rand_tokens(130 + i, 2 + i, cfg.vocab_size) for i in range(3)]
    arg = ";".join(",".join(str(int(t)) for t in p) for p in prompts)
    out = subprocess.run([BIN, "tiny3", str(seed), arg, str(n_steps), "--gpus", "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    assert "weights broadcast to 1 rank(s)" in out.stderr
    rows = [np.array([int(t) for t in line.split()], dtype=np.uint64) for line in out.stdout.strip().splitlines()]
    assert len(rows) == 3
    w = synth.make_weights(cfg, seed=seed, bf16=True)
    for i, p in enumerate(prompts):
        assert np.array_equal(rows[i], oracle.GPT(cfg, w).generate_greedy(p, n_steps)), i
