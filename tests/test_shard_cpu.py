"""world_size-2 gloo test (CPU) of the multi-GPU plumbing: prompt partition, weight-arena broadcast,
token gather — the N>1 path of bench.py without a GPU."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from zig_gpt2_amd import shard, synth


def test_shard_prompts_partition_is_exact():
    for n in (1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            parts = [shard.shard_prompts(n, w, r) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1 << 16
    ref = torch.from_numpy(synth.to_bf16_bits(synth.fill_normal(5, n, 0, 0.02)).view(np.uint8))  # the arena is bytes
    arena = ref.clone() if rank == 0 else torch.zeros(2 * n, dtype=torch.uint8)  # rank 0 "uploaded" the weights
    shard.broadcast_weights(arena, dist, src=0)
    ok = bool(torch.equal(arena, ref))
    mine = shard.shard_prompts(5, world, rank)
    toks = np.array([[100 * i + s for s in range(4)] for i in mine], dtype=np.uint64).reshape(len(mine), 4)
    allt = shard.gather_tokens(toks, dist)
    q.put((rank, ok, allt.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_gather_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expect = [[100 * i + s for s in range(4)] for i in range(5)]
    for rank, ok, allt in res:
        assert ok, f"rank {rank}: broadcast arena differs"
        assert allt == expect
