"""One rank of a multi-rank run of the library's native collective path on ONE device (tests/test_dist_two_ranks_gpu.py starts
several of these): zg_init(0), the communicator id from rank 0 through a file, zg_dist_init, rank 0 loads the weights, EVERY rank
calls zg_gpt_broadcast_weights (ranks > 0 receive the weight region and re-derive the folded LayerNorm vectors), generates its shard
of the prompts, all ranks all-gather their token matrices, every rank writes what it gathered.  The transport is tests/stub_rccl
(ZGPT2_RCCL_LIB): RCCL refuses two ranks on one GPU.
usage: dist_rank_worker.py <rank> <world> <workdir> <model> <weight_seed> <n_prompts> <n_steps>"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from zig_gpt2_amd import _lib, gpt, shard, synth

rank, world, work, name, seed, n_prompts, n_steps = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
lib = _lib.load()
_lib.check(lib.zg_init(0))
_lib.check(lib.zg_dist_available())
uid = (C.c_ubyte * 128)()
idf = os.path.join(work, "id.bin")
if rank == 0:
    _lib.check(lib.zg_dist_unique_id(uid, 128))
    with open(idf + ".tmp", "wb") as f:
        f.write(bytes(uid))
    os.rename(idf + ".tmp", idf)
else:
    for _ in range(60000):
        if os.path.exists(idf):
            break
        time.sleep(0.001)
    uid = (C.c_ubyte * 128).from_buffer_copy(open(idf, "rb").read())
_lib.check(lib.zg_dist_init(uid, 128, rank, world))
r, n = C.c_int(-1), C.c_int(-1)
_lib.check(lib.zg_dist_world(C.byref(r), C.byref(n)))
assert (r.value, n.value) == (rank, world)
cfg = synth.CONFIGS[name]
mine = shard.shard_prompts(n_prompts, world, rank)
assert len(mine) == n_prompts // world, "equal shards: the all-gather below takes equal-sized buffers"
m = gpt.GPT(cfg, batch=len(mine))
if rank == 0:  # load_gpt on one rank only (src/main.zig:304-314); the others hold zeros until the broadcast
    m.load_weights(synth.make_weights(cfg, seed=seed, bf16=True))
ms = C.c_float(-1.0)
_lib.check(lib.zg_gpt_broadcast_weights(m.h, 0, C.byref(ms)))
prompts = [synth.rand_tokens(700 + gi, 1 + gi % 4, cfg.vocab_size) for gi in mine]
ids = m.generate(prompts, n_steps).astype(np.int64)
send = torch.from_numpy(ids).cuda()
recv = torch.zeros((world,) + tuple(ids.shape), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
_lib.check(lib.zg_dist_allgather(send.data_ptr(), recv.data_ptr(), send.numel() * 8))
np.save(os.path.join(work, f"gathered.{rank}.npy"), recv.cpu().numpy().reshape(n_prompts, n_steps))
m.close()
_lib.check(lib.zg_dist_finalize())
print(f"rank {rank}: broadcast {ms.value:.3f} ms, {len(mine)} prompts done", flush=True)
