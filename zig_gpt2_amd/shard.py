"""Multi-GPU plumbing of the batched-prompt case: prompts are independent units, so
they are block-partitioned over the ranks (one process per GPU) and the only collective is a single
broadcast of the weight arena from rank 0 at start-up (RCCL over xGMI on GPUs; any
torch.distributed backend works, which is how the CPU tests drive it with gloo)."""
import numpy as np


def shard_prompts(n_prompts, world_size, rank):
    """Global prompt indices owned by `rank`: contiguous blocks, sizes differing by at most one."""
    base, extra = divmod(n_prompts, world_size)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def broadcast_weights(arena, dist, src=0):
    """Broadcast the (already uploaded on `src`) weight arena tensor in place to every rank."""
    if dist is None or dist.get_world_size() == 1:
        return arena
    dist.broadcast(arena, src=src)
    return arena


def gather_tokens(local_tokens, dist):
    """All ranks' [n_local, n_steps] token matrices concatenated in rank order (host-side result
    collection, outside any timed region).  local_tokens: numpy uint64."""
    if dist is None or dist.get_world_size() == 1:
        return local_tokens
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, np.asarray(local_tokens))
    return np.concatenate([o for o in out if len(o)], axis=0)
