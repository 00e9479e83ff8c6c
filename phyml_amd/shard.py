"""Pattern sharding for the multi-GPU path: contiguous pattern ranges, one process per GPU,
ONE all-reduce of the per-shard log-likelihood per evaluation.  No other collective exists on this path."""
from __future__ import annotations


def shard_range(n_pattern: int, rank: int, world: int):
    """[lo, hi) of rank's contiguous pattern range; sizes differ by at most one."""
    base, rem = divmod(int(n_pattern), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_sum(tensor, dist):
    """Sum a small tensor (lnL, or lnL and dlnL) over all shards.  `dist` is torch.distributed (backend
    nccl == RCCL on the GPUs, gloo in the CPU tests) or None for a single process."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(tensor)
    return tensor
