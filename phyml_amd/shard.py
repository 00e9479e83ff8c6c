"""Pattern ranges of the multi-GPU path: contiguous shards whose sizes differ by at most one -- the same
split libphyhip.so makes inside a sharded instance (phyhip_shard.hpp: create_group), used by bench.py when every rank of a
one-process-per-GPU run builds its own shard.  The collective itself lives in the library (RCCL), not here."""
from __future__ import annotations


def shard_range(n_pattern: int, rank: int, world: int):
    """[lo, hi) of rank's contiguous pattern range; sizes differ by at most one."""
    base, rem = divmod(int(n_pattern), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
