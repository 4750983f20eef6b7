"""Task sharding across ranks (one process per GPU) and the path's only exchange step.

Source tasks are independent (scamlgp/model.py:176-188); the single coupling is a sum over
tasks — the summed marginal likelihood, and the weighted target prior sum_i w_i mu_i,
sum_i w_i^2 Sigma_i (scamlgp/model.py:129-134).  Ranks own contiguous task shards; one
all-reduce (RCCL over xGMI on GPUs, gloo in the CPU tests) combines the per-shard sums."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_tasks: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of the tasks rank `rank` owns; sizes differ by at most one."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, rem = divmod(n_tasks, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_ranges_balanced(n_points: Sequence[int], world_size: int) -> list:
    """Contiguous shards of a ragged stack balanced by the O(n^3) factorisation cost."""
    cost = [max(int(n), 1) ** 3 for n in n_points]
    total = sum(cost)
    bounds, acc, lo = [], 0, 0
    for r in range(world_size):
        target = total * (r + 1) / world_size
        hi = lo
        while hi < len(cost) and (acc + cost[hi] <= target or hi == lo) and len(cost) - hi > world_size - r - 1:
            acc += cost[hi]
            hi += 1
        if r == world_size - 1:
            hi = len(cost)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def allreduce_sum_(buf: torch.Tensor, group: Optional[dist.ProcessGroup] = None, async_op: bool = False):
    """In-place sum over ranks of a fused buffer (e.g. [sum MLL | mu_s | Sigma_s]); no-op without
    an initialised process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return None
