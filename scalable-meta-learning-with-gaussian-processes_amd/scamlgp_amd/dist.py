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


# ---- model-level sharding: every rank holds a contiguous shard of the source tasks ----------------------------------
class TaskShard:
    """Which slice [lo, hi) of the T_global source tasks this rank's SourceGPStack holds."""

    def __init__(self, n_tasks: int, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.n_tasks = int(n_tasks)
        self.lo, self.hi = shard_range(self.n_tasks, self.world, self.rank)

    @property
    def local(self) -> slice:
        return slice(self.lo, self.hi)


def gather_task_axis(local: torch.Tensor, shard: Optional[TaskShard], dim: int = -1) -> torch.Tensor:
    """Per-task quantities of all ranks side by side along ``dim``: the local block goes into its slice of a zero
    buffer over all T_global tasks and one all-reduce adds the ranks' buffers (shards may differ in size by one,
    which all_gather would not take)."""
    if shard is None or shard.world == 1:
        return local
    shape = list(local.shape)
    shape[dim] = shard.n_tasks
    buf = torch.zeros(shape, dtype=local.dtype, device=local.device)
    idx = [slice(None)] * local.dim()
    idx[dim] = shard.local
    buf[tuple(idx)] = local
    allreduce_sum_(buf, shard.group)
    return buf


def fused_allreduce(parts: Sequence[Optional[torch.Tensor]], shard: Optional[TaskShard]) -> list:
    """Sum several tensors over the ranks with ONE collective: they travel as one flat buffer
    (posterior step [mu_s || Sigma_s || var_s], fit step [sum MLL || sum dMLL/dtheta])."""
    if shard is None or shard.world == 1:
        return list(parts)
    live = [p for p in parts if p is not None]
    flat = torch.cat([p.reshape(-1) for p in live])
    allreduce_sum_(flat, shard.group)
    out, off = [], 0
    for p in parts:
        if p is None:
            out.append(None)
            continue
        out.append(flat[off:off + p.numel()].reshape(p.shape))
        off += p.numel()
    return out


def reduce_mll_and_grad(mll: torch.Tensor, grad: torch.Tensor, shard: Optional[TaskShard] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Summed marginal likelihood and summed hyper-gradient over ALL tasks of all ranks (BASELINE configs[3]):
    local sums over the task axis, then one fused all-reduce of [sum_t MLL_t || sum_t dMLL_t/dtheta (P)]."""
    s_mll, s_grad = fused_allreduce([mll.sum(0, keepdim=True), grad.sum(0)], shard)
    return s_mll.squeeze(0), s_grad


def standardize_fit_sharded(y_local: torch.Tensor, extra: torch.Tensor, shard: Optional[TaskShard]) -> Tuple[torch.Tensor, torch.Tensor]:
    """botorch Standardize(m=1) statistics over cat(all ranks' y_local, extra) without gathering the data: mean from
    an all-reduced (sum, count), then the unbiased std from an all-reduced sum of squared deviations.  ``extra`` (the
    target observations) is replicated on every rank and counted once."""
    if shard is None or shard.world == 1:
        raise ValueError("single-process stacks use model.standardize_fit")
    one = y_local.new_ones(())
    s = torch.stack([y_local.sum(), one * y_local.numel()])
    allreduce_sum_(s, shard.group)
    total, count = s[0] + extra.sum(), s[1] + extra.numel()
    mean = total / count
    ss = ((y_local - mean) ** 2).sum().reshape(1)
    allreduce_sum_(ss, shard.group)
    ss = ss[0] + ((extra - mean) ** 2).sum()
    std = torch.sqrt(ss / (count - 1.0)) if float(count) >= 2 else one.clone()
    std = torch.where(std >= 1e-8, std, torch.ones_like(std))
    return mean, std
