"""Multi-GPU plumbing for the match/count path: reads shard trivially (no collective on the data
path); at the end ONE sum-reduce of the dense counter tables and of the outcome counters
(torch.distributed backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU
tests)."""
import torch
import torch.distributed as dist

from ._lib import COUNTER_NAMES


def shard(n_total, rank, world):
    """contiguous read range of `rank`: (first, count); ranges tile [0, n_total) exactly"""
    base, extra = divmod(n_total, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def reduce_table(table, dst=0):
    """in-place sum of the dense u32 counter table (viewed as int32: same bits) onto rank dst"""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(table, dst=dst, op=dist.ReduceOp.SUM)
    return table


def reduce_counters(counters, device, dst=0):
    """sum of the outcome counters (dict as returned by Engine.counters()) onto rank dst"""
    t = torch.tensor([counters[k] for k in COUNTER_NAMES], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM)
    return dict(zip(COUNTER_NAMES, [int(x) for x in t.tolist()]))
