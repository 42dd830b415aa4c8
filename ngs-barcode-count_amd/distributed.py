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


def reduce_table(table, dst=0, method=None):
    """Sum of every rank's dense u32 counter table (viewed as int32: same bits) onto rank dst, in place
    there.  On xGMI every GPU has its own link to every other, so the default is not a ring:
      1. all-to-all (equal slices): rank r receives slice r of every table, 7 peers at once, one link each,
      2. each rank adds up the slices it received (local, HBM speed),
      3. the summed slices are sent to dst point to point, again one link each.
    A ring reduce would push the whole 16 GB table of the DEL workloads through one link's bandwidth.
    Tables whose length is not a multiple of the world size, and method="reduce" /
    BC_TABLE_REDUCE=reduce, use torch.distributed.reduce."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return table
    import os
    world, rank = dist.get_world_size(), dist.get_rank()
    method = method or os.environ.get("BC_TABLE_REDUCE", "alltoall")
    n = table.numel()
    if method == "reduce" or n % world or n == 0:
        dist.reduce(table, dst=dst, op=dist.ReduceOp.SUM)
        return table
    mine = n // world
    recv = torch.empty(n, dtype=table.dtype, device=table.device)
    dist.all_to_all_single(recv, table)
    # one pass over what arrived; 32-bit accumulation wraps like the u32 counters these are
    part = torch.sum(recv.view(world, mine), dim=0, dtype=table.dtype)
    if rank == dst:
        table[dst * mine:(dst + 1) * mine].copy_(part)
        ops = [dist.P2POp(dist.irecv, table[r * mine:(r + 1) * mine], r) for r in range(world) if r != dst]
    else:
        ops = [dist.P2POp(dist.isend, part, dst)]
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    return table


def reduce_counters(counters, device, dst=0):
    """sum of the outcome counters (dict as returned by Engine.counters()) onto rank dst"""
    t = torch.tensor([counters[k] for k in COUNTER_NAMES], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM)
    return dict(zip(COUNTER_NAMES, [int(x) for x in t.tolist()]))


# ---- random-barcode mode: counts are set sizes, so tables must not simply be summed -------------
def key_owner(keys, world):
    """owner rank of every 64-bit key (any well-mixed function works; it only has to be the same
    on every rank)"""
    h = keys * -7046029254386353131  # 0x9E3779B97F4A7C15 as int64; wraps
    return ((h >> 33) & 0x7FFFFFFF) % world


def exchange_keys(keys, counts=None, dst=None):
    """all-to-all so that every key ends up on exactly one rank (duplicates across ranks meet on
    their owner).  keys: int64 tensor of this rank's distinct keys -> int64 tensor received.
    counts: optional int32 tensor travelling with the keys -> (keys, counts) received.
    dst: send everything to that rank instead of spreading the keys by hash (the merge of raw-key
    results, which the root needs whole to write them out)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return keys if counts is None else (keys, counts)
    owner = key_owner(keys, world) if dst is None else torch.full_like(keys, dst)
    order = torch.argsort(owner)
    keys = keys[order].contiguous()
    send = torch.bincount(owner, minlength=world).to(torch.int64)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send)
    n_recv = int(recv.sum().item())
    out = torch.empty(n_recv, dtype=torch.int64, device=keys.device)
    dist.all_to_all_single(out, keys, recv.tolist(), send.tolist())
    if counts is None:
        return out
    counts = counts[order].contiguous()
    out_c = torch.empty(n_recv, dtype=counts.dtype, device=counts.device)
    dist.all_to_all_single(out_c, counts, recv.tolist(), send.tolist())
    return out, out_c


def finish_random(engine, device, dst=0, gather=False):
    """Global PCR-duplicate collapse: export this rank's keys, exchange them, keep
    only the owned ones, fix the matched / duplicate counters, and leave per-tuple distinct counts
    in the engine's table for reduce_table() + bc_engine_finish on the root.  Returns the global
    counters on rank dst.
    gather=True (raw-key plans, whose results are a key map and not a table that could be summed):
    every key goes to rank dst, whose engine then holds the whole set."""
    local = engine.counters()
    n = engine.key_count()
    keys = torch.empty(max(n, 1), dtype=torch.int64, device=device)
    engine.export_keys(keys.data_ptr(), n)
    recv = exchange_keys(keys[:n], dst=dst if gather else None)
    engine.clear_keys()
    owned = engine.import_keys(recv.data_ptr(), recv.numel()) if recv.numel() else 0
    fixed = dict(local)
    # every locally matched read is either the one owner-side survivor of its key or a duplicate
    fixed["duplicates"] = local["duplicates"] + local["matched"] - owned
    fixed["matched"] = owned
    # (summed over ranks: matched = distinct keys overall, duplicates = all other passing reads)
    return reduce_counters(fixed, device, dst=dst)


def finish_sparse(engine, device, dst=0):
    """Raw-key plans (no sample / counted-barcode conversion file): counts live in a per-rank key map.
    With a random barcode the (tuple, random) keys of every rank are gathered on rank dst (set union);
    without one the (key, count) pairs are, and added up there.  Afterwards rank dst's engine holds the
    job's result (bc_engine_finish / result rows); returns the global counters on rank dst."""
    if engine.plan.random_barcode:
        return finish_random(engine, device, dst=dst, gather=True)
    n = engine.export_counts(None, None, 0)
    keys = torch.empty(max(n, 1), dtype=torch.int64, device=device)
    cnts = torch.empty(max(n, 1), dtype=torch.int32, device=device)
    engine.export_counts(keys.data_ptr(), cnts.data_ptr(), n)
    rk, rc = exchange_keys(keys[:n], cnts[:n], dst=dst)
    engine.clear_keys()
    if rk.numel():
        engine.import_counts(rk.data_ptr(), rc.data_ptr(), rk.numel())
    return reduce_counters(engine.counters(), device, dst=dst)
