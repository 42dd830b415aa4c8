"""Multi-GPU plumbing for the match/count path: reads shard trivially (no collective on the data
path); at the end ONE sum-reduce of the dense counter tables and of the outcome counters
(torch.distributed backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU
tests).

Stream contract: an Engine works on its own HIP stream, torch.distributed on torch's.  Every engine call used
here returns only after its stream has drained (export_*, counters, materialize_table, sync), so what the engine
produced is safe to hand to torch; in the other direction _torch_done() drains torch's stream before buffers a
collective has written are handed to the engine."""
import torch
import torch.distributed as dist

from ._lib import COUNTER_NAMES


def _torch_done(t):
    """collectives are enqueued on torch's stream without blocking the host: wait for them before the engine (its own
    stream) reads what they wrote"""
    if t.is_cuda:
        torch.cuda.current_stream(t.device).synchronize()


def shard(n_total, rank, world):
    """contiguous read range of `rank`: (first, count); ranges tile [0, n_total) exactly"""
    base, extra = divmod(n_total, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def pack_table(table):
    """dense int32 (u32 bits) counts -> (uint8 tensor, overflow indices int64, overflow values int32):
    byte[i] = count where it fits a byte, else 0 and (i, count) in the overflow list.  On a GPU tensor this is
    the engine's bc_table_pack_u8 kernel; on a CPU tensor (the gloo tests) the same thing in torch."""
    n = table.numel()
    if table.is_cuda:
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        out = torch.empty(n, dtype=torch.uint8, device=table.device)
        cap = max(1024, n // 4096)
        stream = torch.cuda.current_stream(table.device).cuda_stream
        while True:
            idx = torch.empty(cap, dtype=torch.int64, device=table.device)
            val = torch.empty(cap, dtype=torch.int32, device=table.device)
            need = C.c_uint64()
            rc = lib.bc_table_pack_u8(table.data_ptr(), n, out.data_ptr(), idx.data_ptr(), val.data_ptr(), cap,
                                      C.byref(need), table.device.index or 0, stream)
            if rc != 0:
                raise RuntimeError(_lib.last_error(lib))
            if need.value <= cap:
                return out, idx[:need.value], val[:need.value]
            cap = need.value
    big = (table < 0) | (table > 255)
    idx = torch.nonzero(big).flatten()
    return torch.where(big, torch.zeros_like(table), table).to(torch.uint8), idx, table[idx]


def sum_slices(rows, world, mine, dtype):
    """rows: uint8 tensor of world slices of `mine` bytes -> their element-wise sum as `dtype`"""
    if rows.is_cuda and mine % 4 == 0:
        from . import _lib
        lib = _lib.load()
        out = torch.empty(mine, dtype=dtype, device=rows.device)
        rc = lib.bc_table_sum_u8(rows.data_ptr(), world, mine, out.data_ptr(), rows.device.index or 0,
                                 torch.cuda.current_stream(rows.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(_lib.last_error(lib))
        return out
    return torch.sum(rows.view(world, mine), dim=0, dtype=dtype)


def reduce_table(table, dst=0, method=None):
    """Sum of every rank's dense u32 counter table (viewed as int32: same bits) onto rank dst, in place
    there.  On xGMI every GPU has its own link to every other, so the default is not a ring, and the
    counts travel as bytes (pack_table: exact, the rare count above 255 goes in a side list):
      1. all-to-all (equal slices): rank r receives slice r of every packed table, 7 peers at once,
      2. each rank adds up the slices it received, plus the overflow entries sent to it (local, HBM speed),
      3. the summed slices are packed again and sent to dst point to point, one link each.
    A ring reduce of the raw table would push 16 GB (DEL workloads) through one link's bandwidth.
    Tables whose length is not a multiple of 4 x the world size, and method="reduce" /
    BC_TABLE_REDUCE=reduce, use torch.distributed.reduce.  A failure of the packed exchange is raised, never
    papered over: a rank that fell back on its own would leave its peers inside a different collective."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return table
    import os
    world, rank = dist.get_world_size(), dist.get_rank()
    method = method or os.environ.get("BC_TABLE_REDUCE", "alltoall")
    n = table.numel()
    if method == "reduce" or n % (4 * world) or n == 0:
        dist.reduce(table, dst=dst, op=dist.ReduceOp.SUM)
        return table
    return _reduce_table_packed(table, dst, world, rank, n)


def _reduce_table_packed(table, dst, world, rank, n):
    mine = n // world
    dev = table.device
    # 1. packed slices to their owners (+ the overflow entries, addressed by slice)
    small, oidx, oval = pack_table(table)
    recv = torch.empty(n, dtype=torch.uint8, device=dev)
    dist.all_to_all_single(recv, small)
    del small
    ridx, rval = _exchange_by_owner(oidx, oval, oidx // mine, world)
    # 2. one pass over what arrived
    part = sum_slices(recv, world, mine, table.dtype)
    del recv
    if ridx.numel():
        part.index_add_(0, ridx - rank * mine, rval)
    # 3. summed slices to dst, packed again
    small, oidx, oval = pack_table(part)
    if rank == dst:
        got = torch.empty(n, dtype=torch.uint8, device=dev)
        got[dst * mine:(dst + 1) * mine].copy_(small)
        ops = [dist.P2POp(dist.irecv, got[r * mine:(r + 1) * mine], r) for r in range(world) if r != dst]
    else:
        ops = [dist.P2POp(dist.isend, small, dst)]
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    gidx, gval = _exchange_by_owner(oidx + rank * mine, oval, torch.full_like(oidx, dst), world)
    if rank == dst:
        table.copy_(got)  # uint8 -> int32
        if gidx.numel():
            table.index_add_(0, gidx, gval)  # the bytes there are 0
    return table


def _exchange_by_owner(idx, val, owner, world):
    """(int64 idx, int32 val) pairs -> the rank named by `owner` (all-to-all with uneven splits)"""
    any_pairs = torch.tensor([idx.numel()], dtype=torch.int64, device=idx.device)
    dist.all_reduce(any_pairs, op=dist.ReduceOp.MAX)
    if int(any_pairs.item()) == 0:  # the usual case: every count fitted its byte, nothing to exchange
        return idx[:0], val[:0]
    order = torch.argsort(owner)
    idx, val = idx[order].contiguous(), val[order].contiguous()
    send = torch.bincount(owner, minlength=world).to(torch.int64)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send)
    n_recv = int(recv.sum().item())
    out_i = torch.empty(n_recv, dtype=torch.int64, device=idx.device)
    out_v = torch.empty(n_recv, dtype=val.dtype, device=val.device)
    dist.all_to_all_single(out_i, idx, recv.tolist(), send.tolist())
    dist.all_to_all_single(out_v, val, recv.tolist(), send.tolist())
    return out_i, out_v


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


def finish_random(engine, device, dst=0, gather=False, table=None):
    """Global PCR-duplicate collapse: export this rank's keys, exchange them, keep
    only the owned ones and fix the matched / duplicate counters.  Returns the global counters on rank dst.
    Dense plans (table = the engine's counter table as a tensor): every rank then writes the per-tuple distinct
    counts of the keys it owns into its table (bc_engine_materialize_table) and the tables are summed onto rank
    dst, whose bc_engine_finish / Engine.rows() compacts that sum.
    gather=True (raw-key plans, whose results are a key map and not a table that could be summed):
    every key goes to rank dst, whose engine then holds the whole set."""
    local = engine.counters()
    n = engine.key_count()
    keys = torch.empty(max(n, 1), dtype=torch.int64, device=device)
    engine.export_keys(keys.data_ptr(), n)
    recv = exchange_keys(keys[:n], dst=dst if gather else None)
    _torch_done(recv)
    engine.clear_keys()
    owned = engine.import_keys(recv.data_ptr(), recv.numel()) if recv.numel() else 0
    if not gather:
        if table is None:
            raise ValueError("finish_random: a dense plan needs table= (the tensor behind the engine's counter table)")
        engine.materialize_table()  # returns when the engine's stream has drained
        reduce_table(table, dst=dst)
        _torch_done(table)
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
    _torch_done(rk)
    engine.clear_keys()
    if rk.numel():
        engine.import_counts(rk.data_ptr(), rc.data_ptr(), rk.numel())
    return reduce_counters(engine.counters(), device, dst=dst)
