"""N > 1 path on CPU: two gloo ranks shard a seeded read stream, build their dense counter tables
(with the test-only host emulation of the kernel -- there is no GPU here), and combine them with
the product's reduce helpers; rank 0 must hold exactly what one process computes over all reads."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_TOTAL = 3001  # odd on purpose: ranks get unequal shards


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emu_lib
    import parity
    import workloads
    from ngs_barcode_count_amd import distributed as bcdist

    w = workloads.make("config3", n_sets=(4, 40, 40, 40), lib=None)
    first, count = bcdist.shard(N_TOTAL, rank, world)
    seq, qual = w.synth.generate_host(first, count)
    # this rank's table and counters (host emulation of the kernel's lane code; TEST ONLY)
    import ngs_barcode_count_amd as pkg
    eplan = pkg.Plan(w.scheme, lib=emu_lib.lib())
    for i, s in enumerate(w.samples):
        eplan.add_sample(s, "sample_%d" % i)
    for b, refs in enumerate(w.counted):
        for i, s in enumerate(refs):
            eplan.add_counted(b, s, "bb%d_%d" % (b + 1, i))
    eplan.set_min_quality(20.0)
    outc, idx, entries, discard = emu_lib.emulate(eplan, seq, qual, None, 100, 100)
    table = torch.from_numpy(np.bincount(idx[outc == 0].astype(np.int64), minlength=entries).astype(np.int32))
    counters = {k: int((outc == i).sum()) for i, k in enumerate(pkg.COUNTER_NAMES)}
    counters["total_reads"] = count
    counters["unsupported_reads"] = 0
    # the product's N > 1 logic
    bcdist.reduce_table(table, dst=0)
    total = bcdist.reduce_counters(counters, torch.device("cpu"), dst=0)
    if rank == 0:
        nz = torch.nonzero(table).flatten().numpy()
        rows = parity.decode_rows(eplan, {int(i): int(table[i]) for i in nz}, False)
        allseq, allqual = w.synth.generate_host(0, N_TOTAL)
        o = workloads.oracle_for(w)
        o.process_batch(allseq, allqual, 100, 100)
        ok = rows == o.rows() and all(total[k] == v for k, v in o.counters.items()) and total["total_reads"] == N_TOTAL
        with open(out_path, "w") as f:
            f.write("ok" if ok else "MISMATCH %r vs %r" % (total, o.counters))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_and_reduce(tmp_path):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_shards_tile_the_read_range():
    from ngs_barcode_count_amd import distributed as bcdist
    for n in (0, 1, 7, 100, 3001):
        for world in (1, 2, 3, 8):
            spans = [bcdist.shard(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1


def _worker_random(rank, world, port, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emu_lib
    import parity
    import workloads
    import ngs_barcode_count_amd as pkg
    from ngs_barcode_count_amd import distributed as bcdist

    n_total = 4000
    w = workloads.make("config4", n_sets=(3, 6, 6, 6), n_molecules=1500)
    first, count = bcdist.shard(n_total, rank, world)
    seq, qual = w.synth.generate_host(first, count)
    eplan = pkg.Plan(w.scheme, lib=emu_lib.lib())
    for i, s in enumerate(w.samples):
        eplan.add_sample(s, "sample_%d" % i)
    for b, refs in enumerate(w.counted):
        for i, s in enumerate(refs):
            eplan.add_counted(b, s, "bb%d_%d" % (b + 1, i))
    outc, idx, entries, discard, rcode, rspace = emu_lib.emulate(eplan, seq, qual, None, 100, 100, with_random=True)
    out2, keys = parity.apply_set_semantics(outc, idx, rcode, rspace)  # this rank's local set semantics
    local = {k: int((out2 == i).sum()) for i, k in enumerate(pkg.COUNTER_NAMES)}
    local_keys = torch.from_numpy(np.unique(keys[out2 == 0]).astype(np.int64))
    # the product's exchange: every key gets one owner; duplicates across ranks meet there
    recv = bcdist.exchange_keys(local_keys)
    owned = torch.unique(recv)  # what bc_engine_import_keys does on the device
    assert bool((bcdist.key_owner(owned, world) == rank).all())
    fixed = dict(local)
    fixed["duplicates"] = local["duplicates"] + local["matched"] - owned.numel()
    fixed["matched"] = owned.numel()
    fixed["total_reads"], fixed["unsupported_reads"] = count, 0
    total = bcdist.reduce_counters(fixed, torch.device("cpu"), dst=0)
    table = torch.from_numpy(np.bincount((owned.numpy().astype(np.uint64) // np.uint64(rspace)).astype(np.int64),
                                         minlength=entries).astype(np.int32))
    bcdist.reduce_table(table, dst=0)
    if rank == 0:
        nz = torch.nonzero(table).flatten().numpy()
        rows = parity.decode_rows(eplan, {int(i): int(table[i]) for i in nz}, False)
        allseq, allqual = w.synth.generate_host(0, n_total)
        o = workloads.oracle_for(w)
        o.process_batch(allseq, allqual, 100, 100)
        ok = rows == o.rows() and all(total[k] == v for k, v in o.counters.items())
        with open(out_path, "w") as f:
            f.write("ok" if ok and o.counters["duplicates"] > 500 else "MISMATCH %r vs %r" % (total, o.counters))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_random_barcode_key_exchange(tmp_path):
    """PCR-duplicate collapse across ranks: a molecule sequenced on both ranks must count once"""
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker_random, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def _worker_counts(rank, world, port, out_path):
    """the merge of raw-key count maps: (key, count) pairs gathered on the root and added"""
    import numpy as np
    import torch
    import torch.distributed as dist
    import ngs_barcode_count_amd.distributed as bcdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    # each rank saw some keys of a common universe, some of them on both ranks
    universe = np.arange(1, 5001, dtype=np.int64) * 7919
    mine = rng.choice(universe, size=3000, replace=False)
    cnts = rng.integers(1, 50, size=mine.size).astype(np.int32)
    for dst in (None, 0):
        rk, rc = bcdist.exchange_keys(torch.from_numpy(mine), torch.from_numpy(cnts), dst=dst)
        # what bc_engine_import_counts does on the device: add per key
        got = {}
        for k, c in zip(rk.tolist(), rc.tolist()):
            got[k] = got.get(k, 0) + c
        if dst is None:
            assert all(int(bcdist.key_owner(torch.tensor([k]), world)[0]) == rank for k in list(got)[:200])
        # everyone contributes its local truth; rank 0 checks the union
        gathered = [None] * world
        dist.all_gather_object(gathered, (mine.tolist(), cnts.tolist(), got))
        if rank == 0:
            truth = {}
            for m, c, _ in gathered:
                for k, v in zip(m, c):
                    truth[k] = truth.get(k, 0) + v
            merged = {}
            for _, _, g in gathered:
                for k, v in g.items():
                    assert k not in merged  # every key has exactly one holder after the exchange
                    merged[k] = v
            ok = merged == truth and (dst is None or len(gathered[1][2]) == 0)
            with open(out_path, "a") as f:
                f.write("ok;" if ok else "MISMATCH;")
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_count_map_merge(tmp_path):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker_counts, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok;ok;"


def _worker_table(rank, world, port, out_path):
    """the all-to-all sum of dense tables (distributed.reduce_table) against torch.distributed.reduce"""
    import numpy as np
    import torch
    import torch.distributed as dist
    import ngs_barcode_count_amd.distributed as bcdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    for n in (24, 7, 2400, 4099, 24000):  # multiples of 4 x world take the packed all-to-all path, the others the fallback
        for dst in range(world):
            rng = np.random.default_rng(1000 * rank + n)
            # mostly small counts (one byte on the wire), some large, some with the top bit set (u32 viewed as int32)
            mine = torch.from_numpy(np.where(rng.random(n) < 0.9, rng.integers(0, 200, size=n),
                                             rng.integers(-2**31, 2**31 - 1, size=n)).astype(np.int32))
            a, b = mine.clone(), mine.clone()
            bcdist.reduce_table(a, dst=dst, method="alltoall")
            bcdist.reduce_table(b, dst=dst, method="reduce")
            if rank == dst:
                ok = ok and bool((a == b).all())  # wraps identically (u32 sums)
    res = [None] * world
    dist.all_gather_object(res, ok)
    if rank == 0:
        with open(out_path, "w") as f:
            f.write("ok" if all(res) else "MISMATCH")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_table_all_to_all_sum(tmp_path, world):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker_table, args=(world, _free_port(), out), nprocs=world, join=True)
    assert open(out).read() == "ok"
