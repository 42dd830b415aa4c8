"""The job's end on the GPU: bc_engine_finish in bounded memory (it used to need 12 bytes per row on the
device in one piece), the streaming form, and the two-level counting's fold paths behind tables somebody else reads
(caller-owned tables of awkward lengths / alignments, and an engine-owned table whose pointer was handed out
before the first submit).  Everything against the CPU oracle."""
import numpy as np
import pytest

import workloads

pytestmark = pytest.mark.gpu


def _submit(w, eng, first, n, chunk=1 << 20):
    import torch
    rl = w.read_len
    done = 0
    while done < n:
        m = min(chunk, n - done)
        dseq = torch.empty(m * rl, dtype=torch.uint8, device="cuda")
        dqual = torch.empty(m * rl, dtype=torch.uint8, device="cuda")
        w.synth.generate_device(0, None, first + done, m, dseq.data_ptr(), dqual.data_ptr())
        torch.cuda.synchronize()
        eng.submit_device(dseq.data_ptr(), dqual.data_ptr(), m, rl, rl)
        eng.sync()
        done += m


def _oracle_dense(w, first, n):
    """-> (oracle counters, {dense table index: count}) of reads [first, first + n)"""
    seq, qual = w.synth.generate_host(first, n)
    o = workloads.oracle_for(w)
    o.process_batch(seq, qual, w.read_len, w.read_len)
    s_idx = {s: i for i, s in enumerate(w.samples)}
    b_idx = [{s: i for i, s in enumerate(refs)} for refs in w.counted]
    dense = {}
    for sample, tup, cnt in o.rows():
        di = s_idx[sample]
        for b, part in enumerate(tup.split(",")):
            di = di * len(w.counted[b]) + b_idx[b][part]
        dense[di] = cnt
    return o.counters, dense


def _as_tensor(ptr, n):
    """torch view of n int32 at a raw device pointer (the engine's own table)"""
    import torch

    class Raw:
        __cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (int(ptr), False), "version": 2}

    return torch.as_tensor(Raw(), device="cuda")


def _table_dict(t):
    import torch
    nz = torch.nonzero(t).flatten()
    return dict(zip(nz.tolist(), t[nz].tolist()))


def test_finish_needs_less_device_memory_than_its_rows(monkeypatch):
    """3 M reads over 108 M table entries -> ~2.6 M rows = ~31 MB as (u64 index, u32 count); the device is filled with
    ballast until less than that is free, and finish still returns every row, equal to the oracle's: the table is
    compacted range by range through two small staging buffers"""
    import torch
    import ngs_barcode_count_amd as pkg
    monkeypatch.setenv("BC_FINISH_CHUNK_ROWS", str(1 << 17))  # 2 x 1.5 MB of staging
    w = workloads.make("config3", n_sets=(4, 300, 300, 300))
    n = 3_000_000
    eng = pkg.Engine(w.plan, device=0)
    _submit(w, eng, 0, n)
    exp_counters, exp = _oracle_dense(w, 0, n)
    need = 12 * len(exp)
    torch.cuda.empty_cache()
    ballast = []
    target = need // 2  # leave half of what the rows would need in one piece
    for piece in (1 << 30, 1 << 26, 1 << 22):
        while True:
            free_b, _ = torch.cuda.mem_get_info()
            if free_b - piece < target:
                break
            try:
                ballast.append(torch.empty(piece, dtype=torch.uint8, device="cuda"))
            except RuntimeError:
                break
    free_b, _ = torch.cuda.mem_get_info()
    assert free_b < need, (free_b, need)
    rows = eng.finish()
    assert rows == len(exp)
    s, b, c = eng.rows()
    got = {}
    sizes = [len(x) for x in w.counted]
    for i in range(rows):
        di = int(s[i])
        for k in range(3):
            di = di * sizes[k] + int(b[i, k])
        got[di] = int(c[i])
    assert got == exp
    got_counters = eng.counters()
    assert {k: got_counters[k] for k in exp_counters} == exp_counters
    # the streaming form hands over the same rows and keeps nothing
    chunks = []
    assert eng.finish_stream(lambda k, v: chunks.append((k, v))) == rows
    assert len(chunks) > 4  # really in pieces
    keys = np.concatenate([k for k, _ in chunks])
    vals = np.concatenate([v for _, v in chunks])
    assert dict(zip(keys.tolist(), vals.tolist())) == exp
    assert eng.nonzero_entries() == rows
    del ballast
    eng.close()


def test_finish_reports_out_of_memory_and_stays_usable(monkeypatch):
    """no room for the staging buffers: BC_ERR_NOMEM (not a HIP error, nothing leaked), and once memory is there
    again the same engine finishes"""
    import torch
    import ngs_barcode_count_amd as pkg
    from ngs_barcode_count_amd import _lib
    monkeypatch.setenv("BC_FINISH_CHUNK_ROWS", str(1 << 26))  # 2 x 768 MB of staging wanted (capped by the rows)
    w = workloads.make("config3", n_sets=(4, 300, 300, 300))
    n = 4_000_000
    eng = pkg.Engine(w.plan, device=0)
    _submit(w, eng, 0, n)
    expect_rows = eng.nonzero_entries()  # ~3.5 M rows: one staging slot of ~42 MB
    torch.cuda.empty_cache()
    ballast = []
    for piece in (1 << 30, 1 << 26, 1 << 22, 1 << 21):  # until the device really has nothing left (mem_get_info is
        while True:                                      # only an estimate of what the next hipMalloc can get)
            try:
                ballast.append(torch.empty(piece, dtype=torch.uint8, device="cuda"))
            except RuntimeError:
                break
    free_before, _ = torch.cuda.mem_get_info()
    with pytest.raises(pkg.BarcodeCountError) as err:
        eng.finish()
    assert err.value.code == _lib.BC_ERR_NOMEM, err.value
    free_after, _ = torch.cuda.mem_get_info()
    assert free_after >= free_before - (2 << 20)  # the failed call gave back what it had taken
    del ballast
    torch.cuda.empty_cache()
    assert eng.finish() == expect_rows
    eng.close()


@pytest.mark.parametrize("shape", ["dense_fold_tail", "dense_fold", "sparse_fold", "misaligned"])
def test_caller_owned_table_holds_plain_counts_after_every_sync(monkeypatch, shape):
    """two-level counting behind a caller-owned table: after every bc_engine_sync the caller's memory holds the whole
    counts (that is what a cross-GPU reduce reads).  dense_fold*: many reads per sync over a small table -> the
    streaming fold kernel (945 entries: not a multiple of 4, its partial last group); sparse_fold: few reads per sync
    over a large table -> the per-bit kernel; misaligned: a table that does not start on 16 bytes"""
    import torch
    import ngs_barcode_count_amd as pkg
    monkeypatch.setenv("BC_BITMAP_MIN_ENTRIES", "1")
    n_sets, n, off = {"dense_fold_tail": ((3, 5, 7, 9), 60_000, 0), "dense_fold": ((4, 8, 8, 8), 60_000, 0),
                      "sparse_fold": ((4, 200, 200, 200), 60_000, 0), "misaligned": ((4, 8, 8, 8), 60_000, 1)}[shape]
    w = workloads.make("config3", n_sets=n_sets)
    entries = w.plan.table_entries
    backing = torch.zeros(entries + 4, dtype=torch.int32, device="cuda")
    table = backing[off:off + entries]
    torch.cuda.synchronize()
    eng = pkg.Engine(w.plan, device=0, table_ptr=table.data_ptr())
    cuts = [0, n // 3, n // 3 + 7_000, n]
    for a, b in zip(cuts, cuts[1:]):
        _submit(w, eng, a, b - a, chunk=1 << 20)  # one submit, then sync: the fold runs here
        exp_counters, exp = _oracle_dense(w, 0, b)
        assert _table_dict(table) == exp, (shape, b)
        assert int(backing[:off].sum()) == 0 and int(backing[off + entries:].sum()) == 0  # nothing beside the table
        got = eng.counters()
        assert {k: got[k] for k in exp_counters} == exp_counters
    assert eng.result_rows() == workloads_rows(w, n)
    eng.close()


def workloads_rows(w, n):
    seq, qual = w.synth.generate_host(0, n)
    o = workloads.oracle_for(w)
    o.process_batch(seq, qual, w.read_len, w.read_len)
    return o.rows()


def test_engine_owned_table_pointer_taken_before_the_first_submit(monkeypatch):
    """a caller that fetches bc_engine_table_ptr once at setup and reads (or reduces) that memory after
    submit + sync must see the whole counts, not the counts minus their first occurrence"""
    import ngs_barcode_count_amd as pkg
    monkeypatch.setenv("BC_BITMAP_MIN_ENTRIES", "1")
    w = workloads.make("config3", n_sets=(4, 30, 30, 30))
    eng = pkg.Engine(w.plan, device=0)
    view = _as_tensor(eng.table_ptr, w.plan.table_entries)  # pointer handed out while the bit map is still clean
    n = 50_000
    _submit(w, eng, 0, n // 2)
    _, exp = _oracle_dense(w, 0, n // 2)
    assert _table_dict(view) == exp
    _submit(w, eng, n // 2, n - n // 2)
    exp_counters, exp = _oracle_dense(w, 0, n)
    assert _table_dict(view) == exp
    assert eng.result_rows() == workloads_rows(w, n)
    eng.close()


@pytest.mark.parametrize("n_sets,n", [((4, 6, 6, 6), 40_000), ((4, 60, 60, 60), 60_000), ((4, 200, 200, 200), 50_000)])
def test_reset_by_dirty_blocks_leaves_nothing_behind(monkeypatch, n_sets, n):
    """bc_engine_reset of an engine-owned table with two-level counting zeroes only the 256-byte blocks somebody added
    to (repeat adds, the hot-counter cache's flush, the straight-to-the-table tiles of a wave that meets mostly repeats):
    job after job into one engine, every job's rows equal the oracle's for THAT job alone -- tables from tiny (every add
    a repeat) to sparse (hardly any), and once more after the pointer was handed out (then the whole table is zeroed)"""
    import ngs_barcode_count_amd as pkg
    monkeypatch.setenv("BC_BITMAP_MIN_ENTRIES", "1")
    w = workloads.make("config3", n_sets=n_sets)
    eng = pkg.Engine(w.plan, device=0)
    first = 0
    for job in range(4):
        m = n if job != 2 else n // 3
        _submit(w, eng, first, m, chunk=(m // 3) & ~3)
        exp_counters, exp = _oracle_dense(w, first, m)
        got = eng.counters()
        assert {k: got[k] for k in exp_counters} == exp_counters, job
        assert eng.nonzero_entries() == len(exp), job
        if job == 3:
            view = _as_tensor(eng.table_ptr, w.plan.table_entries)  # exposes the table: folded, plain counts
            assert _table_dict(view) == exp
        else:
            s, b, c = eng.rows()
            sizes = [len(x) for x in w.counted]
            got_rows = {}
            for i in range(len(c)):
                di = int(s[i])
                for k in range(3):
                    di = di * sizes[k] + int(b[i, k])
                got_rows[di] = int(c[i])
            assert got_rows == exp, job
        first += m
        eng.reset()
        assert eng.nonzero_entries() == 0, job
    eng.close()


def test_reset_results_keeps_the_outcome_counters():
    """bc_engine_reset_results: a fresh Results for the next sample, SequenceErrors go on (what bench.py does between
    its steps); bc_engine_reset clears both"""
    import ngs_barcode_count_amd as pkg
    w = workloads.make("config3", n_sets=(4, 40, 40, 40))
    eng = pkg.Engine(w.plan, device=0)
    n = 30_000
    _submit(w, eng, 0, n)
    c1, _ = _oracle_dense(w, 0, n)
    eng.reset_results()
    assert eng.nonzero_entries() == 0
    got = eng.counters()
    assert {k: got[k] for k in c1} == c1 and got["total_reads"] == n
    _submit(w, eng, n, n)
    c2, exp2 = _oracle_dense(w, n, n)
    got = eng.counters()
    assert {k: got[k] for k in c1} == {k: c1[k] + c2[k] for k in c1}
    assert eng.nonzero_entries() == len(exp2)  # the second sample's rows only
    eng.reset()
    assert sum(eng.counters().values()) == 0 and eng.result_rows() == []
    eng.close()
