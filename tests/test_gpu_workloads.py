"""BASELINE.json configs on the GPU: oracle parity on a seeded sample, and size-independent
properties at full size (one outcome per read, table sum = matched, shard additivity, determinism)."""
import os

import numpy as np
import pytest

import workloads

pytestmark = pytest.mark.gpu


def _run(w, first, n, eng=None, chunk=1 << 22):
    import torch
    import ngs_barcode_count_amd as pkg
    eng = eng or pkg.Engine(w.plan, device=0)
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
    return eng


@pytest.mark.parametrize("name,n", [("config2", 200000), ("config3", 200000), ("config5", 20000)])
def test_config_sample_vs_oracle(name, n):
    """full-size reference sets (3 x 1000 / 100 k guides), seeded reads, first n reads vs the oracle"""
    w = workloads.make(name)
    eng = _run(w, 0, n)
    seq, qual = w.synth.generate_host(0, n)
    o = workloads.oracle_for(w)
    o.process_batch(seq, qual, w.read_len, w.read_len)
    got = eng.counters()
    for k, v in o.counters.items():
        assert got[k] == v, (k, got, o.counters)
    assert eng.result_rows() == o.rows()
    eng.close()


@pytest.mark.parametrize("n_sets,n", [((4, 30, 30, 30), 120_000), ((4, 200, 200, 200), 60_000)])
def test_two_level_counting_vs_oracle(monkeypatch, n_sets, n):
    """the first-occurrence bit map in front of the counter table (large dense tables; forced here for small ones): many
    repeats per tuple and few, several submits with a fold between them (the streaming and the per-bit fold kernel),
    every row against the oracle"""
    import ngs_barcode_count_amd as pkg
    monkeypatch.setenv("BC_BITMAP_MIN_ENTRIES", "1")
    w = workloads.make("config3", n_sets=n_sets)
    eng = pkg.Engine(w.plan, device=0)
    _run(w, 0, n // 2, eng=eng, chunk=n // 6)
    mid = eng.counters()  # syncs: folds the bits set so far into the table
    assert mid["total_reads"] == n // 2
    _run(w, n // 2, n - n // 2, eng=eng, chunk=n // 5)
    seq, qual = w.synth.generate_host(0, n)
    o = workloads.oracle_for(w)
    o.process_batch(seq, qual, w.read_len, w.read_len)
    got = eng.counters()
    assert {k: got[k] for k in o.counters} == o.counters
    assert eng.result_rows() == o.rows()
    eng.close()


def test_config4_random_barcode_vs_oracle():
    """config 4 shape: DEL + 12-nt random barcode, molecules drawn with repeats (mean 2 per molecule)"""
    n = 150000
    w = workloads.make("config4", n_molecules=n // 2)
    eng = _run(w, 0, n, chunk=40000)  # several submits: the hash set grows across batches
    seq, qual = w.synth.generate_host(0, n)
    o = workloads.oracle_for(w)
    o.process_batch(seq, qual, w.read_len, w.read_len)
    got = eng.counters()
    for k, v in o.counters.items():
        assert got[k] == v, (k, got, o.counters)
    assert got["duplicates"] > n // 4
    assert eng.result_rows() == o.rows()
    eng.close()


def _oracle_outcomes_parallel(w, seq, qual, threads=12):
    """per-read outcome codes and the merged rows of `threads` independent oracle contexts, one slice each (the
    oracle's linear fix_error over 100 k guides takes ~2 ms per corrected read: a minute on one core)"""
    import threading
    from collections import Counter
    R = w.read_len
    n = seq.size // R
    cuts = [n * i // threads for i in range(threads + 1)]
    ctxs = [workloads.oracle_for(w) for _ in range(threads)]
    outs = [None] * threads

    def work(i):
        a, b = cuts[i], cuts[i + 1]
        outs[i] = ctxs[i].process_batch_outcomes(seq[a * R:b * R], qual[a * R:b * R] if qual is not None else None, R, R)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    rows = Counter()
    counters = Counter()
    for c in ctxs:
        for s_, t_, k in c.rows():
            rows[(s_, t_)] += k
        counters.update(c.counters)
    return np.concatenate(outs), sorted((s_, t_, k) for (s_, t_), k in rows.items()), dict(counters)


def test_config5_scale_errors_reach_every_search_tier():
    """the 100 k-guide searches beyond one mismatch were only ever checked on a few hundred reads.  Here
    9 % substitutions and 2 % N over the full library: nearly every read needs its constant region repaired, most
    captures miss the exact tier, thousands go through the coarse and the full seed index, hundreds carry several
    N -- every read's outcome against the oracle, then the rows"""
    import torch
    import ngs_barcode_count_amd as pkg
    w = workloads.make("config5", synth_overrides=dict(p_sub=0.09, p_n=0.02))
    n = 24_000
    R = w.read_len
    seq, qual = w.synth.generate_host(0, n)
    d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    d_idx = torch.zeros(n, dtype=torch.int64, device="cuda")
    eng = pkg.Engine(w.plan, device=0)
    eng.trace(d_out.data_ptr(), d_idx.data_ptr())
    _run(w, 0, n, eng=eng, chunk=n)
    exp_out, exp_rows, exp_counters = _oracle_outcomes_parallel(w, seq, None)
    got_out = d_out.cpu().numpy()
    bad = np.nonzero(got_out != exp_out)[0]
    assert bad.size == 0, (bad[:5], got_out[bad[:5]], exp_out[bad[:5]],
                           [bytes(seq[i * R:(i + 1) * R]).decode() for i in bad[:2]])
    got = eng.counters()
    assert {k: got[k] for k in exp_counters} == exp_counters
    # the workload really exercises what it is meant to: failures of both kinds and plenty of corrected reads
    assert got["barcode"] > 400 and got["constant_region"] > 100 and got["matched"] > n // 2
    assert eng.result_rows() == exp_rows
    eng.close()


def test_config5_queued_searches_carry_over_between_tiles():
    """the search queue of a wavefront (bc_kernel.h: plans with one known set): enough reads that a wavefront goes
    through several tiles, so captures wait in its queue from one tile to the next and the read a verdict belongs to
    is found again from (tile, lane) -- every read's outcome against the oracle (an outcome slot nobody writes shows
    as 0xFF), then the rows"""
    import torch
    import ngs_barcode_count_amd as pkg
    w = workloads.make("config5")
    n = 420_000  # > 64 reads x 5 waves x 4 SIMDs x 256 CUs: the first ~1,400 wavefronts get a second tile
    R = w.read_len
    seq, qual = w.synth.generate_host(0, n)
    d_out = torch.full((n,), 0xFF, dtype=torch.uint8, device="cuda")
    d_idx = torch.zeros(n, dtype=torch.int64, device="cuda")
    eng = pkg.Engine(w.plan, device=0)
    eng.trace(d_out.data_ptr(), d_idx.data_ptr())
    _run(w, 0, n, eng=eng, chunk=n)
    exp_out, exp_rows, exp_counters = _oracle_outcomes_parallel(w, seq, None)
    got_out = d_out.cpu().numpy()
    bad = np.nonzero(got_out != exp_out)[0]
    assert bad.size == 0, (bad.size, bad[:5], got_out[bad[:5]], exp_out[bad[:5]])
    got = eng.counters()
    assert {k: got[k] for k in exp_counters} == exp_counters
    assert eng.result_rows() == exp_rows
    # a matched read's traced index is its guide's row
    idx = d_idx.cpu().numpy()
    hist = np.bincount(idx[got_out == 0], minlength=len(w.counted[0]))
    by_seq = {t_: k for _, t_, k in exp_rows}
    guides = [g.decode() if isinstance(g, bytes) else g for g in w.counted[0]]
    assert sum(by_seq.values()) == int(hist.sum())
    assert all(by_seq.get(guides[g], 0) == int(c) for g, c in enumerate(hist))
    eng.close()


def test_config4_bench_variant_vs_oracle():
    """config 4 as bench.py runs it: PCR copies per molecule geometric with mean 2, scattered over the job by a fixed
    permutation (geo_total = the job's read count); duplicates, distinct counts and rows against the oracle"""
    n = 120_000
    w = workloads.make("config4", geo_total=n)
    eng = _run(w, 0, n, chunk=50_000)
    seq, qual = w.synth.generate_host(0, n)
    o = workloads.oracle_for(w)
    o.process_batch(seq, qual, w.read_len, w.read_len)
    got = eng.counters()
    for k, v in o.counters.items():
        assert got[k] == v, (k, got, o.counters)
    assert n // 4 < got["duplicates"] < 3 * n // 4  # mean 2 copies per molecule: about half of the passing reads
    assert eng.result_rows() == o.rows()
    eng.close()


@pytest.mark.parametrize("variant", ["config3", "geometric", "zipf"])
def test_synth_device_equals_host(variant):
    import torch
    if variant == "geometric":
        w = workloads.make("config4", n_sets=(4, 100, 100, 100), geo_total=1 << 40)
    elif variant == "zipf":
        w = workloads.make("config5", n_sets=(5000,), zipf=True)
    else:
        w = workloads.make("config3", n_sets=(4, 100, 100, 100))
    n = 5000
    seq, qual = w.synth.generate_host(123456789012, n)
    dseq = torch.empty(n * 100, dtype=torch.uint8, device="cuda")
    dqual = torch.empty(n * 100, dtype=torch.uint8, device="cuda")
    w.synth.generate_device(0, None, 123456789012, n, dseq.data_ptr(), dqual.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(dseq.cpu().numpy(), seq) and np.array_equal(dqual.cpu().numpy(), qual)


@pytest.mark.parametrize("name,n", [("config2", 10_000_000), ("config3", 20_000_000)])
def test_full_size_properties(name, n):
    import torch
    import ngs_barcode_count_amd as pkg
    w = workloads.make(name)
    whole = _run(w, 0, n)
    c = whole.counters()
    six = sum(c[k] for k in ("matched", "constant_region", "sample_barcode", "barcode", "duplicates", "low_quality"))
    assert six == n == c["total_reads"] and c["unsupported_reads"] == 0  # one outcome per read (README.md:160-165)
    table = torch.empty(0)
    s, b, cnt = whole.rows()
    assert int(cnt.sum()) == c["matched"]  # every matched read landed in exactly one table entry
    if name == "config2":
        assert c["matched"] == n  # clean reads, exact matching: everything is counted
    # shard additivity: two halves into one engine (what per-GPU shards + a sum-reduce rely on)
    halves = _run(w, 0, n // 2)
    _run(w, n // 2, n - n // 2, eng=halves)
    assert halves.counters() == c
    s2, b2, cnt2 = halves.rows()
    key = lambda s_, b_, c_: sorted(zip(s_.tolist(), map(tuple, b_.tolist()), c_.tolist()))
    assert key(s, b, cnt) == key(s2, b2, cnt2)
    whole.close()
    halves.close()


@pytest.mark.parametrize("name,n", [("config3", 100_000_000), ("config5", 125_000_000)])
def test_full_shard_properties_on_the_device(name, n):
    """BASELINE.json's own sizes (config 3: 100 M reads; config 5: one GPU's shard of 125 M): one outcome per read, the
    table sums to the matched reads, two half shards add up to the whole shard entry by entry (what per-GPU shards + one
    sum-reduce rely on), and a second run reproduces the table bit for bit.  Checked on the device: 89 M result rows are
    not brought to the host."""
    import torch
    import ngs_barcode_count_amd as pkg
    w = workloads.make(name)
    entries = w.plan.table_entries
    tabs = [torch.zeros(entries, dtype=torch.int32, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    whole = _run(w, 0, n, eng=pkg.Engine(w.plan, device=0, table_ptr=tabs[0].data_ptr()), chunk=1 << 24)
    c = whole.counters()
    six = sum(c[k] for k in ("matched", "constant_region", "sample_barcode", "barcode", "duplicates", "low_quality"))
    assert six == n == c["total_reads"] and c["unsupported_reads"] == 0 and c["duplicates"] == 0
    assert int(tabs[0].sum(dtype=torch.int64)) == c["matched"]
    assert 0.5 * n < c["matched"] < n
    halves = _run(w, 0, n // 2, eng=pkg.Engine(w.plan, device=0, table_ptr=tabs[1].data_ptr()), chunk=1 << 24)
    _run(w, n // 2, n - n // 2, eng=halves, chunk=1 << 24)
    assert halves.counters() == c
    assert torch.equal(tabs[0], tabs[1])
    again = _run(w, 0, n, eng=pkg.Engine(w.plan, device=0, table_ptr=tabs[2].data_ptr()), chunk=(1 << 24) + 4096)
    assert again.counters() == c and torch.equal(tabs[0], tabs[2])  # other batch boundaries, same table
    for e in (whole, halves, again):
        e.close()


def test_specialised_kernel_takes_over_in_the_background(monkeypatch, tmp_path):
    """default mode, empty kernel cache: small batches start on the generic kernel; after 2^20 reads the
    specialised one is compiled on a worker thread and later submits switch to it -- same counts."""
    import time
    import torch
    import ngs_barcode_count_amd as pkg
    import workloads
    monkeypatch.setenv("BC_JIT", "1")
    monkeypatch.setenv("BC_JIT_CACHE", str(tmp_path / "cache"))
    w = workloads.make("config3", n_sets=(4, 37, 41, 43))  # a shape no other test (or build()) has cached
    n, R = 400_000, w.read_len
    dseq = torch.empty(n * R, dtype=torch.uint8, device="cuda")
    dqual = torch.empty(n * R, dtype=torch.uint8, device="cuda")
    w.synth.generate_device(0, None, 0, n, dseq.data_ptr(), dqual.data_ptr())
    torch.cuda.synchronize()
    ref = pkg.Engine(w.plan, device=0)
    ref.submit_device(dseq.data_ptr(), dqual.data_ptr(), n, R, R)
    one = ref.counters()
    assert ref.kernel_name().startswith("match_count_kernel")  # below the threshold: nothing compiled
    ref.close()
    eng = pkg.Engine(w.plan, device=0)
    names, batches, t0 = [], 0, time.time()
    while time.time() - t0 < 120:
        eng.submit_device(dseq.data_ptr(), dqual.data_ptr(), n, R, R)
        eng.sync()
        batches += 1
        names.append(eng.kernel_name())
        if names[-1].startswith("bc_jit_match_count") and batches >= 6:
            break
        time.sleep(0.2)
    assert names[0].startswith("match_count_kernel"), names[:3]
    assert names[-1].startswith("bc_jit_match_count"), (names[-3:], batches)
    got = eng.counters()
    for k, v in one.items():
        assert got[k] == v * batches, (k, got, one, batches)
    assert os.listdir(str(tmp_path / "cache"))  # the code object was stored for the next run
    eng.close()
    # a second engine finds it in the cache and starts on the specialised kernel
    eng2 = pkg.Engine(w.plan, device=0)
    eng2.submit_device(dseq.data_ptr(), dqual.data_ptr(), n, R, R)
    assert eng2.kernel_name().startswith("bc_jit_match_count")
    assert eng2.counters() == one
    eng2.close()


@pytest.mark.parametrize("jit", ["0", "force"])
def test_many_odd_sized_submits_equal_one_submit(monkeypatch, tmp_path, jit):
    """the pipelined persistent kernel at every kind of batch edge: 3 M reads cut into ~300 submits of random sizes
    (1 .. 40,000 reads, so full tiles, partial tiles, single reads, grids smaller than the chip) give exactly the
    counters and the table of one submit"""
    import torch
    import ngs_barcode_count_amd as pkg
    import workloads
    monkeypatch.setenv("BC_JIT", jit)
    monkeypatch.setenv("BC_JIT_CACHE", str(tmp_path / "cache"))
    w = workloads.make("config3", n_sets=(4, 50, 60, 70))
    n, R = 3_000_000, w.read_len
    dseq = torch.empty(n * R, dtype=torch.uint8, device="cuda")
    dqual = torch.empty(n * R, dtype=torch.uint8, device="cuda")
    w.synth.generate_device(0, None, 0, n, dseq.data_ptr(), dqual.data_ptr())
    torch.cuda.synchronize()
    entries = w.plan.table_entries
    t_one = torch.zeros(entries, dtype=torch.int32, device="cuda")
    t_many = torch.zeros(entries, dtype=torch.int32, device="cuda")
    one = pkg.Engine(w.plan, device=0, table_ptr=t_one.data_ptr())
    one.submit_device(dseq.data_ptr(), dqual.data_ptr(), n, R, R)
    many = pkg.Engine(w.plan, device=0, table_ptr=t_many.data_ptr())
    rng = np.random.default_rng(77)
    done = 0
    while done < n:
        k = int(min(n - done, rng.choice([1, 63, 64, 65, 255, 256, 257, int(rng.integers(1, 40001))])))
        # batches must start 16-byte aligned: advance in multiples of 4 reads (R = 100 -> 400 bytes)
        k = max(4, k - k % 4) if done + k < n else k
        many.submit_device(dseq.data_ptr() + done * R, dqual.data_ptr() + done * R, k, R, R)
        done += k
    assert many.counters() == one.counters()
    one.sync()
    assert torch.equal(t_one, t_many)
    one.close()
    many.close()
