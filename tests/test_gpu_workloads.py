"""BASELINE.json configs on the GPU: oracle parity on a seeded sample, and size-independent
properties at full size (one outcome per read, table sum = matched, shard additivity, determinism)."""
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


def test_synth_device_equals_host():
    import torch
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
