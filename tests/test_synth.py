"""The synthetic generator's workload variants, on the host: Zipf-like guide abundances and geometric PCR
copies are what they claim to be.  (Device = host equality: tests/test_gpu_workloads.py.)"""
import numpy as np

import ngs_barcode_count_amd as pkg
import workloads


def test_geometric_copies_have_mean_two_and_are_scattered():
    n = 60_000
    w = workloads.make("config4", n_sets=(3, 6, 6, 6), geo_total=n)
    clean = pkg.Synth(w.plan, seed=4, geo_total=n)  # no substitutions: every copy of a molecule gives the same key
    seq, qual = clean.generate_host(0, n)
    o = workloads.oracle_for(w)
    out = o.process_batch_outcomes(seq, qual, 100, 100)
    c = o.counters
    assert c["matched"] + c["duplicates"] == n
    # molecules with c copies make c - 1 duplicates: E[copies] = 2 -> half of the reads are duplicates
    assert abs(c["duplicates"] / n - 0.5) < 0.02, c
    # scattered: the duplicate of a read is not its neighbour -- the first and the second half of the job share molecules
    first_half_dups = int((out[: n // 2] == 4).sum())
    assert 0.2 < first_half_dups / max(c["duplicates"], 1) < 0.45  # (a clustered layout would give ~0.5)


def test_zipf_like_abundances():
    n = 50_000
    w = workloads.make("config5", n_sets=(4096,), zipf=True)
    clean = pkg.Synth(w.plan, seed=5, zipf=True)
    seq, qual = clean.generate_host(0, n)
    o = workloads.oracle_for(w)
    o.process_batch(seq, qual, 100, 100)
    counts = sorted((r[2] for r in o.rows()), reverse=True)
    octaves = 13  # ranks 1..4096 span 13 octaves, each with 1/13 of the reads
    assert abs(counts[0] / n - 1.0 / octaves) < 0.01
    assert abs(sum(counts[1:3]) / n - 1.0 / octaves) < 0.01
    assert abs(sum(counts[3:7]) / n - 1.0 / octaves) < 0.012
    # the abundant guides are scattered over the set, not its first entries
    top = sorted(o.rows(), key=lambda r: -r[2])[:8]
    idx = sorted(w.counted[0].index(r[1]) for r in top)
    assert idx[-1] > 500
