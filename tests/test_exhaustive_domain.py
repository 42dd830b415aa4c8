"""Exhaustive-domain parity on the CPU (tests/exhaustive.py): the C oracle against the independent Python restatement
read by read, and the kernel's lane code (both counting forms) against the oracle read by read."""
import numpy as np
import pytest

import emu_lib
import exhaustive
import oracle_lib
import parity
import pyref

STRIDE = 12


def _oracle():
    return parity.oracle_for(exhaustive.case())


def test_oracle_vs_pyref_every_read_up_to_length_7():
    """2.4 M (read, quality) pairs -- every read of up to 7 bases with every quality string: every outcome and every
    row of the two independent restatements agree"""
    seq, qual, lens = exhaustive.all_lengths(7, STRIDE)
    o = _oracle()
    outc = o.process_batch_outcomes(seq.reshape(-1), qual.reshape(-1), STRIDE, STRIDE, lens=lens)
    p = pyref.Parser(exhaustive.SCHEME, samples=exhaustive.SAMPLES, counted=exhaustive.COUNTED, **exhaustive.KWARGS)
    names = oracle_lib.NAMES
    sb, qb = seq.tobytes(), qual.tobytes()
    for i in range(seq.shape[0]):
        ln = int(lens[i])
        s = sb[i * STRIDE:i * STRIDE + ln].decode()
        q = qb[i * STRIDE:i * STRIDE + ln].decode()
        # (below 7 bases the reference's usize underflows, parse.rs:291: both restatements call it a constant-region error)
        assert p.process(s, q) == names[outc[i]], (s, q, names[outc[i]])
    assert p.rows() == o.rows() and o.counters["matched"] > 0
    assert all(o.counters[k] == p.counters[k] for k in names)


def test_oracle_vs_pyref_length_8_and_9_sampled_qualities():
    """every read of 8 and 9 bases (327,680 sequences; repairs need a window that is not the last) with six quality
    strings each"""
    o = _oracle()
    p = pyref.Parser(exhaustive.SCHEME, samples=exhaustive.SAMPLES, counted=exhaustive.COUNTED, **exhaustive.KWARGS)
    names = oracle_lib.NAMES
    for ln in (8, 9):
        pats = ["I" * ln, "#" * ln, ("I#" * ln)[:ln], ("#I" * ln)[:ln], ("II##" * ln)[:ln], ("#II#I" * ln)[:ln]]
        for r in range(4 ** ln):
            s = "".join("ACGN"[(r >> (2 * k)) & 3] for k in range(ln))
            for q in pats:
                assert p.process(s, q) == o.process(s, q), (s, q)
    assert p.rows() == o.rows() and o.counters["matched"] > 0
    assert all(o.counters[k] == p.counters[k] for k in names)


@pytest.mark.parametrize("variant", ["generic", "static"])
def test_lane_code_vs_oracle_every_read_up_to_length_8(variant):
    """19.2 M (read, quality) pairs: the lane code's outcome and table index against the oracle, read by read"""
    plan = emu_lib.make_plan(exhaustive.case(), variant)
    total = 0
    for ln in range(0, 9):
        seq, qual = exhaustive.domain(ln, STRIDE)
        n = seq.shape[0]
        for a in range(0, n, 1 << 21):
            b = min(n, a + (1 << 21))
            s, q = np.ascontiguousarray(seq[a:b]).reshape(-1), np.ascontiguousarray(qual[a:b]).reshape(-1)
            lens = np.full(b - a, ln, dtype=np.uint16)
            outc, idx, entries, discard = emu_lib.emulate(plan, s, q, lens, STRIDE, STRIDE)
            o = _oracle()
            exp = o.process_batch_outcomes(s, q, STRIDE, STRIDE, lens=lens)
            bad = np.nonzero(outc != exp)[0]
            assert bad.size == 0, (ln, a + int(bad[0]), bytes(seq[a + bad[0], :ln]), bytes(qual[a + bad[0], :ln]),
                                   int(outc[bad[0]]), int(exp[bad[0]]))
            counts = {}
            for k in idx[outc == 0].tolist():
                counts[k] = counts.get(k, 0) + 1
            assert parity.decode_rows(plan, counts, discard) == o.rows(), ln
            total += b - a
    assert total == sum(8 ** k for k in range(9))
