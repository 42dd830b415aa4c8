"""Where a per-GPU shard of a FASTQ file begins (bc_fastq_record_start: the host logic under bc_fastq_count_shard), for
EVERY byte offset of small files built to mislead it: quality lines that begin with '@' (Phred 31) or with '+' (Phred
10), headers that hold '+' and '@', CRLF line ends, a last record without its newline, a trailing partial record.  The
expected answer is computed from the records' true offsets."""
import ctypes as C
import os

import numpy as np
import pytest

import ngs_barcode_count_amd as pkg
from ngs_barcode_count_amd import _lib


def _starts(path, offsets):
    lib = _lib.load()
    out = []
    v = C.c_uint64()
    for o in offsets:
        rc = lib.bc_fastq_record_start(str(path).encode(), int(o), C.byref(v))
        assert rc == 0, _lib.last_error(lib)
        out.append(v.value)
    return out


def _records(seed, n, nl="\n"):
    rng = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        ln = int(rng.integers(1, 60))
        seq = "".join(rng.choice(list("ACGTN"), ln))
        qual = "".join(chr(int(x)) for x in rng.integers(33, 74, ln))
        k = i % 5
        if k == 0:
            qual = "@" + qual[1:]              # a quality line that looks like a header
        elif k == 1:
            qual = "+" + qual[1:]              # ... or like the separator
        elif k == 2 and ln > 1:
            qual = "@+" + qual[2:]
        head = "@r%d +@ %s" % (i, "x" * int(rng.integers(0, 30)))
        plus = "+" + ("r%d" % i if i % 3 == 0 else "")
        recs.append(nl.join([head, seq, plus, qual]) + nl)
    return recs


@pytest.mark.parametrize("variant", ["plain", "crlf", "no_final_newline", "trailing_partial"])
def test_every_offset_finds_the_next_record(tmp_path, variant):
    recs = _records(5, 40, "\r\n" if variant == "crlf" else "\n")
    text = "".join(recs)
    true_starts = np.cumsum([0] + [len(r) for r in recs])[:-1].tolist()
    if variant == "no_final_newline":
        text = text[:-1]
    elif variant == "trailing_partial":
        text += "@partial\nACGT\n"
    path = tmp_path / "reads.fastq"
    path.write_bytes(text.encode())
    size = len(text)
    got = _starts(path, range(size + 3))
    for o, g in zip(range(size + 3), got):
        nxt = [s for s in true_starts if s >= o]
        # past the last true record start: the file's end (a trailing partial record starts no whole record, and whether
        # the last record counts as one starting at or after `o` only matters for offsets inside it)
        exp = nxt[0] if nxt else size
        assert g == exp, (variant, o, g, exp)


def test_wide_plans_are_recognised_on_the_host():
    """the plan layer decides (no GPU) which plans need keys wider than 64 bits; beyond the widest key they are refused"""
    assert pkg.Plan("[20]ACGT{20}TT{20}").mode == "sparse"
    assert pkg.Plan("GTACCAGTC{40}TGCATGGAC").mode == "sparse"
    assert pkg.Plan("ACGT{8}TT(30)GG").mode == "sparse"   # a 30-base random barcode, nothing else raw
    with pytest.raises(pkg.BarcodeCountError) as err:
        pkg.Plan("[60]ACGT{60}TT{60}").mode
    assert "key bits" in str(err.value)
