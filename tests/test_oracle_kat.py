"""Pins the CPU oracle (oracle/oracle.c) and the independent Python restatement
(tests/pyref.py) to the known answers in tests/golden/kat.json: the reference's own
doctest values (src/parse.rs:540-551, src/info.rs:479-611) and the hand-derived vectors of
SURVEY.md Appendix B."""
import json
import os

import numpy as np
import pytest

import oracle_lib
import pyref

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))


def test_scheme_compile_oracle():
    o = oracle_lib.Oracle(KAT["scheme"])
    assert o.format_string == KAT["format_string"]
    assert o.regions_string == KAT["regions_string"]
    assert o.regex_string == KAT["regex_string"]
    assert o.constant_region_length == KAT["constant_region_length"]
    assert o.length == 67 and o.barcode_num == 3 and o.barcode_lengths == [6, 6, 6] and o.sample_length == 10
    b = KAT["budgets"]
    assert o.budgets == (b["constant"], b["sample"], b["barcodes"])


def test_scheme_compile_pyref():
    f = pyref.SequenceFormat(KAT["scheme"])
    assert f.format_string == KAT["format_string"]
    assert f.regions_string == KAT["regions_string"]
    assert f.regex_string == KAT["regex_string"]
    assert f.constant_region_length == KAT["constant_region_length"]


def test_example_scheme_file_equals_kat_scheme():
    text = open(os.path.join(os.path.dirname(__file__), "golden", "example_inputs", "scheme.example.txt")).read()
    o = oracle_lib.Oracle(text)
    assert o.format_string == KAT["format_string"] and o.regions_string == KAT["regions_string"]
    assert pyref.SequenceFormat(text).regex_string == KAT["regex_string"]


@pytest.mark.parametrize("r", KAT["reads"], ids=[r["id"] for r in KAT["reads"]])
def test_read_kat(r):
    o = oracle_lib.Oracle(KAT["scheme"], min_quality=r["min_quality"])
    p = pyref.Parser(KAT["scheme"], min_quality=r["min_quality"])
    assert o.process(r["seq"], r["qual"]) == r["outcome"]
    assert p.process(r["seq"], r["qual"]) == r["outcome"]
    if r["captures"]:
        c = r["captures"]
        assert o.rows() == [(c["sample"], c["tuple"], 1)]
        assert p.rows() == [(c["sample"], c["tuple"], 1)]
        # same read again: the random barcode repeats -> duplicate (src/info.rs:786-791)
        assert o.process(r["seq"], r["qual"]) == "duplicates"
        assert p.process(r["seq"], r["qual"]) == "duplicates"
        assert o.rows() == [(c["sample"], c["tuple"], 1)]
    else:
        assert o.rows() == [] and p.rows() == []


@pytest.mark.parametrize("k", KAT["fix_error"], ids=[k["id"] for k in KAT["fix_error"]])
def test_fix_error_kat(k):
    assert oracle_lib.fix_error(k["query"], k["set"], k["max"]) == k["expect"]
    assert pyref.fix_error(k["query"], k["set"], k["max"]) == k["expect"]
    # order independence (Appendix A Q5): AHashSet iteration order is random in the reference
    assert oracle_lib.fix_error(k["query"], k["set"][::-1], k["max"]) == k["expect"]


@pytest.mark.parametrize("k", KAT["max_seq_errors"], ids=[k["id"] for k in KAT["max_seq_errors"]])
def test_max_seq_errors_kat(k):
    c, s, b = oracle_lib.max_seq_errors(*k["args"])
    assert [c, s, b] == k["expect"]
    c, s, b = pyref.max_seq_errors(*k["args"])
    assert [c, s, b] == k["expect"]


def test_quality_threshold_kat():
    """Integer thresholds equivalent to the f32 mean test (Appendix A Q10), derived with the oracle's
    own float arithmetic: T_n = min{s : !(fl32(s/n) < min)}"""
    for row in KAT["quality_thresholds"]:
        mq = np.float32(row["min_quality"])
        for n, T in row["T"].items():
            n = int(n)
            s = 0
            while np.float32(np.float32(s) / np.float32(n)) < mq:
                s += 1
            assert s == T
            # and the restatements agree on both sides of the threshold (run of n then a constant)
            regions = "B" * n + "C"
            for total, expect in ((T - 1, True), (T, False)):
                base, extra = divmod(total, n)
                q = "".join(chr(33 + base + (1 if i < extra else 0)) for i in range(n)) + "I"
                assert pyref.low_quality(q, float(mq), regions, 0) is expect


def test_example_files_prefix_compare_quirk():
    """Config 1 plumbing: barcode.example.csv has 7-nt barcodes while scheme.example.txt captures 6
    (Appendix A Q7): every barcode goes through fix_error on the common prefix and is rewritten to
    the 7-nt reference."""
    d = os.path.join(os.path.dirname(__file__), "golden", "example_inputs")
    scheme = open(os.path.join(d, "scheme.example.txt")).read()
    counted = open(os.path.join(d, "barcode.example.csv")).read()
    o = oracle_lib.Oracle(scheme, counted_csv=counted)
    core = "AAAAAAAAAA" + "AGCTACGAATCG" + "CAGAGA" + "TGGA" + "ATGAAA" + "TGGA" + "GATAGC" + "ACTAGAT" + "ACGTACGT" + "TAGA"
    assert o.process("GG" + core + "TT", "I" * 71) == "matched"
    assert o.rows() == [("AAAAAAAAAA", "CAGAGAC,ATGAAAT,GATAGCT", 1)]
    # sample file with 7-nt samples vs [10]: prefix compare on 7 bases, max 2 mismatches
    samples = open(os.path.join(d, "sample_barcode.example.csv")).read()
    o2 = oracle_lib.Oracle(scheme, counted_csv=counted, sample_csv=samples)
    assert o2.sample_keys() == ["AACTTAC", "AGCATAC"]
    read = "AGCATACGGG" + core[10:]
    assert o2.process(read + "T", "I" * 68) == "matched"
    assert o2.rows() == [("AGCATAC", "CAGAGAC,ATGAAAT,GATAGCT", 1)]
    assert o2.process(core + "T", "I" * 68) == "sample_barcode"
