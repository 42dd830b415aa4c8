"""Cross-checks the C oracle against the independent Python restatement on adversarial reads
(ragged lengths, constructs flush with the 3' end, N, foreign bytes, ties, prefix compares,
PCR duplicates).  Two independently written restatements agreeing is what stands in for the
reference binary, which cannot be built in this image (no Rust toolchain)."""
import pytest

import cases
import oracle_lib
import pyref


@pytest.mark.parametrize("name", cases.ALL_CASES)
def test_oracle_vs_pyref(name):
    c = cases.build_case(name, seed=1, n=400)
    kw = c["kwargs"]
    o = oracle_lib.Oracle(c["scheme"], samples=c["samples"], counted=c["counted"], **kw)
    p = pyref.Parser(c["scheme"], samples=c["samples"], counted=c["counted"], **kw)
    outcomes = set()
    for seq, qual in c["reads"]:
        a = o.process(seq, qual)
        b = p.process(seq, qual)
        assert a == b, (seq, qual)
        outcomes.add(a)
    assert o.counters == p.counters
    assert o.rows() == p.rows()
    assert sum(o.counters.values()) == len(c["reads"])  # exactly one outcome per read (README.md:160-165)
    assert len(outcomes) >= 2


@pytest.mark.parametrize("seed", range(40))
def test_oracle_vs_pyref_on_randomly_drawn_schemes(seed):
    """the two independent restatements (C, string level; Python, the reference's regex through `re`) on schemes
    neither was written against"""
    c = cases.random_case(seed, n=300)
    kw = c["kwargs"]
    o = oracle_lib.Oracle(c["scheme"], samples=c["samples"], counted=c["counted"], **kw)
    p = pyref.Parser(c["scheme"], samples=c["samples"], counted=c["counted"], **kw)
    for seq, qual in c["reads"]:
        assert o.process(seq, qual) == p.process(seq, qual), (seq, qual)
    assert o.counters == p.counters
    assert o.rows() == p.rows()


def test_scheme_quirks_agree():
    for text in ["[4]ACGT{3}", "# c\n{5}\nacgt\n(3)\n", "ACGTNNAC{4}NTT", "nnACGT{3}", "AC GT{2}x[3]", "{3}{4}AC",
                 "ACGT\r\n{3}\r\n"]:
        o = oracle_lib.Oracle(text)
        f = pyref.SequenceFormat(text)
        assert o.format_string == f.format_string
        assert o.regions_string == f.regions_string
        assert o.regex_string == f.regex_string
        assert o.constant_region_length == f.constant_region_length
    with pytest.raises(ValueError):
        oracle_lib.Oracle("[3]AC[4]")  # duplicate group name: Regex::new fails (Appendix A Q13)


def test_csv_loader_quirks():
    # header skipped, no trimming, duplicate keeps last ID, short rows -> ("","")  (Appendix A Q14)
    o = oracle_lib.Oracle("[4]AC{3}", sample_csv="h1,h2\nACGT,one\nACGT,two\nTTTT ,x,extra\n\n")
    assert o.sample_keys() == ["", "ACGT", "TTTT "]
    assert oracle_lib.lib().orc_sample_id(o._c, b"ACGT") == b"two"
    with pytest.raises(ValueError):
        oracle_lib.Oracle("[4]AC{3}", counted_csv="a,b,c\nACG,id\n")
    with pytest.raises(ValueError):
        oracle_lib.Oracle("[4]AC{3}TT{3}", counted_csv="a,b,c\nACG,id,1\n")  # barcode 2 missing
    o = oracle_lib.Oracle("[4]AC{3}TT{3}", counted_csv="a,b,c\nACG,id,1\nTTT,q,2\nACG,id2,1\n")
    assert oracle_lib.lib().orc_counted_id(o._c, 0, b"ACG") == b"id2"
