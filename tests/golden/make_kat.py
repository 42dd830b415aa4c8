"""Writes kat.json: hand-derived known answers + reference doctest values."""
import json
import os

SCHEME = """[10]
AGCTACGAATCG
{6}
TGGA
{6}
TGGA
{6}
ACTAGAT
(8)
TAGA
"""
core = "AAAAAAAAAA" + "AGCTACGAATCG" + "CAGAGA" + "TGGA" + "ATGAAA" + "TGGA" + "GATAGC" + "ACTAGAT" + "ACGTACGT" + "TAGA"
assert len(core) == 67
bad = core[:21] + "C" + core[22:]
assert core[21] == "G"
coreN = core[:12] + "N" + core[13:]
caps = {"sample": "AAAAAAAAAA", "tuple": "CAGAGA,ATGAAA,GATAGC", "random": "ACGTACGT"}


def q(n, ch="I"):
    return ch * n


reads = [
    # id, seq, qual, min_quality, expected outcome, expect captures?
    ("K1", "GGG" + core + "TTTTT", None, 0.0, "matched", True),
    ("K2", bad, None, 0.0, "constant_region", False),
    ("K3", bad + "T", None, 0.0, "matched", True),
    ("K4", "G" + bad, None, 0.0, "constant_region", False),
    ("K5", "G" + bad + "T", None, 0.0, "matched", True),
    ("K6", coreN + "T", None, 0.0, "matched", True),
    ("K7", "GGGGG" + bad + "TTT", "#" * 10 + "I" * 65, 20.0, "low_quality", False),
    ("K7b", "GGGGG" + core + "TTT", "#" * 10 + "I" * 65, 20.0, "matched", True),
]
out = {
    "scheme": SCHEME,
    "format_string": "NNNNNNNNNNAGCTACGAATCGNNNNNNTGGANNNNNNTGGANNNNNNACTAGATNNNNNNNNTAGA",
    "regions_string": "SSSSSSSSSSCCCCCCCCCCCCBBBBBBCCCCBBBBBBCCCCBBBBBBCCCCCCCRRRRRRRRCCCC",
    "regex_string": "(?P<sample>.{10})AGCTACGAATCG(?P<barcode1>.{6})TGGA(?P<barcode2>.{6})TGGA(?P<barcode3>.{6})ACTAGAT(?P<random>.{8})TAGA",
    "constant_region_length": 31,
    "budgets": {"constant": 6, "sample": 2, "barcodes": [1, 1, 1]},
    "reads": [
        {"id": i, "seq": s, "qual": (ql if ql is not None else q(len(s))), "min_quality": mq, "outcome": o,
         "captures": caps if c else None} for (i, s, ql, mq, o, c) in reads
    ],
    "fix_error": [
        # reference doctest, src/parse.rs:540-551
        {"id": "K8a", "query": "AGTAG", "set": ["AGCAG", "ACAAG", "AGCAA"], "max": 1, "expect": "AGCAG", "source": "src/parse.rs:540-551"},
        {"id": "K8b", "query": "AGTAG", "set": ["AGCAG", "AGAAG", "AGCAA"], "max": 1, "expect": None, "source": "src/parse.rs:540-551"},
        {"id": "K9a", "query": "CAGAGA", "set": ["CAGAGAC", "TGATTGC"], "max": 1, "expect": "CAGAGAC", "source": "hand (Q7)"},
        {"id": "K9b", "query": "CAGAGT", "set": ["CAGAGAC", "TGATTGC"], "max": 1, "expect": "CAGAGAC", "source": "hand (Q7)"},
        {"id": "K9c", "query": "CAGTGT", "set": ["CAGAGAC", "TGATTGC"], "max": 1, "expect": None, "source": "hand (Q7)"},
        {"id": "K10a", "query": "NNNNNN", "set": ["CAGAGAC", "TGATTGC"], "max": 1, "expect": None, "source": "hand (Q6)"},
        {"id": "K10b", "query": "CNGAGA", "set": ["CAGAGAC", "TGATTGC"], "max": 1, "expect": "CAGAGAC", "source": "hand (Q6)"},
        {"id": "K11", "query": "ACGT", "set": ["", "TTTT"], "max": 0, "expect": "", "source": "hand (Q7)"},
    ],
    "max_seq_errors": [
        # reference doctests, src/info.rs:479-611
        {"id": "K12a", "args": [None, 10, None, [8, 8, 8], None, 30], "expect": [6, 2, [1, 1, 1]], "source": "src/info.rs:559,583,607"},
        {"id": "K12b", "args": [None, 10, None, [8, 8, 8], 3, 30], "expect": [3, 2, [1, 1, 1]], "source": "src/info.rs:561-563"},
        {"id": "K12c", "args": [3, 10, None, [8, 8, 8], None, 30], "expect": [6, 3, [1, 1, 1]], "source": "src/info.rs:585-587"},
        {"id": "K12d", "args": [None, 10, 2, [8, 8, 8], None, 30], "expect": [6, 2, [2, 2, 2]], "source": "src/info.rs:609-611"},
    ],
    "quality_thresholds": [
        {"min_quality": 20.0, "T": {"6": 120, "8": 160, "10": 200, "12": 240, "20": 400}},
        {"min_quality": 30.000002, "T": {"6": 181, "8": 241, "10": 301, "12": 361, "20": 601}},
    ],
}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat.json"), "w") as f:
    json.dump(out, f, indent=1)
