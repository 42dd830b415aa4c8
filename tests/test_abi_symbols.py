"""The C-ABI library loads (no GPU needed) and exports every function include/*.h declares;
the host-only half of the ABI (plan, budgets, CSV loaders, generator) works without a device."""
import ctypes as C
import glob
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names += re.findall(r"\b(bc_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    import ngs_barcode_count_amd as pkg
    lib = pkg._lib.load()
    names = declared_functions()
    assert len(names) >= 50
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    bound = set(pkg._lib.PLAN_API) | set(pkg._lib.ENGINE_API)
    assert set(names) == bound, set(names) ^ bound


def test_host_half_of_the_abi_without_a_gpu():
    import ngs_barcode_count_amd as pkg
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "kat.json")))
    p = pkg.Plan(kat["scheme"])
    assert p.format_string == kat["format_string"] and p.regions_string == kat["regions_string"]
    assert p.regex_string == kat["regex_string"]
    sf = pkg.SequenceFormat(p)
    assert str(sf).startswith("-FORMAT-\n" + kat["format_string"])
    for k in kat["max_seq_errors"]:  # the reference's doctest values, src/info.rs:479-611
        m = pkg.MaxSeqErrors(*k["args"], 0.0)
        assert [m.max_constant_errors(), m.max_sample_errors(), m.max_barcode_errors()] == k["expect"]
    with pytest.raises(pkg.BarcodeCountError):
        pkg.Plan("[3]AC[4]")  # duplicate group name: Regex::new fails in the reference
    p2 = pkg.Plan("[4]AC{3}TT{3}")
    p2.load_counted_csv("a,b,c\nACG,id,1\nTTT,q,2\nACG,id2,1\n")
    assert p2.counted(0) == [("ACG", "id2")] and p2.counted(1) == [("TTT", "q")]
    with pytest.raises(pkg.BarcodeCountError):
        p2.load_counted_csv("a,b,c\nACG,id\n")
    p2.load_sample_csv("h1,h2\nACGT,one\nACGT,two\nTTTT ,x,extra\n\n")
    assert p2.samples() == [("ACGT", "two"), ("TTTT ", "x"), ("", "")]


def test_no_engine_without_a_device():
    """the product has no CPU path: without a HIP device engine creation fails loudly"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import ngs_barcode_count_amd as pkg
    p = pkg.Plan("ACGTACGT{8}TTGG")
    p.add_counted(0, "ACGTACGT", "x")
    with pytest.raises(pkg.BarcodeCountError):
        pkg.Engine(p, device=0)
    with pytest.raises(pkg.BarcodeCountError):
        pkg.fix_error("ACGT", ["ACGA"], 1)


def test_generator_is_deterministic_and_matches_its_spec():
    import numpy as np
    import workloads
    w = workloads.make("config3", n_sets=(4, 50, 50, 50))
    a, qa = w.synth.generate_host(1000, 300)
    b, qb = w.synth.generate_host(1100, 100)
    assert np.array_equal(a[100 * 100:200 * 100], b[:100 * 100])  # read i depends on (seed, i) only
    assert set(np.unique(a)) <= set(b"ACGTN")
    assert qa.min() >= 33 + 2 and qa.max() <= 33 + 40
    o = workloads.oracle_for(w)
    o.process_batch(a, qa, 100, 100)
    c = o.counters
    assert c["matched"] > 200 and sum(c.values()) == 300
