"""The kernel's lane code (csrc/bc_lane.h: bit-plane packing, shifted-vector anchor, bit-sliced
repair, quality thresholds, table/hash/scan barcode lookup) executed on the host through
tests/emu and compared read by read with the CPU oracle.  The GPU parity tests
(tests/test_gpu_parity.py) run the same cases through the real kernel."""
import numpy as np
import pytest

import cases
import emu_lib
import parity
import readgen

NO_RANDOM = cases.NO_RANDOM_CASES


# "static": the counting form of the scheme-specialised kernels (bc_lane.h count_mismatches_static) compiled into the
# emulation, so that it is checked here too and not only on the GPU
VARIANTS = ["generic", "static"]


@pytest.mark.parametrize("name", NO_RANDOM)
@pytest.mark.parametrize("use_lens", [False, True])
@pytest.mark.parametrize("variant", VARIANTS)
def test_lane_code_vs_oracle(name, use_lens, variant):
    c = cases.build_case(name, seed=3 + use_lens, n=500)
    if not use_lens:
        # uniform length: truncate / drop so every read has the same length
        rl = min(len(s) for s, _ in c["reads"])
        c["reads"] = [(s[:rl], q[:rl]) for s, q in c["reads"]]
    plan = emu_lib.make_plan(c, variant)
    seq, qual, lens = readgen.to_arrays(c["reads"])
    stride = seq.shape[1]
    outc, idx, entries, discard = emu_lib.emulate(plan, seq.reshape(-1), qual.reshape(-1), lens if use_lens else None,
                                                  stride, stride)
    parity.check_per_read(c, plan, outc, idx, discard)


@pytest.mark.parametrize("name", ["del_mismatch_quality", "raw_counted", "long_gaps"])
def test_quality_bytes_below_33_wrap_like_the_reference(name):
    c = cases.with_wrapping_quality(cases.build_case(name, seed=17, n=600), seed=1)
    plan = emu_lib.make_plan(c)
    seq, qual, lens = readgen.to_arrays(c["reads"])
    stride = seq.shape[1]
    outc, idx, entries, discard = emu_lib.emulate(plan, seq.reshape(-1), qual.reshape(-1), lens, stride, stride)
    o = parity.check_per_read(c, plan, outc, idx, discard)
    assert o.counters["low_quality"] > 0 and o.counters["matched"] > 0


@pytest.mark.parametrize("seed", range(40))
@pytest.mark.parametrize("variant", VARIANTS)
def test_randomly_drawn_schemes(seed, variant):
    """schemes, sets, budgets and thresholds drawn at random (cases.random_case): lane code vs oracle"""
    c = cases.random_case(seed, n=300)
    plan = emu_lib.make_plan(c, variant)
    seq, qual, lens = readgen.to_arrays(c["reads"])
    stride = seq.shape[1]
    outc, idx, entries, discard = emu_lib.emulate(plan, seq.reshape(-1), qual.reshape(-1), lens, stride, stride)
    parity.check_per_read(c, plan, outc, idx, discard)


@pytest.mark.parametrize("name", cases.RANDOM_ENGINE_CASES)
def test_random_barcode_keys_vs_oracle(name):
    """random-barcode schemes: the lane code's (tuple, random) key + set semantics vs the oracle's
    duplicate collapse (info.rs:770-802)"""
    c = cases.build_case(name, seed=7, n=700)
    plan = emu_lib.make_plan(c)
    seq, qual, lens = readgen.to_arrays(c["reads"])
    stride = seq.shape[1]
    outc, idx, entries, discard, rcode, rspace = emu_lib.emulate(plan, seq.reshape(-1), qual.reshape(-1), lens, stride,
                                                                 stride, with_random=True)
    assert rspace in (5 ** 7, 5 ** 8, 5 ** 12) or name.startswith("rnd_rb_")
    out2, keys = parity.apply_set_semantics(outc, idx, rcode, rspace)
    o = parity.check_per_read(c, plan, out2, keys, discard, rspace)
    assert name == "example_files_samples" or name.startswith("rnd_rb_") or o.counters["duplicates"] > 0


@pytest.mark.parametrize("variant", VARIANTS)
def test_long_reads_use_wider_planes(variant):
    for rl in (150, 250, 300):
        c = cases.build_case("del_mismatch_quality", seed=rl, n=60)
        rng = np.random.default_rng(rl)
        c["reads"] = readgen.gen_reads(rng, c["scheme"], 200, rl, list(c["samples"]), c["counted"], p_sub=0.02,
                                       p_n=0.004)
        plan = emu_lib.make_plan(c, variant)
        seq, qual, lens = readgen.to_arrays(c["reads"])
        outc, idx, entries, discard = emu_lib.emulate(plan, seq.reshape(-1), qual.reshape(-1), None, rl, rl)
        parity.check_per_read(c, plan, outc, idx, discard)


def test_plan_matches_oracle_scheme_compile():
    import json, os
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
    import ngs_barcode_count_amd as pkg
    p = pkg.Plan(kat["scheme"], lib=emu_lib.lib())
    assert p.format_string == kat["format_string"]
    assert p.regions_string == kat["regions_string"]
    assert p.regex_string == kat["regex_string"]
    assert p.constant_region_length == kat["constant_region_length"]
    b = kat["budgets"]
    assert (p.max_constant_errors, p.max_sample_errors, p.max_barcode_errors) == (b["constant"], b["sample"],
                                                                                  b["barcodes"])
    for row in kat["quality_thresholds"]:
        p.set_min_quality(row["min_quality"])
        for n, T in row["T"].items():
            assert p.quality_threshold(int(n)) == T
    for k in kat["max_seq_errors"]:
        m = pkg.MaxSeqErrors(k["args"][0], k["args"][1], k["args"][2], k["args"][3], k["args"][4], k["args"][5], 0.0,
                             lib=emu_lib.lib())
        assert [m.max_constant_errors(), m.max_sample_errors(), m.max_barcode_errors()] == k["expect"]


def _single_n_case(seed, n=1500):
    """clean constructs whose barcodes carry one or two 'N's, against dense sets with many ties"""
    rng = np.random.default_rng(seed)
    c = cases.build_case("del_dense_ties", seed=seed, n=10)
    samples, counted = list(c["samples"]), c["counted"]
    reads = []
    for i in range(n):
        parts = ["AGCTACGAATCG", "TGGA", "TGGA", "ACTAGAT"]
        caps = [samples[rng.integers(len(samples))]] + [counted[b][rng.integers(len(counted[b]))] for b in range(3)]
        caps = [list(readgen.mutate(rng, x, 0.08, 0.0)) for x in caps]
        for _ in range(1 + (i % 3 == 0)):
            g = rng.integers(4)
            caps[g][rng.integers(8)] = "N"
        caps = ["".join(x) for x in caps]
        construct = caps[0] + parts[0] + caps[1] + parts[1] + caps[2] + parts[2] + caps[3] + parts[3]
        off = int(rng.integers(0, 100 - len(construct)))
        seq = readgen.rand_seq(rng, off) + construct + readgen.rand_seq(rng, 100 - len(construct) - off)
        reads.append((seq, "I" * 100))
    c["reads"] = reads
    c["kwargs"] = dict(max_barcode=int(rng.integers(0, 3)), max_sample=int(rng.integers(0, 3)))
    return c


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_single_n_captures(seed):
    c = _single_n_case(seed)
    plan = emu_lib.make_plan(c)
    seq, qual, lens = readgen.to_arrays(c["reads"])
    outc, idx, entries, discard = emu_lib.emulate(plan, seq.reshape(-1), qual.reshape(-1), None, 100, 100)
    parity.check_per_read(c, plan, outc, idx, discard)
    assert 0 < int((outc == 0).sum()) < len(outc)


def test_foreign_byte_in_a_raw_capture_only_matters_for_counted_reads():
    """A raw (file-less) sample capture holding a byte outside ACGTN cannot be coded into the key; the engine
    refuses such a read (BC_UNSUPPORTED_READS) -- but only if it would have been counted: when a later group
    fails, that failure is the outcome, exactly as in the reference (parse.rs:453-454, 487-505)."""
    import numpy as np
    c = dict(name="raw_foreign", scheme="[6]ACGTACGGT{6}TTGGCCAA", samples=None, counted=[["AACCGG", "TTGGAA"]],
             kwargs=dict(max_barcode=0), reads=[])
    q = "I" * 40
    good = "GATTxC" + "ACGTACGGT" + "AACCGG" + "TTGGCCAA"
    bad = "GATTxC" + "ACGTACGGT" + "ACACAC" + "TTGGCCAA"
    clean = "GATTAC" + "ACGTACGGT" + "TTGGAA" + "TTGGCCAA"
    for s in (good, bad, clean):
        s = (s + "ACGT" * 10)[:40]
        c["reads"].append((s, q))
    plan = emu_lib.make_plan(c)
    seq, qual, lens = readgen.to_arrays(c["reads"])
    outc, idx, entries, discard = emu_lib.emulate(plan, seq.reshape(-1), qual.reshape(-1), lens, 40, 40)
    o = parity.oracle_for(c)
    exp = [o.process(s, qq) for s, qq in c["reads"]]
    assert exp == ["matched", "barcode", "matched"]
    assert [int(v) for v in outc] == [7, parity.CODE["barcode"], parity.CODE["matched"]]


def test_quality_line_of_another_length_than_the_sequence_line():
    """scores.iter().skip(start).zip(regions) (parse.rs:340-345): what counts is the quality line's own length"""
    rng = np.random.default_rng(77)
    c = cases.build_case("del_mismatch_quality", seed=53, n=800)
    reads = []
    for s, q in c["reads"]:
        r = rng.random()
        if r < 0.2:
            q = q[:int(rng.integers(0, len(q)))]
        elif r < 0.3:
            q = q + "I" * int(rng.integers(1, 30))
        reads.append((s, q))
    c["reads"] = reads
    plan = emu_lib.make_plan(c)
    n = len(reads)
    st = (max(len(s) for s, _ in reads) + 3) // 4 * 4
    seq = np.full((n, st), ord("N"), dtype=np.uint8)
    qual = np.full((n, st), ord("!"), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint16)
    qlens = np.zeros(n, dtype=np.uint16)
    for i, (s, q) in enumerate(reads):
        seq[i, :len(s)] = np.frombuffer(s.encode(), dtype=np.uint8)
        qq = q[:st]  # only the first `stride` quality bytes are stored; the length travels whole
        qual[i, :len(qq)] = np.frombuffer(qq.encode(), dtype=np.uint8)
        lens[i], qlens[i] = len(s), len(q)
    outc, idx, entries, discard = emu_lib.emulate(plan, seq.reshape(-1), qual.reshape(-1), lens, st, st, qlens=qlens)
    o = parity.check_per_read(c, plan, outc, idx, discard)
    assert o.counters["low_quality"] > 0 and o.counters["matched"] > 0


@pytest.mark.parametrize("variant", VARIANTS)
def test_lower_case_constants_anchor_but_never_repair(variant):
    """info.rs:298-299: the regex gets the upper-cased constants, format_string keeps them as written, so a repaired
    read carries lower-case letters the regex cannot match: anchored reads count as usual, reads that would need a
    repair are constant-region errors"""
    c = cases.build_case("del_mismatch_quality", seed=71, n=800)
    upper = parity.oracle_for(c)
    for s, q in c["reads"]:
        upper.process(s, q)
    c["scheme"] = "[8]AGCTacgaATCG{8}TGGA{8}tgga{8}ACTAGAT"
    plan = emu_lib.make_plan(c, variant)
    assert plan.format_string == "NNNNNNNNAGCTacgaATCGNNNNNNNNTGGANNNNNNNNtggaNNNNNNNNACTAGAT"
    seq, qual, lens = readgen.to_arrays(c["reads"])
    stride = seq.shape[1]
    outc, idx, entries, discard = emu_lib.emulate(plan, seq.reshape(-1), qual.reshape(-1), lens, stride, stride)
    o = parity.check_per_read(c, plan, outc, idx, discard)
    assert o.counters["constant_region"] > upper.counters["constant_region"] and o.counters["matched"] > 0


@pytest.mark.parametrize("variant", VARIANTS)
def test_token_mixing_upper_and_lower_case_n(variant):
    """info.rs:287-295: of a token like "NnN" the regex takes two valid bases ([AGCT]{2}: only the upper-case N's are
    counted) while format_string keeps all three characters -- the format is one longer than what the regex matches.
    Anchored reads count by the regex's geometry; a read that needs its constant region repaired is rebuilt from
    format_string, 'n' and all, and can no longer match (the plan proves that for the scheme, else it refuses it)."""
    import ngs_barcode_count_amd as pkg
    c = cases.build_case("del_mismatch_quality", seed=73, n=10)
    rng = np.random.default_rng(73)
    # reads generated for the scheme as the REGEX sees it: two free bases where the token stands
    c["reads"] = readgen.gen_reads(rng, "[8]AGCTACNNGAATCG{8}TGGA{8}TGGA{8}ACTAGAT", 1500, 100, list(c["samples"]), c["counted"],
                                   p_sub=0.01, p_n=0.004)
    c["scheme"] = "[8]AGCTACNnNGAATCG{8}TGGA{8}TGGA{8}ACTAGAT"
    plan = emu_lib.make_plan(c, variant)
    assert plan.format_string == "NNNNNNNNAGCTACNnNGAATCGNNNNNNNNTGGANNNNNNNNTGGANNNNNNNNACTAGAT"
    assert plan.regex_string.count("[AGCT]{2}") == 1 and plan.length == len(plan.format_string) == 62
    seq, qual, lens = readgen.to_arrays(c["reads"])
    stride = seq.shape[1]
    outc, idx, entries, discard = emu_lib.emulate(plan, seq.reshape(-1), qual.reshape(-1), lens, stride, stride)
    o = parity.check_per_read(c, plan, outc, idx, discard)
    assert o.counters["matched"] > 500 and o.counters["constant_region"] > 100
    # a scheme whose repaired reads could still match at a shifted offset is refused, not guessed at
    # ("nN{8}": the repaired read is "n" + nine of the read's own bases, and [AGCT].{8} matches those at offset 1)
    with pytest.raises(pkg.BarcodeCountError) as err:
        pkg.Plan("nN{8}", lib=emu_lib.lib(variant)).mode
    assert "shifted offset" in str(err.value)
