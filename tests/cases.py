"""Parity-test cases shared by the oracle cross-check (CPU) and the HIP parity tests (GPU)."""
import numpy as np

import readgen

DEL_SCHEME = "[8]AGCTACGAATCG{8}TGGA{8}TGGA{8}ACTAGAT"
DEL_RANDOM_SCHEME = DEL_SCHEME + "(12)TAGA"
EXAMPLE_SCHEME = "[10]\nAGCTACGAATCG\n{6}\nTGGA\n{6}\nTGGA\n{6}\nACTAGAT\n(8)\nTAGA\n"
CRISPR_SCHEME = "TTGTGGAAAGGACGAAACACCG{20}GTTTTAGAGCTAGAAATAGCAAGTT"
FMTN_SCHEME = "[6]ACGTNNACGT{7}TTGNCA{5}GGATCC"
NOSAMPLE_SCHEME = "GATTACA{9}CCTAGG{4}TTAACCGG"
GAP_SCHEME = "[32]AC{32}GT{8}ACGGT"  # first constant at position 32 and 34-base gaps: shift-only steps


def build_case(name, seed=0, n=600):
    """-> dict(scheme, samples {seq:id} | None, counted [list of seqs] | None, kwargs, reads)"""
    if name.startswith("rnd_rb_"):
        return random_case(int(name[7:]) + 7 * seed, n, with_random=True)
    rng = np.random.default_rng(seed + 1000 * (abs(hash(name)) % 1000 if False else sum(map(ord, name))))
    c = dict(name=name, kwargs={})
    if name == "del_exact":
        c["scheme"] = DEL_SCHEME
        s = readgen.make_set(rng, 4, 8, 3)
        c["samples"] = {x: "S%d" % i for i, x in enumerate(s)}
        c["counted"] = [readgen.make_set(rng, 50, 8, 2) for _ in range(3)]
        c["kwargs"] = dict(max_sample=0, max_barcode=0, max_constant=0)
        c["reads"] = readgen.gen_reads(rng, DEL_SCHEME, n, 100, s, c["counted"], p_sub=0.004, p_n=0.001)
    elif name == "del_mismatch_quality":
        c["scheme"] = DEL_SCHEME
        s = readgen.make_set(rng, 4, 8, 3)
        c["samples"] = {x: "S%d" % i for i, x in enumerate(s)}
        c["counted"] = [readgen.make_set(rng, 60, 8, 2) for _ in range(3)]
        c["kwargs"] = dict(min_quality=20.0)
        c["reads"] = readgen.gen_reads(rng, DEL_SCHEME, n, 100, s, c["counted"], p_sub=0.02, p_n=0.004)
    elif name == "del_dense_ties":
        # many close references: ties and ambiguous corrections are common
        c["scheme"] = DEL_SCHEME
        s = readgen.make_set(rng, 6, 8, 1)
        c["samples"] = {x: "S%d" % i for i, x in enumerate(s)}
        c["counted"] = [readgen.make_set(rng, 300, 8, 1) for _ in range(3)]
        c["kwargs"] = dict(max_barcode=2, max_sample=2, max_constant=7, min_quality=25.5)
        c["reads"] = readgen.gen_reads(rng, DEL_SCHEME, n, 100, s, c["counted"], p_sub=0.05, p_n=0.01)
    elif name == "del_random":
        c["scheme"] = DEL_RANDOM_SCHEME
        s = readgen.make_set(rng, 3, 8, 3)
        c["samples"] = {x: "S%d" % i for i, x in enumerate(s)}
        c["counted"] = [readgen.make_set(rng, 5, 8, 3) for _ in range(3)]
        c["reads"] = readgen.gen_reads(rng, DEL_RANDOM_SCHEME, n, 110, s, c["counted"], p_sub=0.01, p_n=0.002,
                                       dup_frac=0.4)
    elif name == "example_files":
        c["scheme"] = EXAMPLE_SCHEME
        c["samples"] = None
        c["counted"] = [["CAGAGAC", "TGATTGC"], ["ATGAAAT", "GCGCCAT"], ["GATAGCT", "TTAGCTA"]]
        c["reads"] = readgen.gen_reads(rng, EXAMPLE_SCHEME, n, 80, ["AAAAAAAAAA", "CCCCCCCCCC"], c["counted"],
                                       p_sub=0.02, p_n=0.003, dup_frac=0.1, var_len=True)
    elif name == "example_files_samples":
        c["scheme"] = EXAMPLE_SCHEME
        c["samples"] = {"AGCATAC": "Sample_name_1", "AACTTAC": "Sample_name_2"}
        c["counted"] = [["CAGAGAC", "TGATTGC"], ["ATGAAAT", "GCGCCAT"], ["GATAGCT", "TTAGCTA"]]
        c["kwargs"] = dict(min_quality=18.0)
        # duplicates: same construct twice
        c["reads"] = readgen.gen_reads(rng, EXAMPLE_SCHEME, n, 78, ["AGCATACGGG", "AACTTACTTT"], c["counted"],
                                       p_sub=0.02, p_n=0.003)
    elif name == "crispr":
        c["scheme"] = CRISPR_SCHEME
        c["samples"] = None
        c["counted"] = [readgen.make_set(rng, 400, 20, 3)]
        c["reads"] = readgen.gen_reads(rng, CRISPR_SCHEME, n, 100, None, c["counted"], p_sub=0.03, p_n=0.003)
    elif name == "large_set_ties":
        # a large set (hash + tiered search path) whose members are only two mismatches apart: captures one
        # mismatch from TWO references (a tie -> no match) occur next to uniquely correctable ones
        c["scheme"] = "TTGTGGAAAGGACGAAACACCG{16}GTTTTAGAGCTAGAAATAGCAAGTT"
        c["samples"] = None
        c["counted"] = [readgen.make_set(rng, 600, 16, 2)]
        c["reads"] = readgen.gen_reads(rng, c["scheme"], n, 100, None, c["counted"], p_sub=0.04, p_n=0.003)
    elif name == "large_set_many_n":
        # 20-nt guides, budget 4 (tier + coarse + full seed index): one capture in two holds an 'N', many hold two to
        # five -- the substitution passes of the wave-cooperative search (bc_kernel.h tier_single_n / nearest)
        c["scheme"] = CRISPR_SCHEME
        c["samples"] = None
        c["counted"] = [readgen.make_set(rng, 700, 20, 2)]
        c["reads"] = readgen.gen_reads(rng, CRISPR_SCHEME, n, 100, None, c["counted"], p_sub=0.04, p_n=0.04)
    elif name == "fmtn":
        # scheme with N positions: [AGCT] in the regex, wildcard in repair, regions_string shift (Q9)
        c["scheme"] = FMTN_SCHEME
        s = readgen.make_set(rng, 3, 6, 2)
        c["samples"] = {x: "S%d" % i for i, x in enumerate(s)}
        c["counted"] = [readgen.make_set(rng, 20, 7, 2), readgen.make_set(rng, 10, 5, 2)]
        c["kwargs"] = dict(min_quality=22.0)
        c["reads"] = readgen.gen_reads(rng, FMTN_SCHEME, n, 64, s, c["counted"], p_sub=0.03, p_n=0.02, var_len=True)
    elif name == "nosample_with_sample_file":
        # sample file given but no [n] in the scheme: counts are discarded yet matched (Q11)
        c["scheme"] = NOSAMPLE_SCHEME
        c["samples"] = {"ACGTACGT": "X"}
        c["counted"] = [readgen.make_set(rng, 30, 9, 2), readgen.make_set(rng, 8, 4, 2)]
        c["reads"] = readgen.gen_reads(rng, NOSAMPLE_SCHEME, n, 50, None, c["counted"], p_sub=0.02, p_n=0.003)
    elif name == "nosample":
        c["scheme"] = NOSAMPLE_SCHEME
        c["samples"] = None
        c["counted"] = [readgen.make_set(rng, 30, 9, 2), readgen.make_set(rng, 8, 4, 2)]
        c["kwargs"] = dict(min_quality=30.5)
        c["reads"] = readgen.gen_reads(rng, NOSAMPLE_SCHEME, n, 50, None, c["counted"], p_sub=0.02, p_n=0.003)
    elif name == "refs_with_n_and_ragged":
        # references containing N and of mixed lengths (Q6, Q7)
        c["scheme"] = DEL_SCHEME
        c["samples"] = {"ACGTACGT": "a", "ACGTACGN": "b", "TTTT": "c", "GGGGGGGGGG": "d"}
        c["counted"] = [["ACGTACGT", "ACGTACGA", "ACNTACGT", "CCCCCC", "GGGGGGGGGGG"],
                        readgen.make_set(rng, 20, 8, 2) + ["NNNNNNNN"][:0],
                        readgen.make_set(rng, 20, 8, 2) + ["ACGTNNNN"]]
        c["reads"] = readgen.gen_reads(rng, DEL_SCHEME, n, 90, list(c["samples"]), c["counted"], p_sub=0.03,
                                       p_n=0.02)
    elif name == "other_chars":
        # bytes outside ACGTN in reads: mismatch everywhere, never wildcard
        c["scheme"] = DEL_SCHEME
        s = readgen.make_set(rng, 4, 8, 3)
        c["samples"] = {x: "S%d" % i for i, x in enumerate(s)}
        c["counted"] = [readgen.make_set(rng, 40, 8, 2) for _ in range(3)]
        c["reads"] = readgen.gen_reads(rng, DEL_SCHEME, n, 100, s, c["counted"], p_sub=0.01, p_n=0.002,
                                       p_other=0.01)
    elif name == "example_files_random_nosample":
        # random barcode, sample FILE given but no [n] group: the counts go under "barcode" (info.rs:792-801)
        c["scheme"] = "AGCTACGAATCG{6}TGGA{6}TGGA{6}ACTAGAT(8)TAGA"
        c["samples"] = {"AGCATAC": "Sample_name_1"}
        c["counted"] = [["CAGAGA", "TGATTG"], ["ATGAAA", "GCGCCA"], ["GATAGC", "TTAGCT"]]
        c["reads"] = readgen.gen_reads(rng, c["scheme"], n, 70, None, c["counted"], p_sub=0.01, p_n=0.004,
                                       dup_frac=0.3)
    elif name == "raw_counted":
        # Barcode-seq style: no counted-barcode file, captures are the keys (README.md:149-152)
        c["scheme"] = NOSAMPLE_SCHEME
        c["samples"] = None
        c["counted"] = None
        c["kwargs"] = dict(min_quality=25.0)
        pool = [readgen.make_set(rng, 12, 9, 2), readgen.make_set(rng, 5, 4, 2)]
        c["reads"] = readgen.gen_reads(rng, NOSAMPLE_SCHEME, n, 50, None, pool, p_sub=0.02, p_n=0.01)
    elif name == "raw_sample":
        # sample barcode in the scheme but no sample file: sample captures become keys (info.rs:742-757)
        c["scheme"] = DEL_SCHEME
        c["samples"] = None
        c["counted"] = [readgen.make_set(rng, 20, 8, 2) for _ in range(3)]
        pool = readgen.make_set(rng, 6, 8, 3)
        c["reads"] = readgen.gen_reads(rng, DEL_SCHEME, n, 100, pool, c["counted"], p_sub=0.02, p_n=0.004)
    elif name == "raw_all_random":
        c["scheme"] = "[5]ACGTTGCA{6}GGATC(7)TTGACA"
        c["samples"] = None
        c["counted"] = None
        pool = [readgen.make_set(rng, 6, 6, 2)]
        c["reads"] = readgen.gen_reads(rng, c["scheme"], n, 60, readgen.make_set(rng, 3, 5, 2), pool, p_sub=0.01,
                                       p_n=0.01, dup_frac=0.4)
    elif name == "long_gaps":
        c["scheme"] = GAP_SCHEME
        s = readgen.make_set(rng, 3, 32, 6)
        c["samples"] = {x: "S%d" % i for i, x in enumerate(s)}
        c["counted"] = [readgen.make_set(rng, 12, 32, 6), readgen.make_set(rng, 30, 8, 2)]
        c["kwargs"] = dict(min_quality=21.0, max_constant=2)
        c["reads"] = readgen.gen_reads(rng, GAP_SCHEME, n, 120, s, c["counted"], p_sub=0.01, p_n=0.002)
    elif name == "sample_only_large":
        # a large SAMPLE set and no counted barcode: the one-group plan whose search verdicts are queued per wavefront
        # (bc_kernel.h), with SampleBarcode -- not Barcode -- as the failure's outcome (parse.rs:132-140)
        c["scheme"] = "TTGTGGAAAGGACGAAACACCG[16]GTTTTAGAGCTAGAAATAGCAAGTT"
        s = readgen.make_set(rng, 300, 16, 3)
        c["samples"] = {x: "S%d" % i for i, x in enumerate(s)}
        c["counted"] = None
        c["reads"] = readgen.gen_reads(rng, c["scheme"], n, 100, s, None, p_sub=0.05, p_n=0.004)
    else:
        raise KeyError(name)
    return c


ALL_CASES = ["del_exact", "del_mismatch_quality", "del_dense_ties", "del_random", "example_files",
             "example_files_samples", "crispr", "large_set_ties", "large_set_many_n", "fmtn", "nosample_with_sample_file", "nosample",
             "refs_with_n_and_ragged", "other_chars", "long_gaps", "example_files_random_nosample", "raw_counted",
             "raw_sample", "raw_all_random", "sample_only_large"]

RANDOM_CASES = ["del_random", "example_files", "example_files_samples", "example_files_random_nosample",
                "raw_all_random"]
NO_RANDOM_CASES = [c for c in ALL_CASES if c not in RANDOM_CASES]
RANDOM_ENGINE_CASES = list(RANDOM_CASES) + ["rnd_rb_%d" % i for i in range(8)]


def with_wrapping_quality(c, seed=0, frac=0.3):
    """quality bytes below '!' (33) in a share of the reads: `ch as u8 - 33` wraps in the reference's release build
    (parse.rs:326), so such a byte scores 223..255 -- the engine's two-v_sad_u8 fast path must notice and defer to the
    wrapping form"""
    import numpy as np
    rng = np.random.default_rng(1000 + seed)
    out = []
    for s, q in c["reads"]:
        if rng.random() < frac and len(q) > 4:
            q = list(q)
            for _ in range(int(rng.integers(1, 6))):
                q[int(rng.integers(0, len(q)))] = chr(int(rng.integers(1, 33)))
            q = "".join(q)
        out.append((s, q))
    c["reads"] = out
    return c


def random_case(seed, n=300, with_random=False):
    """A randomly drawn scheme (constants with the odd N, optional sample group, 1-4 counted groups of 3-20
    bases), random set sizes / budgets / quality threshold, conversion files present or not: the shapes the
    hand-written cases above do not reach.  No random barcode (set semantics are covered there)."""
    rng = np.random.default_rng(90000 + seed)
    c = dict(name="random_%d" % seed, kwargs={})

    def const(lo, hi):
        k = int(rng.integers(lo, hi + 1))
        t = readgen.rand_seq(rng, k)
        if rng.random() < 0.25 and k >= 5:  # a scheme N or two inside the constant
            t = list(t)
            for _ in range(int(rng.integers(1, 3))):
                t[int(rng.integers(1, k - 1))] = "N"
            t = "".join(t)
        return t

    parts, sample_len, group_lens = [], None, []
    if rng.random() < 0.3:
        parts.append(const(3, 12))
    if rng.random() < 0.6:
        sample_len = int(rng.integers(4, 11))
        parts.append("[%d]" % sample_len)
    parts.append(const(4, 20))
    for g in range(int(rng.integers(1, 5))):
        r = rng.random()
        k = int(rng.integers(3, 11)) if r < 0.75 else (int(rng.integers(11, 16)) if r < 0.9 else int(rng.integers(16, 21)))
        group_lens.append(k)
        parts.append("{%d}" % k)
        parts.append(const(2, 14))
    random_len = 0
    if with_random:  # a random (UMI) barcode somewhere behind the first counted group
        random_len = int(rng.integers(5, 13))
        parts.insert(int(rng.integers(parts.index("{%d}" % group_lens[0]) + 1, len(parts) + 1)), "(%d)" % random_len)
        if parts[-1].startswith("("):
            parts.append(const(2, 6))
    scheme = "".join(parts)
    if rng.random() < 0.3:
        scheme = scheme.replace("{", "\n{").replace("}", "}\n")  # the multi-line form of the format file
    c["scheme"] = scheme
    L = len(scheme.replace("\n", "").replace("[", "").replace("]", "").replace("{", "").replace("}", ""))
    L = sum(v if k != "C" else len(v) for k, v in readgen.scheme_layout(scheme))
    # known sets, small enough that the dense table stays below ~4 M entries
    budget = 4_000_000
    samples = None
    if sample_len is not None:
        ns = int(rng.integers(2, 9))
        samples = readgen.make_set(rng, ns, sample_len, int(rng.integers(1, 4)) if sample_len >= 6 else 1)
        budget //= ns
    counted = []
    for k in group_lens:
        cap = max(2, min(4 ** k // 4, int(budget ** (1.0 / max(len(group_lens) - len(counted), 1)))))
        hi = min(cap, 300 if k >= 16 else 120)
        m = int(rng.integers(2, hi + 1))
        counted.append(readgen.make_set(rng, m, k, 1 if k < 5 else int(rng.integers(1, 3))))
        budget = max(budget // m, 2)
    have_sample_file = samples is not None and rng.random() < 0.8
    have_counted_file = rng.random() < 0.8
    # raw captures must fit a 64-bit mixed-radix key: keep it dense unless the groups are short
    raw_bits = (0 if have_sample_file or samples is None else sample_len) + (0 if have_counted_file else sum(group_lens))
    if raw_bits > 24 or with_random:  # (raw captures + random barcode: the fixed case raw_all_random)
        have_sample_file, have_counted_file = samples is not None, True
    c["samples"] = {x: "S%d" % i for i, x in enumerate(samples)} if have_sample_file else None
    c["counted"] = counted if have_counted_file else None
    kw = {}
    if rng.random() < 0.4:
        kw["max_barcode"] = int(rng.integers(0, 3))
    if rng.random() < 0.4 and samples is not None:
        kw["max_sample"] = int(rng.integers(0, 3))
    if rng.random() < 0.4:
        kw["max_constant"] = int(rng.integers(0, 6))
    if rng.random() < 0.5:
        kw["min_quality"] = float(rng.integers(10, 31)) + (0.5 if rng.random() < 0.3 else 0.0)
    c["kwargs"] = kw
    read_len = min(128, L + int(rng.integers(0, 45)))
    # a raw capture holding a byte outside ACGTN is the engine's one documented refusal (BC_UNSUPPORTED_READS;
    # the reference would count it under the literal string): such bytes only where every group has a file
    raw_mode = (samples is not None and not have_sample_file) or not have_counted_file
    p_other = 0.0 if (raw_mode or with_random) else float(rng.choice([0.0, 0.0, 0.002]))  # (the same goes for a random barcode)
    c["reads"] = readgen.gen_reads(rng, scheme, n, read_len, samples, counted, p_sub=float(rng.choice([0.0, 0.01, 0.03, 0.06])),
                                   p_n=float(rng.choice([0.0, 0.003, 0.02])), p_other=p_other,
                                   var_len=bool(rng.random() < 0.5), dup_frac=0.35 if with_random else 0.0)
    return c
