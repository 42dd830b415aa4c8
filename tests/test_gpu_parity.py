"""GPU parity: the HIP engine, called through the C ABI, against the CPU oracle.
Bit-exact bar: six outcome counters, every per-read outcome, every (sample, tuple, count) row."""
import json
import os

import numpy as np
import pytest

import cases
import parity
import readgen

pytestmark = pytest.mark.gpu

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
NO_RANDOM = cases.NO_RANDOM_CASES


def _pkg():
    import ngs_barcode_count_amd as pkg
    return pkg


def make_plan(case):
    pkg = _pkg()
    p = pkg.Plan(case["scheme"])
    if case.get("samples"):
        for s, i in case["samples"].items():
            p.add_sample(s, i)
    if case.get("counted"):
        for bi, refs in enumerate(case["counted"]):
            for r in refs:
                p.add_counted(bi, r, r)
    kw = case.get("kwargs", {})
    p.set_max_errors(kw.get("max_sample"), kw.get("max_barcode"), kw.get("max_constant"))
    p.set_min_quality(kw.get("min_quality", 0.0))
    return p


def run_device(plan, seq, qual, lens, stride, read_len):
    """-> (engine, per-read outcomes, per-read dense index)"""
    import torch
    pkg = _pkg()
    n = seq.size // stride
    dseq = torch.from_numpy(seq.reshape(-1)).cuda()
    dqual = torch.from_numpy(qual.reshape(-1)).cuda() if qual is not None else None
    dlens = torch.from_numpy(lens.view(np.int16)).cuda() if lens is not None else None
    outc = torch.zeros(n, dtype=torch.uint8, device="cuda")
    idx = torch.zeros(n, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    eng = pkg.Engine(plan, device=0)
    eng.trace(outc.data_ptr(), idx.data_ptr())
    eng.submit_device(dseq.data_ptr(), dqual.data_ptr() if dqual is not None else None, n, stride, read_len,
                      dlens.data_ptr() if dlens is not None else None)
    eng.sync()
    return eng, outc.cpu().numpy(), idx.cpu().numpy().astype(np.uint64)


@pytest.mark.parametrize("k", KAT["fix_error"], ids=[k["id"] for k in KAT["fix_error"]])
def test_fix_error_kat_on_device(k):
    """the reference's fix_error doctest values (src/parse.rs:540-551) + Appendix B, on the GPU"""
    pkg = _pkg()
    assert pkg.fix_error(k["query"], k["set"], k["max"]) == k["expect"]
    assert pkg.fix_error(k["query"], k["set"][::-1], k["max"]) == k["expect"]


@pytest.fixture
def kernel(request, monkeypatch, tmp_path_factory):
    """generic: the kernel that reads the plan from memory; specialised: the one compiled for the plan's
    scheme (forced here -- by default it only serves batches of 2^20 reads or more)"""
    monkeypatch.setenv("BC_JIT", "force" if request.param == "specialised" else "0")
    monkeypatch.setenv("BC_JIT_CACHE", str(tmp_path_factory.getbasetemp() / "jit_cache"))
    return request.param


def check_kernel(eng, kernel):
    name = eng.kernel_name()
    nw, nww = (int(v) for v in name[name.index("<") + 1:-1].split(","))
    if kernel == "specialised" and nw <= 8 and nww <= 4:
        assert name.startswith("bc_jit_match_count"), name
    else:
        assert name.startswith("match_count_kernel"), name


@pytest.mark.parametrize("name", NO_RANDOM)
@pytest.mark.parametrize("use_lens", [False, True])
@pytest.mark.parametrize("kernel", ["generic", "specialised"], indirect=True)
def test_engine_vs_oracle(name, use_lens, kernel):
    c = cases.build_case(name, seed=11 + use_lens, n=3000)
    if not use_lens:
        rl = min(len(s) for s, _ in c["reads"])
        c["reads"] = [(s[:rl], q[:rl]) for s, q in c["reads"]]
    plan = make_plan(c)
    seq, qual, lens = readgen.to_arrays(c["reads"])
    stride = seq.shape[1]
    eng, outc, idx = run_device(plan, seq, qual, lens if use_lens else None, stride, stride)
    discard = (not plan.sample_barcode) and len(plan.samples()) > 0
    o = parity.check_per_read(c, plan, outc, idx, discard)
    got = eng.counters()
    for k, v in o.counters.items():
        assert got[k] == v, (k, got, o.counters)
    assert got["total_reads"] == len(c["reads"]) and got["unsupported_reads"] == 0
    assert eng.result_rows() == o.rows()
    check_kernel(eng, kernel)
    eng.close()


@pytest.mark.parametrize("name", cases.RANDOM_ENGINE_CASES)
@pytest.mark.parametrize("kernel", ["generic", "specialised"], indirect=True)
def test_random_barcode_engine_vs_oracle(name, kernel):
    """PCR-duplicate collapse on the device hash set (info.rs:770-802).  Which copy of a duplicated
    molecule is the "matched" one depends on scheduling, so per read matched/duplicate are one
    class; the counters and the (sample, tuple, distinct count) rows are compared exactly."""
    c = cases.build_case(name, seed=21, n=4000)
    plan = make_plan(c)
    seq, qual, lens = readgen.to_arrays(c["reads"])
    stride = seq.shape[1]
    eng, outc, keys = run_device(plan, seq, qual, lens, stride, stride)
    o = parity.oracle_for(c)
    exp = [o.process(s, q) for s, q in c["reads"]]
    fold = lambda v: 0 if v == parity.CODE["duplicates"] else v
    for i, e in enumerate(exp):
        assert fold(int(outc[i])) == fold(parity.CODE[e]), (i, e, int(outc[i]))
    got = eng.counters()
    for k, v in o.counters.items():
        assert got[k] == v, (k, got, o.counters)
    assert eng.key_count() == o.counters["matched"]
    assert eng.result_rows() == o.rows()
    check_kernel(eng, kernel)
    # the same reads again: every one of them is now a duplicate
    eng2, _, _ = run_device(plan, np.concatenate([seq, seq]), np.concatenate([qual, qual]),
                            np.concatenate([lens, lens]), stride, stride)
    g2 = eng2.counters()
    assert g2["matched"] == o.counters["matched"]
    assert g2["duplicates"] == 2 * o.counters["duplicates"] + o.counters["matched"]
    assert eng2.result_rows() == o.rows()
    eng.close()
    eng2.close()


@pytest.mark.parametrize("seed", range(16))
@pytest.mark.parametrize("kernel", ["generic", "specialised"], indirect=True)
def test_randomly_drawn_schemes(seed, kernel):
    """schemes, sets, budgets and thresholds drawn at random (cases.random_case), both kernels vs the oracle"""
    c = cases.random_case(seed, n=2000)
    plan = make_plan(c)
    seq, qual, lens = readgen.to_arrays(c["reads"])
    stride = seq.shape[1]
    eng, outc, idx = run_device(plan, seq, qual, lens, stride, stride)
    discard = (not plan.sample_barcode) and len(plan.samples()) > 0
    o = parity.check_per_read(c, plan, outc, idx, discard)
    got = eng.counters()
    for k, v in o.counters.items():
        assert got[k] == v, (k, got, o.counters)
    assert eng.result_rows() == o.rows()
    check_kernel(eng, kernel)
    eng.close()


@pytest.mark.parametrize("name", ["raw_counted", "raw_sample"])
def test_count_map_export_import_merges_shards(name):
    """raw-key plans across GPUs: two engines count one half of the reads each; adding the second one's
    (key, count) pairs into the first reproduces the single-engine result (bc_engine_export_counts /
    bc_engine_import_counts, the primitives of distributed.finish_sparse)"""
    import torch
    c = cases.build_case(name, seed=31, n=3000)
    plan = make_plan(c)
    assert plan.mode == "sparse" and not plan.random_barcode
    seq, qual, lens = readgen.to_arrays(c["reads"])
    stride, half = seq.shape[1], seq.shape[0] // 2
    a, _, _ = run_device(plan, seq[:half], qual[:half], lens[:half], stride, stride)
    b, _, _ = run_device(plan, seq[half:], qual[half:], lens[half:], stride, stride)
    n = b.export_counts(None, None, 0)
    keys = torch.zeros(max(n, 1), dtype=torch.int64, device="cuda")
    cnts = torch.zeros(max(n, 1), dtype=torch.int32, device="cuda")
    assert b.export_counts(keys.data_ptr(), cnts.data_ptr(), n) == n
    assert int(cnts[:n].sum()) == b.counters()["matched"]
    a.import_counts(keys.data_ptr(), cnts.data_ptr(), n)
    o = parity.oracle_for(c)
    for sq, ql in c["reads"]:
        o.process(sq, ql)
    assert a.result_rows() == o.rows()
    # clearing leaves an empty map
    b.clear_keys()
    assert b.export_counts(None, None, 0) == 0 and b.result_rows() == []
    a.close()
    b.close()


def test_key_export_import_roundtrip():
    """the exchange primitives of the multi-GPU random-barcode path: export -> import into a fresh
    engine reproduces the set; importing twice adds nothing"""
    import torch
    pkg = _pkg()
    c = cases.build_case("del_random", seed=22, n=3000)
    plan = make_plan(c)
    seq, qual, lens = readgen.to_arrays(c["reads"])
    eng, _, _ = run_device(plan, seq, qual, lens, seq.shape[1], seq.shape[1])
    n = eng.key_count()
    buf = torch.zeros(n, dtype=torch.int64, device="cuda")
    assert eng.export_keys(buf.data_ptr(), n) == n
    assert len(torch.unique(buf)) == n
    e2 = pkg.Engine(plan, device=0)
    assert e2.import_keys(buf.data_ptr(), n) == n
    assert e2.import_keys(buf.data_ptr(), n) == 0
    assert e2.key_count() == n and e2.result_rows() == eng.result_rows()
    e2.clear_keys()
    assert e2.key_count() == 0 and e2.result_rows() == []
    eng.close()
    e2.close()


def test_random_barcode_multi_rank_flow_with_real_engines():
    """the documented multi-GPU flow of a dense plan with a random barcode -- key exchange, then a sum
    of tables, then finish on the root -- driven through REAL engines: two engines stand for two ranks (each counted
    its own shard), keys go to their owner (distributed.key_owner), every rank materializes the distinct counts of the
    keys it owns, the tables are added onto rank 0 and rank 0's finish compacts that sum as it stands.  Rows and
    counters must equal the oracle's over all reads."""
    import torch
    pkg = _pkg()
    from ngs_barcode_count_amd import distributed as bcdist
    c = cases.build_case("del_random", seed=41, n=6000)
    plan = make_plan(c)
    assert plan.mode == "dense"
    seq, qual, lens = readgen.to_arrays(c["reads"])
    stride = seq.shape[1]
    half = len(c["reads"]) // 2
    world = 2
    tables = [torch.zeros(plan.table_entries, dtype=torch.int32, device="cuda") for _ in range(world)]
    torch.cuda.synchronize()
    engs, local = [], []
    for r in range(world):
        a, b = (0, half) if r == 0 else (half, len(c["reads"]))
        e = pkg.Engine(plan, device=0, table_ptr=tables[r].data_ptr())
        e.submit_host(seq[a:b].reshape(-1), qual[a:b].reshape(-1), stride, stride, lens[a:b])
        engs.append(e)
        local.append(e.counters())
    # the exchange: every key to its owner rank
    outbox = [[None] * world for _ in range(world)]
    for r, e in enumerate(engs):
        n = e.key_count()
        buf = torch.zeros(max(n, 1), dtype=torch.int64, device="cuda")
        e.export_keys(buf.data_ptr(), n)
        owner = bcdist.key_owner(buf[:n], world)
        for d in range(world):
            outbox[r][d] = buf[:n][owner == d].contiguous()
    total = dict.fromkeys(pkg.COUNTER_NAMES, 0)
    for r, e in enumerate(engs):
        recv = torch.cat([outbox[src][r] for src in range(world)])
        torch.cuda.synchronize()
        e.clear_keys()
        owned = e.import_keys(recv.data_ptr(), recv.numel()) if recv.numel() else 0
        fixed = dict(local[r])
        fixed["duplicates"] = local[r]["duplicates"] + local[r]["matched"] - owned
        fixed["matched"] = owned
        for k in total:
            total[k] += fixed[k]
        e.materialize_table()
    # a sum of zero tables (what the flow did before materialize existed) would give no rows at all
    assert int(tables[1].sum()) > 0
    tables[0] += tables[1]
    torch.cuda.synchronize()
    o = parity.oracle_for(c)
    for sq, ql in c["reads"]:
        o.process(sq, ql)
    assert engs[0].result_rows() == o.rows()
    assert {k: total[k] for k in o.counters} == o.counters and o.counters["duplicates"] > 0
    # a later submit invalidates the materialized table: finish then rebuilds it from the engine's own keys
    engs[1].submit_host(seq[:10].reshape(-1), qual[:10].reshape(-1), stride, stride, lens[:10])
    assert sum(cnt for _, _, cnt in engs[1].result_rows()) == engs[1].key_count()
    for e in engs:
        e.close()


@pytest.mark.parametrize("kernel", ["generic", "specialised"], indirect=True)
def test_quality_bytes_below_33_wrap_like_the_reference(kernel):
    """`ch as u8 - 33` wraps for bytes below '!' (parse.rs:326): the kernel's fast quality sums must defer to the
    wrapping form whenever such a byte is in a run"""
    c = cases.with_wrapping_quality(cases.build_case("del_mismatch_quality", seed=23, n=3000), seed=2)
    plan = make_plan(c)
    seq, qual, lens = readgen.to_arrays(c["reads"])
    eng, outc, idx = run_device(plan, seq, qual, lens, seq.shape[1], seq.shape[1])
    o = parity.check_per_read(c, plan, outc, idx, False)
    got = eng.counters()
    assert {k: got[k] for k in o.counters} == o.counters and o.counters["low_quality"] > 0
    check_kernel(eng, kernel)
    eng.close()


def test_release_library_ignores_experiment_switches(monkeypatch, tmp_path):
    """an environment variable must not be able to make the counting tool miscount.  BC_ABLATE (phases
    skipped), BC_LHASH / BC_PIPE (kernel variants) are honoured only by -DBC_EXPERIMENT builds."""
    monkeypatch.setenv("BC_ABLATE", "0xffff")
    monkeypatch.setenv("BC_LHASH", "0")
    monkeypatch.setenv("BC_PIPE", "0")
    monkeypatch.setenv("BC_JIT_FLAGS", "-DBC_MIN_WAVES=1")
    monkeypatch.setenv("BC_JIT", "force")
    monkeypatch.setenv("BC_JIT_CACHE", str(tmp_path / "jit_cache"))
    c = cases.build_case("del_mismatch_quality", seed=5, n=4000)
    plan = make_plan(c)
    seq, qual, lens = readgen.to_arrays(c["reads"])
    eng, outc, idx = run_device(plan, seq, qual, lens, seq.shape[1], seq.shape[1])
    assert eng.kernel_name().startswith("bc_jit_match_count")
    o = parity.oracle_for(c)
    for sq, ql in c["reads"]:
        o.process(sq, ql)
    got = eng.counters()
    assert {k: got[k] for k in o.counters} == o.counters
    assert eng.result_rows() == o.rows()
    eng.close()


def test_kat_reads_on_device():
    """Appendix B K1-K7b (anchor, exclusive last window, N-free repair, quality offset after repair),
    the example scheme verbatim (with its random barcode), known sets holding the KAT captures"""
    scheme = KAT["scheme"]
    for r in KAT["reads"]:
        c = dict(scheme=scheme, samples={"AAAAAAAAAA": "s"}, counted=[["CAGAGA"], ["ATGAAA"], ["GATAGC"]],
                 kwargs=dict(min_quality=r["min_quality"]), reads=[(r["seq"], r["qual"])])
        plan = make_plan(c)
        seq, qual, lens = readgen.to_arrays(c["reads"])
        eng, outc, idx = run_device(plan, seq, qual, None, seq.shape[1], seq.shape[1])
        o = parity.check_per_read(c, plan, outc, idx, False, rspace=5 ** 8)
        assert parity.CODE[r["outcome"]] == int(outc[0]), r["id"]
        eng.close()


@pytest.mark.parametrize("kernel", ["generic", "specialised"], indirect=True)
def test_long_reads_and_odd_strides(kernel):
    for rl, stride in ((150, 152), (250, 251), (300, 304), (75, 77)):
        c = cases.build_case("del_mismatch_quality", seed=rl, n=10)
        rng = np.random.default_rng(rl)
        c["reads"] = readgen.gen_reads(rng, c["scheme"], 700, rl, list(c["samples"]), c["counted"], p_sub=0.02,
                                       p_n=0.004)
        plan = make_plan(c)
        seq, qual, lens = readgen.to_arrays(c["reads"], stride=stride)
        eng, outc, idx = run_device(plan, seq, qual, None, stride, rl)
        parity.check_per_read(c, plan, outc, idx, False)
        check_kernel(eng, kernel)
        eng.close()


def test_submit_host_equals_submit_device():
    pkg = _pkg()
    c = cases.build_case("del_mismatch_quality", seed=5, n=5000)
    rl = min(len(s) for s, _ in c["reads"])
    c["reads"] = [(s[:rl], q[:rl]) for s, q in c["reads"]]
    plan = make_plan(c)
    seq, qual, lens = readgen.to_arrays(c["reads"])
    eng, outc, idx = run_device(plan, seq, qual, None, rl, rl)
    e2 = pkg.Engine(plan, device=0)
    e2.submit_host(seq.reshape(-1), qual.reshape(-1), rl, rl)
    assert e2.counters() == eng.counters()
    assert e2.result_rows() == eng.result_rows()
    e2.reset()
    assert sum(e2.counters().values()) == 0 and e2.result_rows() == []
    eng.close()
    e2.close()


def test_errors_are_loud():
    pkg = _pkg()
    with pytest.raises(pkg.BarcodeCountError):  # a token mixing 'N' and 'n' whose repaired reads could match at a shifted offset
        pkg.Engine(pkg.Plan("nN{8}"), device=0)
    with pytest.raises(pkg.BarcodeCountError):  # raw captures beyond even the widest key (448 payload bits)
        pkg.Engine(pkg.Plan("[60]ACGT{60}TT{60}"), device=0)
    p = make_plan(dict(scheme="ACGTACGT{8}TTGG", counted=[["ACGTACGT"]], kwargs=dict(min_quality=10.0)))
    e = pkg.Engine(p, device=0)
    import torch
    d = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    with pytest.raises(pkg.BarcodeCountError):  # quality filter on, no quality buffer
        e.submit_device(d.data_ptr(), None, 4, 50, 50)
    with pytest.raises(pkg.BarcodeCountError):  # no such device
        pkg.Engine(p, device=99)
    e.close()


def test_table_pack_kernel_matches_its_torch_form():
    """bc_table_pack_u8 (what dense tables travel as between GPUs) against the torch statement of it"""
    import torch
    from ngs_barcode_count_amd import distributed as bcdist
    g = torch.Generator(device="cpu").manual_seed(5)
    for n in (1, 3, 4, 1001, 1 << 20):
        small = torch.randint(0, 256, (n,), generator=g, dtype=torch.int32)
        large = torch.randint(-2**31, 2**31 - 1, (n,), generator=g, dtype=torch.int32)
        t = torch.where(torch.rand(n, generator=g) < 0.97, small, large)
        b_cpu, i_cpu, v_cpu = bcdist.pack_table(t)
        b_gpu, i_gpu, v_gpu = bcdist.pack_table(t.cuda())
        assert torch.equal(b_cpu, b_gpu.cpu())
        order = torch.argsort(i_gpu.cpu())  # the kernel appends in no particular order
        assert torch.equal(i_cpu, i_gpu.cpu()[order]) and torch.equal(v_cpu, v_gpu.cpu()[order])
        if n % 4 == 0 and n >= 8:
            w = 4 if n % 16 == 0 else 1
            assert torch.equal(bcdist.sum_slices(b_gpu, w, n // w, torch.int32).cpu(),
                               torch.sum(b_cpu.view(w, n // w), dim=0, dtype=torch.int32))
        # and unpacking restores the table
        back = b_gpu.to(torch.int32)
        back.index_add_(0, i_gpu, v_gpu)
        assert torch.equal(back.cpu(), t)


@pytest.mark.parametrize("kernel", ["generic", "specialised"], indirect=True)
def test_exhaustive_small_domain(kernel):
    """tests/exhaustive.py on the GPU: EVERY read of up to 8 bases over {A,C,G,N} with EVERY quality string over two
    symbols (19.2 M pairs), plus every 9-base read with eight quality strings -- each read's outcome and table index
    against the oracle, through both kernels.  A proof over that domain, not a sample."""
    import torch
    import exhaustive
    pkg = _pkg()
    c = exhaustive.case()
    plan = make_plan(c)
    STRIDE = 12
    eng = pkg.Engine(plan, device=0)
    total = 0

    def check(seq, qual, ln):
        nonlocal total
        n = seq.shape[0]
        for a in range(0, n, 1 << 22):
            b = min(n, a + (1 << 22))
            s = np.ascontiguousarray(seq[a:b]).reshape(-1)
            q = np.ascontiguousarray(qual[a:b]).reshape(-1)
            lens = np.full(b - a, ln, dtype=np.uint16)
            o = parity.oracle_for(c)
            exp = o.process_batch_outcomes(s, q, STRIDE, STRIDE, lens=lens)
            ds, dq = torch.from_numpy(s).cuda(), torch.from_numpy(q).cuda()
            dl = torch.from_numpy(lens.view(np.int16)).cuda()
            outc = torch.zeros(b - a, dtype=torch.uint8, device="cuda")
            idx = torch.zeros(b - a, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            eng.reset()
            eng.trace(outc.data_ptr(), idx.data_ptr())
            eng.submit_device(ds.data_ptr(), dq.data_ptr(), b - a, STRIDE, STRIDE, dl.data_ptr())
            eng.sync()
            got = outc.cpu().numpy()
            bad = np.nonzero(got != exp)[0]
            assert bad.size == 0, (ln, bytes(seq[a + bad[0], :ln]), bytes(qual[a + bad[0], :ln]), int(got[bad[0]]),
                                   int(exp[bad[0]]))
            assert eng.result_rows() == o.rows(), ln
            eng.trace(None, None)
            total += b - a

    for ln in range(0, 9):
        seq, qual = exhaustive.domain(ln, STRIDE)
        check(seq, qual, ln)
    # nine bases: every read, eight quality strings (bit patterns of the quality index)
    ln = 9
    r = np.arange(4 ** ln, dtype=np.int64)
    for qbits in (0, 0x1FF, 0x155, 0x0AA, 0x033, 0x1C7, 0x00F, 0x1F0):
        seq = np.full((r.size, STRIDE), ord("N"), dtype=np.uint8)
        qual = np.full((r.size, STRIDE), ord("!"), dtype=np.uint8)
        for p in range(ln):
            seq[:, p] = exhaustive.LETTERS[(r >> (2 * p)) & 3]
            qual[:, p] = exhaustive.QUALS[(qbits >> p) & 1]
        check(seq, qual, ln)
    assert total == sum(8 ** k for k in range(9)) + 8 * 4 ** 9
    check_kernel(eng, kernel)
    eng.close()


@pytest.mark.parametrize("kernel", ["generic", "specialised"], indirect=True)
@pytest.mark.parametrize("scheme,long_only", [("[8]AGCTacgaATCG{8}TGGA{8}tgga{8}ACTAGAT", False),
                                              ("[8]AGCTACNnNGAATCG{8}TGGA{8}TGGA{8}ACTAGAT", False),
                                              ("[8]AGCTACGAATCG{8}TGnnGA{8}TGGA{8}ACTAGAT", True)])
def test_lower_case_scheme_letters(kernel, scheme, long_only):
    """lower-case constants anchor like upper-case ones and make every repair fail (info.rs:298-299, parse.rs:270-283);
    lower-case n's are LITERAL 'N' bases to the regex (the token pattern is case-insensitive, contains('N') is not):
    those schemes run on the wave-per-read kernel"""
    c = cases.build_case("del_mismatch_quality", seed=72, n=10)
    rng = np.random.default_rng(72)
    gen_scheme = scheme.upper().replace("NN", "AC") if long_only else scheme.upper()
    if "NnN" in scheme:  # (the token mixing 'N' and 'n': the regex takes two free bases there, info.rs:287-295)
        gen_scheme = scheme.replace("NnN", "NN")
    c["reads"] = readgen.gen_reads(rng, gen_scheme, 1500, 100, list(c["samples"]), c["counted"], p_sub=0.01, p_n=0.004)
    if long_only:  # put the literal N's the regex wants into two thirds of the reads
        at = scheme.index("nn") - 3 + 8 - len("[8]") + 3  # offset of the n's inside a match: "[8]" stands for 8 bases
        lay_off = 8 + len("AGCTACGAATCG") + 8 + 2
        reads = []
        for i, (s, q) in enumerate(c["reads"]):
            k = s.find("TGACGA")
            if i % 3 and k >= 0:
                s = s[:k + 2] + "NN" + s[k + 4:]
            reads.append((s, q))
        c["reads"] = reads
    c["scheme"] = scheme
    plan = make_plan(c)
    seq, qual, lens = readgen.to_arrays(c["reads"])
    eng, outc, idx = run_device(plan, seq, qual, lens, seq.shape[1], seq.shape[1])
    o = parity.check_per_read(c, plan, outc, idx, False)
    got = eng.counters()
    assert {k: got[k] for k in o.counters} == o.counters and eng.result_rows() == o.rows()
    assert o.counters["matched"] > 100 and o.counters["constant_region"] > 0
    if long_only:
        assert eng.kernel_name() == "long_match_kernel"
    else:
        check_kernel(eng, kernel)
    eng.close()


LONG_SCHEME = "[8]ACGTTGCA{40}GGATCC{36}TTGACA"


def _long_cases():
    """inputs beyond the lane-per-read kernel's widths (refused in round 1): reads above 320 bases, barcode
    groups and known barcodes above 32 bases, more than 31 tolerated constant-region errors"""
    out = {}
    rng = np.random.default_rng(91)
    c = cases.build_case("del_mismatch_quality", seed=61, n=10)
    c["reads"] = readgen.gen_reads(rng, c["scheme"], 900, 520, list(c["samples"]), c["counted"], p_sub=0.02, p_n=0.004)
    out["reads_of_520_bases"] = c
    c = cases.build_case("del_random", seed=62, n=10)
    c["reads"] = readgen.gen_reads(rng, c["scheme"], 900, 400, list(c["samples"]), c["counted"], p_sub=0.01, p_n=0.004,
                                   dup_frac=0.3)
    out["random_barcode_reads_of_400_bases"] = c
    c = cases.build_case("raw_counted", seed=63, n=10)
    pool = [readgen.make_set(rng, 12, 9, 2), readgen.make_set(rng, 5, 4, 2)]
    c["reads"] = readgen.gen_reads(rng, c["scheme"], 700, 350, None, pool, p_sub=0.02, p_n=0.01)
    out["raw_keys_reads_of_350_bases"] = c
    s = readgen.make_set(rng, 4, 8, 3)
    c = {"scheme": LONG_SCHEME, "samples": {x: "S%d" % i for i, x in enumerate(s)},
         "counted": [readgen.make_set(rng, 30, 40, 6) + ["ACGT" * 9], readgen.make_set(rng, 20, 36, 6)],
         "kwargs": dict(min_quality=18.0)}
    # (the 36-base entry of the 40-base group is a reference of another length: compared on the common prefix)
    c["reads"] = readgen.gen_reads(rng, LONG_SCHEME, 900, 150, s, [c["counted"][0][:-1], c["counted"][1]], p_sub=0.03,
                                   p_n=0.006)
    out["groups_of_40_and_36_bases"] = c
    c = cases.build_case("del_mismatch_quality", seed=64, n=900)
    c["kwargs"] = dict(c.get("kwargs", {}), max_constant=40)
    out["forty_constant_errors_allowed"] = c
    return out


@pytest.mark.parametrize("name", ["reads_of_520_bases", "random_barcode_reads_of_400_bases", "raw_keys_reads_of_350_bases",
                                  "groups_of_40_and_36_bases", "forty_constant_errors_allowed"])
@pytest.mark.parametrize("use_lens", [False, True])
def test_wave_per_read_kernel(name, use_lens):
    c = _long_cases()[name]
    if not use_lens:
        rl = min(len(s) for s, _ in c["reads"])
        c["reads"] = [(s[:rl], q[:rl]) for s, q in c["reads"]]
    plan = make_plan(c)
    seq, qual, lens = readgen.to_arrays(c["reads"])
    stride = seq.shape[1]
    eng, outc, idx = run_device(plan, seq, qual, lens if use_lens else None, stride, stride)
    assert eng.kernel_name() == "long_match_kernel"
    o = parity.oracle_for(c)
    exp = [o.process(s, q) for s, q in c["reads"]]
    if not plan.random_barcode:  # (which copy of a PCR duplicate counts as the duplicate depends on the order of arrival)
        bad = [i for i, e in enumerate(exp) if int(outc[i]) != parity.CODE[e]]
        assert not bad, (bad[:3], exp[bad[0]], int(outc[bad[0]]), c["reads"][bad[0]])
    got = eng.counters()
    assert {k: got[k] for k in o.counters} == o.counters, (got, o.counters)
    assert got["total_reads"] == len(c["reads"]) and got["unsupported_reads"] == 0
    assert eng.result_rows() == o.rows()
    assert o.counters["matched"] > 0
    eng.close()
