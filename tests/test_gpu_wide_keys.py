"""Captures kept raw (no conversion file: README.md "Barcode-seq", info.rs:742-757) and random barcodes that do not fit
the 64-bit tuple key -- above 27 bases, or several that overflow it together -- used to be refused; they
are now counted on the GPU under keys several u64 wide (csrc/bc_long.h).  Every case against the CPU oracle: per-read
outcomes, counters and every (sample, tuple, count) row, on one engine and as ranks of a multi-process job."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BASES = "ACGT"


def _reads(scheme_parts, n, seed, pools, p_sub=0.01, p_n=0.004, flank=(0, 12)):
    """reads for a scheme given as a list of ("const", text) / ("cap", length, pool index or None): captures are drawn
    from small pools (so that keys repeat) or fresh at random (pool None), then substitutions and N are sprinkled"""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        body = []
        for part in scheme_parts:
            if part[0] == "const":
                body.append(part[1])
            else:
                _, ln, pool = part
                if pool is None:
                    body.append("".join(rng.choice(list(BASES), ln)))
                else:
                    body.append(pools[pool][int(rng.integers(len(pools[pool])))])
        s = list("".join(body))
        for i in range(len(s)):
            r = rng.random()
            if r < p_sub:
                s[i] = BASES[int(rng.integers(4))]
            elif r < p_sub + p_n:
                s[i] = "N"
        pre = "".join(rng.choice(list(BASES), int(rng.integers(flank[0], flank[1] + 1))))
        post = "".join(rng.choice(list(BASES), int(rng.integers(1, 9))))
        out.append(pre + "".join(s) + post)
    return out


def _pool(rng, k, ln):
    return ["".join(rng.choice(list(BASES), ln)) for _ in range(k)]


def _run_engine(plan, reads, trace=True):
    import torch
    import ngs_barcode_count_amd as pkg
    stride = (max(len(r) for r in reads) + 3) & ~3
    n = len(reads)
    seq = np.full((n, stride), ord("\n"), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint16)
    for i, r in enumerate(reads):
        seq[i, :len(r)] = np.frombuffer(r.encode(), dtype=np.uint8)
        lens[i] = len(r)
    eng = pkg.Engine(plan, device=0)
    d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    d_idx = torch.zeros(n, dtype=torch.int64, device="cuda")
    dseq = torch.from_numpy(seq.reshape(-1)).cuda()
    dlens = torch.from_numpy(lens.view(np.int16)).cuda()
    # several submits: the wide table grows (rehash of wide slots) between them
    cuts = [0, (n // 7) & ~7, (n // 2) & ~7, n]  # (batches start 16-byte aligned)
    for a, b in zip(cuts, cuts[1:]):
        if b > a:
            if trace:  # (a submit writes its reads' outcomes from the start of the buffers it was given)
                eng.trace(d_out.data_ptr() + a, d_idx.data_ptr() + 8 * a)
            eng.submit_device(dseq.data_ptr() + a * stride, None, b - a, stride, stride, d_lens=dlens.data_ptr() + 2 * a)
    eng.sync()
    return eng, d_out.cpu().numpy()


CASES = {
    # the VERDICT's case: three 20-base captures, no conversion file at all (180 payload bits)
    "three_raw_20": dict(scheme="[20]ACGT{20}TT{20}", parts=[("cap", 20, 0), ("const", "ACGT"), ("cap", 20, 1), ("const", "TT"), ("cap", 20, 2)],
                         pools=[(40, 20), (300, 20), (300, 20)]),
    # Barcode-seq: one 40-base lineage barcode, nothing known
    "barcode_seq_40": dict(scheme="GTACCAGTC{40}TGCATGGAC", parts=[("const", "GTACCAGTC"), ("cap", 40, 0), ("const", "TGCATGGAC")],
                           pools=[(1500, 40)]),
    # ... with sample barcodes that ARE known (an index field next to the raw planes) and errors to correct in them
    "samples_plus_raw_35": dict(scheme="[8]GTACCAGTC{35}TGCATGGAC", parts=[("cap", 8, 0), ("const", "GTACCAGTC"), ("cap", 35, 1), ("const", "TGCATGGAC")],
                                pools=[(5, 8), (800, 35)], samples_from=0),
    # a 30-base random barcode behind a known DEL-style tuple: PCR-duplicate collapse under wide keys
    "known_plus_random_30": dict(scheme="[8]AGCTACG{8}TGGA(30)TAGAC", parts=[("cap", 8, 0), ("const", "AGCTACG"), ("cap", 8, 1), ("const", "TGGA"), ("cap", 30, 2), ("const", "TAGAC")],
                                 pools=[(2, 8), (8, 8), (600, 30)], samples_from=0, counted_from=[1]),
    # everything raw AND a long random barcode
    "raw_plus_random_28": dict(scheme="CCTAGG{24}AATT(28)GGATCC", parts=[("const", "CCTAGG"), ("cap", 24, 0), ("const", "AATT"), ("cap", 28, 1), ("const", "GGATCC")],
                               pools=[(20, 24), (200, 28)]),
}


def _build(name, n, seed):
    import ngs_barcode_count_amd as pkg
    c = CASES[name]
    rng = np.random.default_rng(seed + 1000)
    pools = [_pool(rng, k, ln) for k, ln in c["pools"]]
    reads = _reads(c["parts"], n, seed, pools)
    plan = pkg.Plan(c["scheme"])
    samples = None
    counted = None
    if "samples_from" in c:
        samples = {s: "sample_%d" % i for i, s in enumerate(pools[c["samples_from"]])}
        for s, i in samples.items():
            plan.add_sample(s, i)
    if "counted_from" in c:
        counted = [pools[k] for k in c["counted_from"]]
        for b, refs in enumerate(counted):
            for s in refs:
                plan.add_counted(b, s, s)
    o = oracle_lib.Oracle(c["scheme"], samples=samples, counted=counted)
    return plan, o, reads


@pytest.mark.parametrize("name", sorted(CASES))
def test_wide_keys_vs_oracle(name):
    import ngs_barcode_count_amd as pkg
    plan, o, reads = _build(name, 12_000, 7)
    assert plan.mode == "sparse"
    eng, outcomes = _run_engine(plan, reads)
    assert eng.kernel_name() == "long_match_kernel"
    exp = [parity_code(o.process(r)) for r in reads]
    # (which read of several with one key is the "duplicate" depends on the order they are counted in: the reference's
    # workers race for it too -- per read, matched and duplicate are one outcome; the counters below tell them apart)
    same = lambda a, b: a == b or {int(a), int(b)} == {0, 4}
    bad = [i for i in range(len(reads)) if not same(outcomes[i], exp[i])]
    assert not bad, (bad[:5], [(int(outcomes[i]), exp[i], reads[i]) for i in bad[:3]])
    got = eng.counters()
    assert {k: got[k] for k in o.counters} == o.counters
    rows = eng.result_rows()
    assert rows == o.rows()
    assert len(rows) >= 10 and max(r[2] for r in rows) > 1  # keys really repeat
    if "random" in name:
        assert got["duplicates"] > 500
    eng.close()


def parity_code(name):
    return {"matched": 0, "constant_region": 1, "sample_barcode": 2, "barcode": 3, "duplicates": 4, "low_quality": 5}[name]


def test_wide_keys_reset_and_reuse():
    """reset clears the wide table (slots and their ready flags); a second pass over the same reads counts the same"""
    plan, o, reads = _build("barcode_seq_40", 4000, 11)
    eng, _ = _run_engine(plan, reads, trace=False)
    first = (eng.counters(), eng.result_rows())
    eng.reset()
    assert eng.result_rows() == [] and sum(eng.counters().values()) == 0
    eng.close()
    eng, _ = _run_engine(plan, reads, trace=False)
    assert (eng.counters(), eng.result_rows()) == first
    for r in reads:
        o.process(r)
    assert first[1] == o.rows()
    eng.close()


@pytest.mark.parametrize("name,world", [("barcode_seq_40", 2), ("known_plus_random_30", 3), ("raw_plus_random_28", 2)])
def test_wide_keys_across_ranks(tmp_path, name, world):
    """the job's end-of-run exchange with keys several u64 wide: raw-key maps gathered on the root, random-barcode keys
    sent whole to the root's set (bc_engine_finish_all over the message-file transport; ranks on this one GPU)"""
    n = 9000
    cdir = tmp_path / "comm"
    cdir.mkdir()
    out = tmp_path / "job.json"
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), name, str(r), str(world), str(cdir), str(n), str(out)],
                              stderr=subprocess.PIPE, env=dict(os.environ, BC_COMM_TIMEOUT_S="120")) for r in range(world)]
    for r, p in enumerate(procs):
        _, err = p.communicate(timeout=300)
        assert p.returncode == 0, (r, err.decode()[-1500:])
    job = json.load(open(out))
    plan, o, reads = _build(name, n, 7)
    for r in reads:
        o.process(r)
    assert {k: job["counters"][k] for k in o.counters} == o.counters
    assert [tuple(r) for r in job["rows"]] == o.rows()


def _rank_main():
    """one rank of test_wide_keys_across_ranks"""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    name, rank, world, cdir, n, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), sys.argv[6]
    import torch  # noqa: F401 -- before the engine library: torch brings its own copy of the HIP runtime, and the one
    #                loaded second in a process finds no GPU
    import ngs_barcode_count_amd as pkg
    plan, _, reads = _build(name, n, 7)
    a, b = n * rank // world, n * (rank + 1) // world
    eng, _ = _run_engine(plan, reads[a:b], trace=False)
    comm = pkg.Comm.host(cdir, rank, world)
    counters, n_rows = eng.finish_all(comm, 0)
    if rank == 0:
        rows = eng.result_rows()
        assert len(rows) == n_rows
        json.dump({"counters": counters, "rows": rows}, open(out, "w"))
    comm.barrier()
    comm.close()
    eng.close()


if __name__ == "__main__":
    _rank_main()
