"""The multi-GPU job end to end THROUGH THE C ABI on the one GPU this box has: several processes, each with its own
engine on device 0 and its own shard of the reads, joined by bc_comm_create_host + bc_engine_finish_all (the
message-file transport; on a multi-GPU node the same exchange runs over RCCL).  The root's counters and rows must equal
the oracle's over ALL reads -- dense tables (incl. counts above the byte-packing limit), the random-barcode key
exchange, and both raw-key forms."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _oracle(case, n):
    import oracle_lib
    import workloads
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mp_rank
    w = mp_rank.make_case(case)
    seq, qual = w.synth.generate_host(0, n)
    if case.startswith("sparse"):
        o = oracle_lib.Oracle(w.scheme)
    else:
        o = workloads.oracle_for(w)
    o.process_batch(seq, qual if w.min_quality > 0 else None, w.read_len, w.read_len)
    return o.counters, o.rows()


@pytest.mark.parametrize("case,world,n,root", [("dense", 2, 60_001, 0), ("dense", 3, 40_000, 2), ("dense_hot", 2, 50_000, 1),
                                               # (two-level counting on: the bit map travels inside the packed bytes)
                                               ("dense+bits", 2, 60_001, 1), ("dense_hot+bits", 3, 50_000, 0),
                                               # (... or, tables sparse enough, as bit-map slices + the tables' non-zero entries)
                                               ("dense_big+bits", 3, 90_000, 2),
                                               ("random", 2, 60_000, 0), ("random", 3, 45_000, 1),
                                               ("sparse", 2, 30_000, 0), ("sparse_random", 3, 30_000, 2)])
def test_ranks_on_one_gpu_equal_the_oracle(tmp_path, case, world, n, root):
    cdir = tmp_path / "comm"
    cdir.mkdir()
    out = tmp_path / "job.json"
    env = dict(os.environ, BC_COMM_TIMEOUT_S="120", BC_COMM_VERBOSE="1")
    if case.endswith("+bits"):
        case = case[:-5]
        env["BC_BITMAP_MIN_ENTRIES"] = "1"
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_rank.py"), case, str(r), str(world), str(cdir),
                               str(n), str(root), str(out)], env=env, stderr=subprocess.PIPE) for r in range(world)]
    errs = []
    for r, p in enumerate(procs):
        _, err = p.communicate(timeout=300)
        assert p.returncode == 0, (r, err.decode()[-1500:])
        errs.append(err.decode())
    if case.startswith("dense"):  # which form the table exchange took (BC_COMM_VERBOSE: the root says)
        if "BC_BITMAP_MIN_ENTRIES" not in env:
            assert "byte-packed slices" in errs[root], errs[root][-400:]
        elif case in ("dense_big", "dense_hot"):  # (sparse enough / so small that every entry fits the side list)
            assert "bit-map slices" in errs[root], errs[root][-400:]
            if case == "dense_big":  # ... and the sums are small: two bit planes to the root
                assert "two bit planes" in errs[root], errs[root][-400:]
    job = json.load(open(out))
    exp_counters, exp_rows = _oracle(case, n)
    assert {k: job["counters"][k] for k in exp_counters} == exp_counters
    assert job["counters"]["total_reads"] == n
    assert [tuple(r) for r in job["rows"]] == exp_rows
    if case == "dense_hot":
        assert max(r[2] for r in job["rows"]) > 255 * world  # the side list really was needed
    if case.startswith("random") or case == "sparse_random":
        assert job["counters"]["duplicates"] > n // 10


def test_one_rank_communicator_is_a_plain_finish(tmp_path):
    import torch
    import ngs_barcode_count_amd as pkg
    import workloads
    w = workloads.make("config3", n_sets=(4, 30, 30, 30))
    n, R = 20_000, w.read_len
    seq, qual = w.synth.generate_host(0, n)
    eng = pkg.Engine(w.plan, device=0)
    eng.submit_host(seq, qual, R, R)
    comm = pkg.Comm.host(str(tmp_path), 0, 1)
    counters, rows = eng.finish_all(comm, 0)
    o = workloads.oracle_for(w)
    o.process_batch(seq, qual, R, R)
    assert {k: counters[k] for k in o.counters} == o.counters and rows == len(o.rows())
    assert eng.result_rows() == o.rows()
    c2, r2 = eng.finish_all(None, 0)  # no communicator at all
    assert c2 == counters and r2 == rows
    eng.close()


def test_rccl_communicator_of_one_rank():
    """librccl is found and loaded on first use, a unique id is made, a communicator initialised on this device and
    used (a one-rank job's exchange is the plain finish); the peer-to-peer sends themselves need a second GPU"""
    import torch
    import ngs_barcode_count_amd as pkg
    import workloads
    ident = pkg.Comm.unique_id()
    assert len(ident) == 128 and any(ident)
    comm = pkg.Comm.rccl(ident, 0, 1, 0)
    assert (comm.rank, comm.world) == (0, 1)
    comm.barrier()
    assert comm.sum_u64([5, 7], 0) == [5, 7]
    w = workloads.make("config3", n_sets=(4, 20, 20, 20))
    n, R = 10_000, w.read_len
    seq, qual = w.synth.generate_host(0, n)
    eng = pkg.Engine(w.plan, device=0)
    eng.submit_host(seq, qual, R, R)
    counters = eng.reduce_all(comm, 0)
    o = workloads.oracle_for(w)
    o.process_batch(seq, qual, R, R)
    assert {k: counters[k] for k in o.counters} == o.counters
    assert eng.result_rows() == o.rows()
    eng.close()
    comm.close()


@pytest.mark.parametrize("config,reads", [("config3", 1_000_000), ("config4", 400_000), ("config5", 1_000_000)])
def test_bench_n_ranks_path_on_one_gpu(config, reads):
    """bench.py's N > 1 code path end to end (launcher, one engine per rank, the ABI's end-of-run exchange inside the
    timed region, counters merged on rank 0) with both ranks on this box's one GPU: torch.distributed on gloo, the
    engines' exchange over message files.  What a multi-GPU node adds to this is RCCL in place of the files."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-one-gpu", "--config", config,
                        "--reads", str(reads), "--steps", "3", "--warmup", "1", "--no-extra", "--no-cpu"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["valid"] is False
    assert line["reduce_ms"] > 0
    out = line["outcomes"]
    assert out["total_reads"] == 2 * 3 * reads
    assert sum(out[k] for k in ("matched", "constant_region", "sample_barcode", "barcode", "duplicates", "low_quality")) == out["total_reads"]
    if config == "config4":
        # (a step is a job of its own there -- the key set is cleared every step -- and two ranks cut the read stream
        # into other steps than one rank does: only the exchange itself can be checked, in test_ranks_on_one_gpu_...)
        assert out["duplicates"] > reads // 4
        return
    # the same reads through ONE rank give the same outcome counters
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", config, "--reads", str(2 * reads),
                          "--steps", "3", "--warmup", "1", "--no-extra", "--no-cpu"], env=env, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    ref = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])["outcomes"]
    assert out == ref


def test_bench_falls_back_to_message_files_when_rccl_fails_on_one_rank():
    """bench.py at N > 1: the library's RCCL transport failing on ONE rank (faked here) must not leave the ranks in
    different exchanges or the driver without a line -- all of them vote, switch to the message-file transport, and the
    line says so"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["BC_BENCH_FAKE_RCCL_FAILURE"] = "1"  # rank 1 fails
    reads = 200_000
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-one-gpu", "--config", "config3",
                        "--reads", str(reads), "--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert "faked for the test" in line["config"]["exchange"] and line["config"]["exchange"].startswith("message files")
    assert line["outcomes"]["total_reads"] == 2 * 2 * reads and line["reduce_ms"] > 0
    assert "not usable" in p.stderr
