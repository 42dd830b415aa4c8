#!/usr/bin/env python3
"""bench.py -- reads/s of the per-read match/count hot path on MI355X.

One "step" = one pass of the hot path (bc_engine_submit_device) over one batch of synthetic
reads that is already resident in HBM.  Default workload = BASELINE.json configs[2]
("DEL 3x8-nt vs 3x1000 refs, 20% mismatch Hamming correction + min-quality filter, 100M reads,
1 MI355X"), the configuration the north-star target is quoted on.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config config3|config2|config4|config5] [--reads M]

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); every rank counts its own
contiguous shard of reads (weak scaling, no collective on the data path) and the dense counter
tables are sum-reduced ONCE with RCCL inside the timed region, as the job would do at its end.
Under torchrun (WORLD_SIZE set) this process is one rank; started plainly with --gpus N > 1 it
starts the N ranks itself as child processes -- before anything here has touched a GPU -- and
passes rank 0's line through.
Prints ONE JSON line on rank 0.  Besides the contract's fields: `roofline` (dominant kernel, HIP events
on the engine's stream), `cpu_baseline`, `end_to_end` (host buffers -> counts over PCIe), `reset_ms` /
`finish_ms` (zeroing the 16 GB table; compacting it into sparse rows on the host), `box` (this box's own
copy and random-atomic rates: boxes of the pool differ) and, at N = 1, `extra`: the other BASELINE
configs in the same invocation, each with its own roofline.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); a device copy reaches ~5-6.3 TB/s depending on the box
# reads per step per GPU: the BASELINE.json sizes (config 4: 400 M over 8 GPUs; config 5: 1 B over 8 GPUs)
DEFAULT_READS = {"config2": 10_000_000, "config3": 100_000_000, "config4": 50_000_000, "config5": 125_000_000,
                 "config5z": 125_000_000}
WORKLOAD_TEXT = {
    "config2": "DEL [8]+3x{8} vs 4 samples + 3x1000 refs, clean reads, exact match only (BASELINE configs[1])",
    "config3": "DEL [8]+3x{8} vs 4 samples + 3x1000 refs, 1% substitutions + 0.1% N, 20% mismatch budgets, "
               "--min-quality 20 (BASELINE configs[2])",
    "config4": "DEL [8]+3x{8}+(12) random barcode vs 4 samples + 3x1000 refs, PCR duplicates (copies per molecule geometric, mean 2), "
               "1% substitutions + 0.1% N (BASELINE configs[3], per-GPU shard of 50M reads; set cleared every step)",
    "config5": "CRISPR {20} vs 100k guides, 1% substitutions + 0.1% N, <=4 mismatches (BASELINE configs[4], per-GPU shard of 125M reads)",
    "config5z": "config5 with guide ranks drawn Zipf-like (P(k) ~ 1/k): the counter hot-spot variant of SURVEY.md 8(d)",
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="config3", choices=sorted(DEFAULT_READS))
    ap.add_argument("--reads", type=int, default=0, help="reads per step per GPU (default: the config's size)")
    ap.add_argument("--cpu-sample", type=int, default=1_500_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the other configs and the end-to-end / finish legs")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="TEST ONLY (tests/test_bench_launcher.py): the rank plumbing on gloo with the host emulation of "
                         "the lane code; prints a line marked invalid, measures nothing")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torchrun
# ------------------------------------------------------------------------------------------------
def launch_ranks(n, argv):
    """Starts n rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), waits
    for them, passes rank 0's stdout through and returns the worst exit code.  The parent never initialises a GPU
    (nothing GPU-related is imported before this point) and does not exec: the ranks are fresh children."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = max(rc, abs(p.wait()))
    sys.stdout.write(out.decode(errors="replace"))
    sys.stdout.flush()
    return rc


def cpu_baseline(w, seq, qual, threads):
    """The CPU oracle (C restatement of the reference's parse.rs path) run in the reference's own structure
    (src/main.rs:69-121): this thread is the reader posting packed 4-line records to a mutex-guarded deque, threads-1
    workers pop, match on their own clone of the static inputs and add into ONE mutex-guarded Results."""
    import oracle_lib
    import workloads
    workers = [workloads.oracle_for(w) for _ in range(max(1, threads - 1))]
    shared = workloads.oracle_for(w)
    t0 = time.perf_counter()
    total = oracle_lib.run_reference_threads(workers, shared, seq, qual, w.read_len, w.read_len)
    dt = time.perf_counter() - t0
    done = seq.size // w.read_len
    assert sum(total.values()) == done, total
    return done / dt, done, total


def ingest_leg(eng, w, hs, hq, m, R):
    """writes the first m synthetic reads as a plain FASTQ file (4 lines per record, as a sequencer writes it), then
    times bc_fastq_count on it twice: the first call also pins its chunk buffers, the second finds them ready"""
    import tempfile
    import numpy as np
    tmp = tempfile.mkdtemp(prefix="bc_bench_")
    fq = os.path.join(tmp, "reads.fastq")
    name = np.frombuffer(b"@SYN:1:FC:1:1101:", dtype=np.uint8)
    try:
        with open(fq, "wb") as f:
            s2 = hs.reshape(m, R)
            q2 = (hq if hq is not None else np.full(m * R, ord("I"), dtype=np.uint8)).reshape(m, R)
            step = 500_000
            for a in range(0, m, step):
                b = min(m, a + step)
                idx = np.char.zfill(np.arange(a, b).astype("U9"), 9).astype("S9").view(np.uint8).reshape(b - a, 9)
                rec = np.empty((b - a, name.size + 9 + 1 + R + 1 + 2 + R + 1), dtype=np.uint8)
                o = 0
                rec[:, o:o + name.size] = name; o += name.size
                rec[:, o:o + 9] = idx; o += 9
                rec[:, o] = 10; o += 1
                rec[:, o:o + R] = s2[a:b]; o += R
                rec[:, o] = 10; rec[:, o + 1] = ord("+"); rec[:, o + 2] = 10; o += 3
                rec[:, o:o + R] = q2[a:b]; o += R
                rec[:, o] = 10
                f.write(rec.tobytes())
        size = os.path.getsize(fq)
        out = {"what": "bc_fastq_count on a plain FASTQ file of %d reads (%.2f GB, page cache) to the end of counting; engine "
                       "creation excluded" % (m, size / 1e9), "unit": "reads/s"}
        for label in ("first_call", "value"):
            eng.reset()
            eng.sync()
            t0 = time.perf_counter()
            total = eng.count_fastq(fq)
            eng.sync()
            dt = time.perf_counter() - t0
            assert total == m and eng.counters()["total_reads"] == m
            out[label] = m / dt
            out["file_GBps" if label == "value" else "first_call_file_GBps"] = size / dt / 1e9
        return out
    finally:
        try:
            os.remove(fq)
            os.rmdir(tmp)
        except OSError:
            pass


def selftest_cpu(args, world, rank):
    """TEST ONLY: the launcher / rank / reduce plumbing on CPU ranks (gloo), tables built by the host emulation of the
    lane code (tests/emu).  Nothing is measured and the line says so."""
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_lib
    import ngs_barcode_count_amd as pkg
    from ngs_barcode_count_amd import distributed as bcdist
    import workloads
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n = args.reads or 2000
    w = workloads.make("config3", n_sets=(4, 40, 40, 40))
    first, count = bcdist.shard(n * world, rank, world)
    seq, qual = w.synth.generate_host(first, count)
    eplan = pkg.Plan(w.scheme, lib=emu_lib.lib())
    for i, s in enumerate(w.samples):
        eplan.add_sample(s, "sample_%d" % i)
    for b, refs in enumerate(w.counted):
        for i, s in enumerate(refs):
            eplan.add_counted(b, s, "bb%d_%d" % (b + 1, i))
    eplan.set_min_quality(20.0)
    outc, idx, entries, _ = emu_lib.emulate(eplan, seq, qual, None, 100, 100)
    table = torch.from_numpy(np.bincount(idx[outc == 0].astype(np.int64), minlength=entries).astype(np.int32))
    counters = {k: int((outc == i).sum()) for i, k in enumerate(pkg.COUNTER_NAMES)}
    counters["total_reads"], counters["unsupported_reads"] = count, 0
    t0 = time.perf_counter()
    bcdist.reduce_table(table, dst=0)
    reduce_ms = (time.perf_counter() - t0) * 1e3
    total = bcdist.reduce_counters(counters, torch.device("cpu"), dst=0)
    if rank == 0:
        print(json.dumps({"metric": "selftest (CPU ranks, host emulation; NOT a measurement)", "valid": False, "n_gpus": world,
                          "value": 0.0, "unit": "reads/s", "reduce_ms": reduce_ms, "table_sum": int(table.sum()),
                          "outcomes": total}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------
# one workload on this rank's GPU
# ------------------------------------------------------------------------------------------------
def run_config(name, n, steps, warmup, world, rank, local, dev, legs):
    """Counts `steps` x n resident reads of BASELINE config `name`; returns a dict of measurements (rank 0: complete).
    legs: also measure reset / finish / host-submit / cpu baseline (rank 0, N = 1 reporting only)."""
    import torch
    import torch.distributed as dist
    import ngs_barcode_count_amd as pkg
    from ngs_barcode_count_amd import distributed as bcdist
    import workloads

    if name == "config5z":
        w = workloads.make("config5", zipf=True)
    elif name == "config4":
        # PCR copies per molecule geometric with mean 2, scattered over one job = one step's reads of all ranks (the key
        # set is cleared every step; every step draws the same molecules again with fresh sequencing errors)
        w = workloads.make(name, geo_total=n * world)
    else:
        w = workloads.make(name)
    R = w.read_len
    with_qual = w.min_quality > 0
    # N > 1: the counter table is this process's own tensor, so that it can be reduced with RCCL (allocated first: it is
    # the randomly accessed one and should get the most contiguous device memory the process can have).  N = 1: the
    # engine owns it, as in the command-line program.
    table = torch.zeros(w.plan.table_entries, dtype=torch.int32, device=dev) if world > 1 else None
    torch.cuda.synchronize()
    # --- resident inputs: this rank's contiguous shard of the seeded read stream, one batch per step -----------------
    # Every step counts reads it has not seen before (a job never counts the same batch twice), as many distinct
    # batches as the HBM holds next to the table; with more steps than that the batches are gone through again.
    batch_bytes = n * R * (2 if with_qual else 1)
    free_b, _ = torch.cuda.mem_get_info(dev)
    # (left free: the engine's table and bit map at N = 1; at N > 1 also the byte-packed slices of the table exchange)
    spare = (24 << 30) if world == 1 else (48 << 30)
    n_batches = int(max(1, min(steps, (free_b - spare) // batch_bytes)))
    first_read, _ = bcdist.shard(n * n_batches * world, rank, world)
    batches = []
    for k in range(n_batches):
        bs = torch.empty(n * R, dtype=torch.uint8, device=dev)
        bq = torch.empty(n * R, dtype=torch.uint8, device=dev) if with_qual else None
        w.synth.generate_device(local, None, first_read + k * n, n, bs.data_ptr(), bq.data_ptr() if with_qual else None)
        batches.append((bs, bq))
    torch.cuda.synchronize()
    dseq, dqual = batches[0]
    eng = pkg.Engine(w.plan, device=local, table_ptr=table.data_ptr() if table is not None else None)
    random_mode = w.plan.random_barcode
    step_no = [0]

    def step():
        if random_mode:
            eng.clear_keys()  # a step is one whole job: otherwise every later step would see only duplicates
        bs, bq = batches[step_no[0] % n_batches]
        step_no[0] += 1
        eng.submit_device(bs.data_ptr(), bq.data_ptr() if with_qual else None, n, R, R)

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(warmup):
        step()
    if world > 1:
        # warm-up of the end-of-job exchange too: RCCL sets up its peer-to-peer connections on first use
        if random_mode:
            bcdist.exchange_keys(torch.arange(world * 64, dtype=torch.int64, device=dev))
        bcdist.reduce_table(torch.ones(world * 4096, dtype=torch.int32, device=dev), dst=0)
    barrier()
    t_r = time.perf_counter()
    eng.reset()
    eng.sync()
    reset_ms = (time.perf_counter() - t_r) * 1e3
    eng.timing(True)
    step_no[0] = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    eng.sync()  # (large dense tables: folds the first-occurrence bits into the table -- part of the job, inside the region)
    t_steps = time.perf_counter() - t0
    reduce_ms = 0.0
    fixed_counters = None
    if world > 1:
        tr = time.perf_counter()
        if random_mode:
            # set sizes do not add: exchange the keys so that each has one owner; every rank then
            # turns its keys into per-tuple distinct counts and those tables are summed onto the root
            fixed_counters = bcdist.finish_random(eng, dev, dst=0, table=table)
        else:
            bcdist.reduce_table(table, dst=0)  # the job's one exchange: all-to-all sum of the counter tables
        torch.cuda.synchronize()
        reduce_ms = (time.perf_counter() - tr) * 1e3
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed, t_steps, reduce_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, t_steps, reduce_ms = t.tolist()

    sclk = eng.sclk_mhz()  # straight after the timed steps: the clock the kernels actually ran at
    kernel_ms, launches = eng.kernel_ms()
    eng.timing(False)
    counters = fixed_counters if fixed_counters is not None else bcdist.reduce_counters(eng.counters(), dev, dst=0)
    total_reads = n * steps * world
    res = None
    if rank == 0:
        six = sum(counters[k] for k in ("matched", "constant_region", "sample_barcode", "barcode", "duplicates", "low_quality"))
        # exactly one outcome per read (random-barcode mode clears its set every step, so it holds there too)
        assert six == total_reads == counters["total_reads"], counters
        f_matched = (counters["matched"] + (counters["duplicates"] if random_mode else 0)) / max(counters["total_reads"], 1)
        b_alg = workloads.bytes_per_read(w, f_matched)
        avg_ms = kernel_ms / max(launches, 1)
        achieved = (b_alg * n) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        res = {"config": name, "workload": WORKLOAD_TEXT[name], "reads_per_step_per_gpu": n, "read_len": R,
               "distinct_batches": n_batches,
               "value": total_reads / elapsed, "ms_per_step": elapsed * 1e3 / steps, "reduce_ms": reduce_ms,
               "reset_ms": reset_ms, "outcomes": {k: counters[k] for k in pkg.COUNTER_NAMES},
               "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK_GBS, "traffic": None, "alg_bytes_per_launch": b_alg * n,
                            "alg_bytes_per_read": b_alg, "kernel": eng.kernel_name(), "kernel_avg_ms": avg_ms,
                            "launches": launches, "kernel_reads_per_s": n / (avg_ms * 1e-3) if avg_ms > 0 else 0.0,
                            "sclk_mhz": sclk}}
        # HBM bytes per launch from the PMC counters (FETCH_SIZE doubled per the gfx950 correction, + WRITE_SIZE),
        # measured in separate rocprofv3 passes of this very workload (tools/profile.sh) and committed
        # under profiles/; null when no summary of this config and size exists
        try:
            prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_summary.json"))
            for f in reversed(prof):
                js = json.load(open(os.path.join(ROOT, "profiles", f)))
                wl = js.get("_workload", "")
                if name in wl and "{:,}".format(n) in wl and "hbm_traffic_bytes_per_dispatch" in js:
                    res["roofline"]["traffic"] = js["hbm_traffic_bytes_per_dispatch"]["total"]
                    res["roofline"]["traffic_source"] = "profiles/" + f
                    break
        except OSError:
            pass

    if rank == 0 and legs:
        # ---- the job's end: compaction of the table into sparse rows on the host (bc_engine_finish) ----
        if not random_mode:
            t_f = time.perf_counter()
            n_rows = eng.finish()
            res["finish_ms"] = (time.perf_counter() - t_f) * 1e3
            res["finish_rows"] = n_rows

        # ---- this box's own ceilings, measured after the timed region: boxes of the pool differ by up to ~20 % ----
        box = {}
        try:
            src = dseq[: min(dseq.numel(), 2 << 30)]
            dst = torch.empty_like(src)
            dst.copy_(src)
            torch.cuda.synchronize()
            tc = time.perf_counter()
            for _ in range(5):
                dst.copy_(src)
            torch.cuda.synchronize()
            box["copy_GBps"] = 2.0 * src.numel() * 5 / (time.perf_counter() - tc) / 1e9
            del dst
        except RuntimeError:
            pass
        if w.plan.table_entries:
            # random no-return atomics over the whole counter table (+1, then -1 at the same entries: the counts end as they
            # were): the rate the memory system sustains for plain counting alone
            box["atomic_Gps"] = pkg.probe_atomic_rate(local, eng.table_ptr, w.plan.table_entries, 1 << 27) / 1e9
            box["atomic_table_bytes"] = w.plan.table_entries * 4
        res["box"] = box
        res["roofline"]["box_copy_GBps"] = box.get("copy_GBps")
        res["roofline"]["frac_of_box_copy"] = (res["roofline"]["achieved"] / box["copy_GBps"]) if box.get("copy_GBps") else None

        # ---- end to end: host buffers -> counts (bc_engine_submit_host: pinned double buffers, H2D on a side stream) ----
        m = min(n, 8_000_000)
        hs = dseq[:m * R].cpu().numpy()
        hq = dqual[:m * R].cpu().numpy() if with_qual else None
        del batches[1:]
        torch.cuda.empty_cache()
        eng.reset()
        eng.submit_host(hs[: 1_000_000 * R], hq[: 1_000_000 * R] if with_qual else None, R, R)  # staging buffers allocated
        eng.sync()
        eng.reset()
        eng.sync()
        t_h = time.perf_counter()
        eng.submit_host(hs, hq, R, R)
        eng.sync()
        dt = time.perf_counter() - t_h
        res["end_to_end"] = {"what": "bc_engine_submit_host: %d reads in pageable host arrays -> pinned staging -> H2D on a side "
                                     "stream -> kernel, to the end of counting" % m,
                             "value": m / dt, "unit": "reads/s", "pcie_GBps": m * R * (2 if with_qual else 1) / dt / 1e9}
        assert eng.counters()["total_reads"] == m
        res["_host_sample"] = (hs, hq)

        # ---- ingest: a FASTQ file in the page cache -> counts (bc_fastq_count: parallel pread into pinned chunks, raw
        # text over PCIe, newline scan / record split / gather on the device, match kernel) ----
        try:
            res["ingest"] = ingest_leg(eng, w, hs, hq, m, R)
        except OSError as err:  # no room for the file
            res["ingest"] = {"error": str(err)}
    if rank != 0:
        res = None
    eng.close()
    del table, dseq, dqual, batches
    torch.cuda.empty_cache()
    return res, w


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: start one rank per GPU (or leave WORLD_SIZE unset and "
                 "let bench.py start them)" % (args.gpus, world))

    for _p in (ROOT, os.path.join(ROOT, "tests")):
        if _p not in sys.path:
            sys.path.insert(0, _p)
    if args.selftest_cpu:
        return selftest_cpu(args, world, rank)

    # the specialised kernel is normally precompiled by build(); should the cache miss, compile it during the
    # (untimed) warm-up rather than on a worker thread part-way through the timed steps
    os.environ.setdefault("BC_JIT", "force")
    import torch
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    assert torch.cuda.is_available(), "bench.py needs a GPU: the engine has no CPU path"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    n = args.reads or DEFAULT_READS[args.config]
    legs = world == 1 and not args.no_extra
    res, w = run_config(args.config, n, args.steps, args.warmup, world, rank, local, dev, legs)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    host_sample = res.pop("_host_sample", None)
    crispr = args.config.startswith("config5")
    out = {
        "metric": "reads/sec (whole node), CRISPR 20nt vs 100k guides" if crispr else "reads/sec (whole node), 3x8nt DEL vs 3x1k refs",
        "value": res["value"],
        "unit": "reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": res["workload"], "config": args.config, "reads_per_step_per_gpu": n, "read_len": res["read_len"],
                   "distinct_batches": res["distinct_batches"],
                   "parallelism": "reads sharded over %d GPU(s); one all-to-all sum of the counter tables over xGMI at the end" % world},
        "roofline": res["roofline"],
        "outcomes": res["outcomes"],
        "reduce_ms": res["reduce_ms"],
        "reset_ms": res["reset_ms"],
    }
    for k in ("finish_ms", "finish_rows", "end_to_end", "ingest", "box"):
        if k in res:
            out[k] = res[k]

    if not args.no_cpu:
        m = min(args.cpu_sample, n)
        if host_sample is not None:
            hs, hq = host_sample[0][: m * w.read_len], (host_sample[1][: m * w.read_len] if host_sample[1] is not None else None)
        else:
            synth_seq, synth_qual = w.synth.generate_host(0, m)
            hs, hq = synth_seq, (synth_qual if w.min_quality > 0 else None)
        threads = max(2, min(os.cpu_count() or 2, 16))
        rate, done, _ = cpu_baseline(w, hs, hq, threads)
        out["cpu_baseline"] = {"value": rate, "unit": "reads/s", "cores": threads, "kind": "port",
                               "sample": "first %d reads of rank 0's batch; CPU oracle (C restatement of parse.rs) in the "
                                         "reference's structure: 1 reader + %d workers on a mutex-guarded deque, one "
                                         "mutex-guarded Results (main.rs:69-121)" % (done, threads - 1)}
    del host_sample

    if legs and args.config == "config3":
        # the other BASELINE configs, each at its own size, a few steps each (the headline stays config 3)
        extra = []
        for name in ("config2", "config4", "config5", "config5z"):
            r, _ = run_config(name, DEFAULT_READS[name], 3, 1, 1, 0, local, dev, False)
            extra.append({k: r[k] for k in ("config", "workload", "reads_per_step_per_gpu", "distinct_batches", "value",
                                            "ms_per_step", "roofline", "outcomes")})
        out["extra"] = extra
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
