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
Prints ONE JSON line on rank 0 (stdout), as soon as the timed region, the box's copy rate and the CPU baseline are done:
the contract's fields, `roofline` (dominant kernel, HIP events on the engine's stream; per-launch first / median / last),
`cpu_baseline`, `box`, `reset_ms`.  Only then, at N = 1, the legs beside the headline run, each on its own and with its
failure recorded instead of raised: `finish` (table -> sparse rows on the host), `end_to_end` (host buffers -> counts
over PCIe), `ingest` (FASTQ file -> counts) and `extra` (the other BASELINE configs, each with its own roofline).  They
go to stderr as one `BENCH_EXTRA {...}` line and to gpurun_out/bench_extra.json -- never to stdout.
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
    ap.add_argument("--cpu-sample", type=int, default=4_000_000,
                    help="reads of the CPU baseline's independent-context leg (the reference-structure leg takes the first 1.5 M)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the legs beside the headline (finish, end-to-end, ingest, the other configs)")
    ap.add_argument("--selftest-one-gpu", action="store_true",
                    help="TEST ONLY (tests/test_gpu_multirank.py): the N > 1 path with every rank on device 0 -- the box has one "
                         "GPU --, torch.distributed on gloo and the engines' exchange over the message-file transport "
                         "instead of RCCL; prints a line marked invalid")
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
class Run:
    """One BASELINE config on this rank's GPU: resident batches, an engine, the timed steps.  measure() returns the
    contract's numbers (rank 0: complete); the legs that are NOT part of the headline (finish, box ceilings, host
    submit, ingest) are separate methods that main() calls only after the contract line is out."""

    def __init__(self, name, n, steps, warmup, world, rank, local, dev, comm=None, agree=None):
        self.agree, self.exchange_note = agree, ""  # agree(comm, why) -> (comm, note): see agree_on_transport
        import torch
        import ngs_barcode_count_amd as pkg
        from ngs_barcode_count_amd import distributed as bcdist
        import workloads
        self.torch, self.pkg, self.bcdist, self.workloads = torch, pkg, bcdist, workloads
        self.name, self.n, self.steps, self.warmup = name, n, steps, warmup
        self.world, self.rank, self.local, self.dev = world, rank, local, dev
        if name == "config5z":
            w = workloads.make("config5", zipf=True)
        elif name == "config4":
            # PCR copies per molecule geometric with mean 2, scattered over one job = one step's reads of all ranks (the
            # key set is cleared every step; every step draws the same molecules again with fresh sequencing errors)
            w = workloads.make(name, geo_total=n * world)
        else:
            w = workloads.make(name)
        self.w = w
        self.R = R = w.read_len
        self.with_qual = w.min_quality > 0
        self.random_mode = w.plan.random_barcode
        # The counters come first: the table is the randomly accessed allocation and gets the device memory of a fresh
        # process; the resident batches then take what is left.  The engine owns its table at every N, as in the
        # command-line program: the end-of-run exchange goes through the C ABI (bc_engine_reduce_all).
        self.eng = pkg.Engine(w.plan, device=local)
        self.comm = comm  # N > 1: the job's communicator (bc_comm: RCCL over xGMI), made once per process in main()
        # --- resident inputs: this rank's contiguous shard of the seeded read stream, one batch per step ------------
        # Every step counts reads it has not seen before (a job never counts the same batch twice), as many distinct
        # batches as the HBM holds next to the counters and the working memory of everything this process does later
        # (finish staging, host-submit and ingest buffers; N > 1: the byte-packed slices of the table exchange).
        batch_bytes = n * R * (2 if self.with_qual else 1)
        free_b, _ = torch.cuda.mem_get_info(dev)
        reserve = (6 << 30) if world == 1 else (6 << 30) + 3 * w.plan.table_entries
        self.n_batches = int(max(1, min(steps, (free_b - reserve) // batch_bytes)))
        first_read, _ = bcdist.shard(n * self.n_batches * world, rank, world)
        self.first_read = first_read
        self.batches = []
        for k in range(self.n_batches):
            bs = torch.empty(n * R, dtype=torch.uint8, device=dev)
            bq = torch.empty(n * R, dtype=torch.uint8, device=dev) if self.with_qual else None
            w.synth.generate_device(local, None, first_read + k * n, n, bs.data_ptr(), bq.data_ptr() if bq is not None else None)
            self.batches.append((bs, bq))
        torch.cuda.synchronize()

    def close(self):
        self.eng.close()
        self.batches = []
        self.torch.cuda.empty_cache()

    def _barrier(self):
        self.eng.sync()
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.torch.distributed.barrier()

    def measure(self):
        torch, pkg, eng, w = self.torch, self.pkg, self.eng, self.w
        dist = torch.distributed
        n, R, steps, world, rank, dev = self.n, self.R, self.steps, self.world, self.rank, self.dev
        step_no = [0]
        carried = dict.fromkeys(pkg.COUNTER_NAMES, 0)  # (outcome counters survive reset_results: nothing to carry)
        resets = []

        def step():
            # One step = one job of the BASELINE size: `n` reads nobody has counted before, into an empty Results.  (Rounds
            # 1-2 let the steps accumulate in one table: the kernel then slows down step by step as the 4 x 10^9 tuples
            # fill up -- every repeat is a second atomic --, which measured a 2 G-read job the config does not name.)
            # The emptying between jobs is part of the timed region: bc_engine_reset_results zeroes the blocks the last
            # job touched (two-level counting leaves the table almost all zeros) and the bit map.
            k = step_no[0] % self.n_batches
            if self.random_mode:
                eng.clear_keys()
            elif step_no[0] > 0:
                eng.reset_results()
                resets.append(0.0)
            bs, bq = self.batches[k]
            step_no[0] += 1
            eng.submit_device(bs.data_ptr(), bq.data_ptr() if bq is not None else None, n, R, R)

        for _ in range(self.warmup):
            step()
        if world > 1:
            # warm-up of the end-of-job exchange too: RCCL sets up its peer-to-peer connections on first use.  A failure
            # here (on any rank) moves every rank to the message-file transport (agree_on_transport).
            why = ""
            try:
                eng.reduce_all(self.comm, 0)
            except Exception as e:  # noqa: BLE001
                if self.agree is None:
                    raise
                why = "%s: %s" % (type(e).__name__, e)
            if self.agree is not None:
                self.comm, note = self.agree(self.comm, why)
                if note:
                    self.exchange_note = note
                    eng.reset()
                    eng.reduce_all(self.comm, 0)
        self._barrier()
        t_r = time.perf_counter()
        eng.reset()
        eng.sync()
        reset_ms = (time.perf_counter() - t_r) * 1e3
        eng.timing(True)
        step_no[0] = 0
        resets.clear()
        for key in carried:
            carried[key] = 0
        self._barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        eng.sync()  # (a caller-owned table: folds the first-occurrence bits into it -- part of the job, inside the region)
        t_steps = time.perf_counter() - t0
        reduce_ms = 0.0
        job_counters = None
        if world > 1:
            # the job's one exchange, through the C ABI: dense tables summed onto the root (all-to-all of byte-packed
            # slices over xGMI); random-barcode mode: keys to their owner ranks first, then the owners' per-tuple
            # distinct counts are summed
            tr = time.perf_counter()
            job_counters = eng.reduce_all(self.comm, 0)
            torch.cuda.synchronize()
            reduce_ms = (time.perf_counter() - tr) * 1e3
        self._barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed, t_steps, reduce_ms], dtype=torch.float64,
                             device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, t_steps, reduce_ms = t.tolist()

        sclk = eng.sclk_mhz()  # straight after the timed steps: the clock the kernels actually ran at
        each = eng.kernel_ms_each()
        kernel_ms, launches = eng.kernel_ms()
        eng.timing(False)
        if job_counters is not None:
            # (the counters of the passes before a reset never went through the exchange: add them up separately)
            extra = self.comm.sum_u64([carried[k] for k in pkg.COUNTER_NAMES], 0)
            counters = {k: job_counters[k] + extra[i] for i, k in enumerate(pkg.COUNTER_NAMES)}
        else:
            counters = eng.counters()
            for key in carried:
                counters[key] += carried[key]
        if rank != 0:
            return None
        total_reads = n * steps * world
        six = sum(counters[k] for k in ("matched", "constant_region", "sample_barcode", "barcode", "duplicates", "low_quality"))
        # exactly one outcome per read (random-barcode mode clears its set every step, so it holds there too)
        assert six == total_reads == counters["total_reads"], counters
        f_matched = (counters["matched"] + (counters["duplicates"] if self.random_mode else 0)) / max(counters["total_reads"], 1)
        b_alg = self.workloads.bytes_per_read(w, f_matched)
        avg_ms = kernel_ms / max(launches, 1)
        achieved = (b_alg * n) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        srt = sorted(each) or [0.0]
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None, "alg_bytes_per_launch": b_alg * n,
                "alg_bytes_per_read": b_alg, "kernel": eng.kernel_name(), "kernel_avg_ms": avg_ms,
                "launches": launches, "kernel_reads_per_s": n / (avg_ms * 1e-3) if avg_ms > 0 else 0.0,
                # per launch, in step order: the table fills up as a job goes on (two-level counting: more second
                # occurrences, each a table add on top of the bit), so first / median / last are stated, not one mean
                "kernel_ms_first": each[0] if each else None, "kernel_ms_median": srt[len(srt) // 2],
                "kernel_ms_last": each[-1] if each else None, "kernel_ms_min": srt[0], "kernel_ms_max": srt[-1],
                "kernel_ms_each": [round(x, 4) for x in each[:64]],
                "sclk_mhz": sclk}
        # HBM bytes per launch from the PMC counters (FETCH_SIZE doubled per the gfx950 correction, + WRITE_SIZE),
        # measured in separate rocprofv3 passes of this very workload (tools/profile.sh) and committed
        # under profiles/; null when no summary of this config and size exists
        try:
            prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_summary.json"))
            for f in reversed(prof):
                js = json.load(open(os.path.join(ROOT, "profiles", f)))
                wl = js.get("_workload", "")
                if (self.name + ":") in wl and "{:,}".format(n) in wl and "hbm_traffic_bytes_per_dispatch" in js:
                    roof["traffic"] = js["hbm_traffic_bytes_per_dispatch"]["total"]
                    roof["traffic_source"] = "profiles/" + f
                    break
        except (OSError, ValueError, KeyError):
            pass
        res = {"config": self.name, "workload": WORKLOAD_TEXT[self.name], "reads_per_step_per_gpu": n, "read_len": R,
               "distinct_batches": self.n_batches, "resets_in_region": len(resets),
               # (not timed one by one -- that would need a sync per step; reset_ms is one of them with its sync)
               "reset_in_region_ms": len(resets) * reset_ms,
               "value": total_reads / elapsed, "ms_per_step": elapsed * 1e3 / steps, "reduce_ms": reduce_ms,
               "reset_ms": reset_ms, "outcomes": {k: counters[k] for k in pkg.COUNTER_NAMES}, "roofline": roof}
        if not self.random_mode and world == 1 and w.plan.table_entries:
            try:  # rows the table holds at the end of the region (since the last reset): one sweep of the table
                res["table_rows_at_end"] = eng.nonzero_entries()
                res["table_entries"] = w.plan.table_entries
            except Exception as err:  # noqa: BLE001 -- a diagnostic never costs the measurement
                res["table_rows_at_end"] = {"error": str(err)}
        return res

    # ---- legs beside the headline (rank 0, N = 1), each called inside main()'s try/except ------------------------
    def leg_finish(self):
        """the job's end: compaction of the table into sparse rows on the host (bc_engine_finish: bounded staging)"""
        t_f = time.perf_counter()
        n_rows = self.eng.finish()
        return {"finish_ms": (time.perf_counter() - t_f) * 1e3, "finish_rows": n_rows}

    def leg_box(self):
        """this box's own ceilings: boxes of the pool differ by up to ~20 %"""
        torch = self.torch
        box = {}
        dseq = self.batches[0][0]
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
        if self.w.plan.table_entries and not self.random_mode:
            # random no-return atomics over the whole counter table (+1, then -1 at the same entries: the counts end
            # as they were): the rate the memory system sustains for plain counting alone
            self.eng.sync()
            box["atomic_Gps"] = self.pkg.probe_atomic_rate(self.local, self.eng.table_ptr, self.w.plan.table_entries, 1 << 27) / 1e9
            box["atomic_table_bytes"] = self.w.plan.table_entries * 4
        return box

    def host_sample(self, m):
        dseq, dqual = self.batches[0]
        hs = dseq[: m * self.R].cpu().numpy()
        hq = dqual[: m * self.R].cpu().numpy() if self.with_qual else None
        return hs, hq

    def leg_end_to_end(self, hs, hq):
        """host buffers -> counts (bc_engine_submit_host: pinned double buffers, H2D on a side stream)"""
        eng, R = self.eng, self.R
        m = hs.size // R
        eng.reset()
        eng.submit_host(hs[: 1_000_000 * R], hq[: 1_000_000 * R] if hq is not None else None, R, R)  # staging buffers allocated
        eng.sync()
        eng.reset()
        eng.sync()
        t_h = time.perf_counter()
        eng.submit_host(hs, hq, R, R)
        eng.sync()
        dt = time.perf_counter() - t_h
        assert eng.counters()["total_reads"] == m
        return {"what": "bc_engine_submit_host: %d reads in pageable host arrays -> pinned staging -> H2D on a side "
                        "stream -> kernel, to the end of counting" % m,
                "value": m / dt, "unit": "reads/s", "pcie_GBps": m * R * (2 if hq is not None else 1) / dt / 1e9}


def cpu_baselines(w, sample, threads):
    """The CPU oracle timed two ways on the host cores (baseline, not target): in the reference's own structure (the
    contract's `value`), and with one independent context per thread -- no shared queue, no shared Results -- which is
    the most the same per-read code gives on these cores."""
    import threading
    import numpy as np
    import oracle_lib
    R = w.read_len
    m = min(sample, 1_500_000)
    seq, qual = w.synth.generate_host(0, sample)
    if w.min_quality <= 0:
        qual = None
    rate, done, _ = cpu_baseline(w, seq[: m * R], qual[: m * R] if qual is not None else None, threads)
    out = {"value": rate, "unit": "reads/s", "cores": threads, "kind": "port",
           "sample": "first %d reads of rank 0's shard; CPU oracle (C restatement of parse.rs) in the reference's "
                     "structure: 1 reader + %d workers on a mutex-guarded deque, one mutex-guarded Results "
                     "(main.rs:69-121)" % (done, threads - 1)}
    try:
        ctxs = [workloads_oracle(w) for _ in range(threads)]
        n = seq.size // R
        cuts = [n * i // threads for i in range(threads + 1)]

        def work(i):
            a, b = cuts[i], cuts[i + 1]
            ctxs[i].process_batch(seq[a * R:b * R], qual[a * R:b * R] if qual is not None else None, R, R)

        ths = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        assert sum(sum(c.counters.values()) for c in ctxs) == n
        out["independent_contexts"] = {"value": n / dt, "unit": "reads/s", "cores": threads,
                                       "sample": "first %d reads, one oracle context per thread on its own slice, no shared "
                                                 "queue or Results (not the reference's structure)" % n}
    except Exception as err:  # noqa: BLE001
        out["independent_contexts"] = {"error": repr(err)}
    return out


def workloads_oracle(w):
    import workloads
    return workloads.oracle_for(w)


def emit_extras(extras):
    """what is measured beside the contract line goes to stderr and to gpurun_out/bench_extra.json: stdout carries ONE line"""
    text = json.dumps(extras)
    sys.stderr.write("BENCH_EXTRA " + text + "\n")
    sys.stderr.flush()
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "bench_extra.json"), "w") as f:
            f.write(text + "\n")
    except OSError:
        pass


def agree_on_transport(dist, torch, pkg, dev, rank, world, comm, why):
    """all ranks: keep the RCCL communicator if every rank's is fine, else all switch to message files"""
    import tempfile
    flag = torch.tensor([0 if why else 1], dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        return comm, ""
    sys.stderr.write("[bench rank %d] RCCL transport of the library not usable (%s); the job's exchange goes through message "
                     "files\n" % (rank, why or "another rank failed"))
    if comm is not None:
        try:
            comm.close()
        except Exception:  # noqa: BLE001
            pass
    box = [tempfile.mkdtemp(prefix="bc_bench_comm_") if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    whys = [None] * world
    dist.all_gather_object(whys, why)
    note = "message files (the library's RCCL transport failed: %s)" % "; ".join(sorted({w for w in whys if w}))[:300]
    return pkg.Comm.host(box[0], rank, world), note


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

    one_gpu = args.selftest_one_gpu  # TEST ONLY: all ranks on device 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if one_gpu else "nccl", rank=rank, world_size=world)
    assert torch.cuda.is_available(), "bench.py needs a GPU: the engine has no CPU path"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    comm, exchange_note = None, ""
    if world > 1:
        # the engine library's own communicator (bc_comm over RCCL): rank 0 makes the id, torch.distributed -- which the
        # timing contract needs anyway for its barrier and its max over ranks -- carries it to the others
        import ngs_barcode_count_amd as pkg
        fake_failure = os.environ.get("BC_BENCH_FAKE_RCCL_FAILURE")  # TEST ONLY (with --selftest-one-gpu): the fallback below
        if one_gpu and not fake_failure:
            import tempfile
            box = [tempfile.mkdtemp(prefix="bc_bench_comm_") if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            comm = pkg.Comm.host(box[0], rank, world)
        else:
            # If the library's RCCL transport cannot be set up, or its first exchange fails, on ANY rank, all ranks
            # agree (through torch.distributed) to run the job's exchange over the library's other transport -- device
            # buffers staged through message files -- and the line says so (config.exchange): a slower, honest number
            # instead of none.  (The transport has run on one GPU only so far: DESIGN.md 6.)
            why = ""
            try:
                if fake_failure:  # (the other ranks would be fine: they learn of the failure from the vote)
                    if rank == int(fake_failure):
                        raise RuntimeError("faked for the test")
                else:
                    ident = torch.zeros(128, dtype=torch.uint8, device=dev)
                    if rank == 0:
                        ident.copy_(torch.frombuffer(bytearray(pkg.Comm.unique_id()), dtype=torch.uint8))
                    dist.broadcast(ident, src=0)
                    comm = pkg.Comm.rccl(bytes(ident.cpu().numpy().tobytes()), rank, world, local)
                    comm.barrier()
                    got = comm.sum_u64([rank + 1], 0)
                    if rank == 0 and got[0] != world * (world + 1) // 2:
                        raise RuntimeError("sum over the ranks came out as %d" % got[0])
            except Exception as e:  # noqa: BLE001 -- whatever it is, the ranks must agree on what to do next
                why = "%s: %s" % (type(e).__name__, e)
            comm, exchange_note = agree_on_transport(dist, torch, pkg, dev, rank, world, comm, why)
    n = args.reads or DEFAULT_READS[args.config]
    agree = None
    if world > 1 and not one_gpu:
        agree = lambda c, why: agree_on_transport(dist, torch, pkg, dev, rank, world, c, why)  # noqa: E731
    run = Run(args.config, n, args.steps, args.warmup, world, rank, local, dev, comm, agree)
    res = run.measure()
    exchange_note = run.exchange_note or exchange_note
    if rank != 0:
        run.close()
        if world > 1:
            dist.destroy_process_group()
        return

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
                   "distinct_batches": res["distinct_batches"], "resets_in_region": res["resets_in_region"],
                   "parallelism": "reads sharded over %d GPU(s), one process each; one exchange at the end through the C ABI "
                                  "(bc_engine_reduce_all: all-to-all sum of the counter tables, RCCL over xGMI)" % world,
                   **({"exchange": exchange_note} if exchange_note else {})},
        "roofline": res["roofline"],
        "outcomes": res["outcomes"],
        **({"valid": False, "selftest": "all ranks on one GPU, message-file exchange: NOT a measurement"} if args.selftest_one_gpu else {}),
        "reduce_ms": res["reduce_ms"],
        "reset_ms": res["reset_ms"],
        "reset_in_region_ms": res["reset_in_region_ms"],
    }
    for k in ("table_rows_at_end", "table_entries"):
        if k in res:
            out[k] = res[k]
    if world == 1:
        try:  # this box's copy rate (5 copies of 2 GiB: 10 ms): the headline's second denominator
            box = run.leg_box()
            out["box"] = box
            out["roofline"]["box_copy_GBps"] = box.get("copy_GBps")
            out["roofline"]["frac_of_box_copy"] = out["roofline"]["achieved"] / box["copy_GBps"] if box.get("copy_GBps") else None
        except Exception as err:  # noqa: BLE001 -- nothing beside the timed region may cost the line
            out["box"] = {"error": repr(err)}
    if not args.no_cpu and world == 1:  # (rank 0 at N = 1 only: the contract's baseline leg)
        try:
            out["cpu_baseline"] = cpu_baselines(run.w, min(args.cpu_sample, n), max(2, min(os.cpu_count() or 2, 16)))
        except Exception as err:  # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(err)}
    # ---- the contract line: out before anything optional runs ---------------------------------------------------
    print(json.dumps(out))
    sys.stdout.flush()

    if world == 1 and not args.no_extra:
        extras = {"of": {"config": args.config, "steps": args.steps, "warmup": args.warmup}}

        def leg(name, fn):
            t0 = time.perf_counter()
            try:
                extras[name] = fn()
            except BaseException as err:  # noqa: BLE001 -- recorded, the next leg still runs
                extras[name] = {"error": repr(err)}
                if isinstance(err, KeyboardInterrupt):
                    raise
            sys.stderr.write("bench.py: leg %s done in %.1f s\n" % (name, time.perf_counter() - t0))

        if not run.random_mode:
            leg("finish", run.leg_finish)
        sample = [None, None]

        def e2e():
            m = min(n, 8_000_000)
            sample[0], sample[1] = run.host_sample(m)
            del run.batches[1:]
            torch.cuda.empty_cache()
            return run.leg_end_to_end(sample[0], sample[1])

        leg("end_to_end", e2e)
        if sample[0] is not None:
            leg("ingest", lambda: ingest_leg(run.eng, run.w, sample[0], sample[1], sample[0].size // run.R, run.R))
        run.close()
        del sample
        if args.config == "config3":
            # the other BASELINE configs, each at its own size, a few steps each (the headline stays config 3)
            extras["extra"] = []
            for name in ("config2", "config4", "config5", "config5z"):
                def other(name=name):
                    r2 = Run(name, DEFAULT_READS[name], 5, 2, 1, 0, local, dev)
                    try:
                        r = r2.measure()
                    finally:
                        r2.close()
                    return {k: r[k] for k in ("config", "workload", "reads_per_step_per_gpu", "distinct_batches", "value",
                                              "ms_per_step", "roofline", "outcomes")}
                leg("_" + name, other)
                extras["extra"].append(extras.pop("_" + name))
        emit_extras(extras)
    else:
        run.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
