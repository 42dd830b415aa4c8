#!/usr/bin/env python3
"""bench.py -- reads/s of the per-read match/count hot path on MI355X.

One "step" = one pass of the hot path (bc_engine_submit_device) over one batch of synthetic
reads that is already resident in HBM.  Default workload = BASELINE.json configs[2]
("DEL 3x8-nt vs 3x1000 refs, 20% mismatch Hamming correction + min-quality filter, 100M reads,
1 MI355X"), the configuration the north-star target is quoted on.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config config3|config2|config5] [--reads M]

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); every rank counts its own
contiguous shard of reads (weak scaling, no collective on the data path) and the dense counter
tables are sum-reduced ONCE with RCCL inside the timed region, as the job would do at its end.
Prints ONE JSON line on rank 0.
"""
import argparse
import concurrent.futures
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch
import torch.distributed as dist

import ngs_barcode_count_amd as pkg
from ngs_barcode_count_amd import distributed as bcdist
import workloads

# the specialised kernel is normally precompiled by build(); should the cache miss, compile it during the
# (untimed) warm-up rather than on a worker thread part-way through the timed steps
os.environ.setdefault("BC_JIT", "force")

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); a device copy reaches ~5-6.3 TB/s depending on the box
DEFAULT_READS = {"config2": 10_000_000, "config3": 100_000_000, "config4": 50_000_000, "config5": 20_000_000}
WORKLOAD_TEXT = {
    "config2": "DEL [8]+3x{8} vs 4 samples + 3x1000 refs, clean reads, exact match only (BASELINE configs[1])",
    "config3": "DEL [8]+3x{8} vs 4 samples + 3x1000 refs, 1% substitutions + 0.1% N, 20% mismatch budgets, "
               "--min-quality 20 (BASELINE configs[2])",
    "config4": "DEL [8]+3x{8}+(12) random barcode vs 4 samples + 3x1000 refs, PCR duplicates (2 reads per molecule), "
               "1% substitutions + 0.1% N (BASELINE configs[3], per-GPU shard of 50M reads; set cleared every step)",
    "config5": "CRISPR {20} vs 100k guides, 1% substitutions + 0.1% N, <=4 mismatches (BASELINE configs[4], per-GPU shard)",
}


def cpu_baseline(w, seq, qual, threads):
    """the CPU oracle (restatement of the reference's parse.rs path) on a bounded sample, `threads`
    contexts over disjoint slices -- the reference's own structure is N identical workers"""
    n = seq.size // w.read_len
    per = (n + threads - 1) // threads
    ctxs = [workloads.oracle_for(w) for _ in range(threads)]

    def work(t):
        a, b = t * per, min(n, (t + 1) * per)
        if a < b:
            ctxs[t].process_batch(seq[a * w.read_len:b * w.read_len], qual[a * w.read_len:b * w.read_len], w.read_len,
                                  w.read_len)
        return b - a

    t0 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(threads) as ex:
        done = sum(ex.map(work, range(threads)))
    dt = time.perf_counter() - t0
    matched = sum(c.counters["matched"] for c in ctxs)
    return done / dt, done, matched


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="config3", choices=sorted(DEFAULT_READS))
    ap.add_argument("--reads", type=int, default=0, help="reads per step per GPU (default: the config's size)")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    assert torch.cuda.is_available(), "bench.py needs a GPU: the engine has no CPU path"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    n = args.reads or DEFAULT_READS[args.config]
    w = workloads.make(args.config, n_molecules=(n * world) // 2 if args.config == "config4" else None)
    R = w.read_len
    with_qual = w.min_quality > 0

    # the counter table first: it is the randomly accessed one, so it should get the most contiguous device memory
    # (largest page fragments) the process can have
    table = torch.zeros(w.plan.table_entries, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    # --- resident inputs: this rank's contiguous shard of the seeded read stream -------------------
    dseq = torch.empty(n * R, dtype=torch.uint8, device=dev)
    dqual = torch.empty(n * R, dtype=torch.uint8, device=dev)
    first_read, _ = bcdist.shard(n * world, rank, world)
    w.synth.generate_device(local, None, first_read, n, dseq.data_ptr(), dqual.data_ptr())
    torch.cuda.synchronize()
    eng = pkg.Engine(w.plan, device=local, table_ptr=table.data_ptr())
    qptr = dqual.data_ptr() if with_qual else None

    random_mode = w.plan.random_barcode

    def step():
        if random_mode:
            eng.clear_keys()  # a step is one whole job: otherwise every later step would see only duplicates
        eng.submit_device(dseq.data_ptr(), qptr, n, R, R)

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    if world > 1:
        # warm-up of the end-of-job exchange too: RCCL sets up its peer-to-peer connections on first use
        if random_mode:
            bcdist.exchange_keys(torch.arange(world * 64, dtype=torch.int64, device=dev))
        else:
            bcdist.reduce_table(torch.ones(world * 4096, dtype=torch.int32, device=dev), dst=0)
    barrier()
    eng.reset()
    eng.timing(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.sync()
    t_steps = time.perf_counter() - t0
    reduce_ms = 0.0
    fixed_counters = None
    if world > 1:
        tr = time.perf_counter()
        if random_mode:
            # set sizes do not add: exchange the keys so that each has one owner, then
            # the per-tuple distinct counts may be summed (done by the root at output time)
            fixed_counters = bcdist.finish_random(eng, dev, dst=0)
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
    counters = fixed_counters if fixed_counters is not None else bcdist.reduce_counters(eng.counters(), dev, dst=0)
    total_reads = n * args.steps * world
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    six = sum(counters[k] for k in ("matched", "constant_region", "sample_barcode", "barcode", "duplicates", "low_quality"))
    assert os.environ.get("BC_ABLATE") or random_mode or six == total_reads == counters["total_reads"], counters
    f_matched = counters["matched"] / max(counters["total_reads"], 1)
    b_alg = workloads.bytes_per_read(w, f_matched)
    avg_ms = kernel_ms / max(launches, 1)
    achieved = (b_alg * n) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0

    # HBM bytes per launch from the PMC counters (FETCH_SIZE doubled per the gfx950 correction,
    # + WRITE_SIZE), measured by tools/scripts/tools_profile.sh in separate rocprofv3 passes of this very workload
    # and committed under profiles/; null when no summary of this workload size exists
    traffic, traffic_src = None, None
    try:
        prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_summary.json"))
        for f in reversed(prof):
            js = json.load(open(os.path.join(ROOT, "profiles", f)))
            wl = js.get("_workload", "")
            if args.config in wl and "{:,}".format(n) in wl and "hbm_traffic_bytes_per_dispatch" in js:
                traffic = js["hbm_traffic_bytes_per_dispatch"]["total"]
                traffic_src = "profiles/" + f
                break
    except OSError:
        pass

    # this box's own streaming ceiling (device-to-device copy of 2 GiB, read + write bytes), measured after
    # the timed region: boxes of the pool differ by up to ~20 %, and this says which kind this run got
    box_copy = None
    try:
        src = dseq[: min(dseq.numel(), 2 << 30)]
        dst = torch.empty_like(src)
        dst.copy_(src)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(5):
            dst.copy_(src)
        torch.cuda.synchronize()
        box_copy = 2.0 * src.numel() * 5 / (time.perf_counter() - tc) / 1e9
        del dst
    except RuntimeError:
        pass

    out = {
        "metric": "reads/sec (whole node), 3x8nt DEL vs 3x1k refs" if args.config != "config5" else "reads/sec (whole node), CRISPR 20nt vs 100k guides",
        "value": total_reads / elapsed,
        "unit": "reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": WORKLOAD_TEXT[args.config], "config": args.config, "reads_per_step_per_gpu": n,
                   "read_len": R, "parallelism": "reads sharded over %d GPU(s); one all-to-all sum of the counter tables over xGMI at the end" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "alg_bytes_per_launch": b_alg * n, "kernel": eng.kernel_name(),
                     "kernel_avg_ms": avg_ms, "launches": launches, "alg_bytes_per_read": b_alg,
                     "kernel_reads_per_s": n / (avg_ms * 1e-3) if avg_ms > 0 else 0.0,
                     "box_copy_GBps": box_copy, "sclk_mhz": sclk,
                     "frac_of_box_copy": (achieved / box_copy) if box_copy else None},
        "outcomes": {k: counters[k] for k in pkg.COUNTER_NAMES},
        "reduce_ms": reduce_ms,
    }
    if not args.no_cpu:
        m = min(args.cpu_sample, n)
        hs = dseq[:m * R].cpu().numpy()
        hq = dqual[:m * R].cpu().numpy()
        threads = max(1, min(os.cpu_count() or 1, 16))
        rate, done, _ = cpu_baseline(w, hs, hq, threads)
        out["cpu_baseline"] = {"value": rate, "unit": "reads/s", "cores": threads, "kind": "port",
                               "sample": "first %d reads of rank 0's batch, CPU oracle (C restatement of parse.rs), %d threads"
                                         % (done, threads)}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
