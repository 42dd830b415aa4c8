"""A/B of engine library variants on ONE box, interleaved: boxes drift by 15 % within a minute, so variants are only
comparable when their launches alternate.  One process, one set of resident reads, one counter table; every round
runs `--steps` launches per variant.
    python tools/ab_bench.py [--config config3] [--reads N] [--rounds 6] [--steps 3] lib_a.so lib_b.so ..."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("BC_JIT", "force")

import torch
import ngs_barcode_count_amd as pkg
from ngs_barcode_count_amd import _lib
import workloads

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="config3")
ap.add_argument("--reads", type=int, default=0)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--fresh", action="store_true", help="reset the engine before every launch: every matched read is a first occurrence")
ap.add_argument("libs", nargs="+")
args = ap.parse_args()
zipf = args.config == "config5z"  # config 5 with Zipf-like guide abundances
if zipf:
    args.config = "config5"
n = args.reads or {"config2": 10_000_000, "config3": 100_000_000, "config4": 50_000_000, "config5": 125_000_000}[args.config]
dev = torch.device("cuda", 0)
variants = []
table = None
dseq = dqual = None
for spec in args.libs:
    # "lib.so" or "lib.so:0x4" (an -DBC_EXPERIMENT build with BC_ABLATE=0x4: phases skipped, counts then differ)
    path, _, ablate = spec.partition(":")
    name = os.path.basename(path).replace(".so", "") + ((":" + ablate) if ablate else "")
    if ablate:
        os.environ["BC_ABLATE"] = ablate
    else:
        os.environ.pop("BC_ABLATE", None)
    os.environ["BC_JIT_CACHE"] = os.path.join(ROOT, "gpurun_out", "ab", "cache_" + name.replace(":", "_"))
    os.makedirs(os.environ["BC_JIT_CACHE"], exist_ok=True)
    lib = _lib.load(os.path.abspath(path))
    w = workloads.make(args.config, lib=lib, n_molecules=n // 2 if args.config == "config4" else None, zipf=zipf)
    R = w.read_len
    if table is None:
        table = torch.zeros(max(w.plan.table_entries, 1), dtype=torch.int32, device=dev)
        dseq = torch.empty(n * R, dtype=torch.uint8, device=dev)
        dqual = torch.empty(n * R, dtype=torch.uint8, device=dev)
        w.synth.generate_device(0, None, 0, n, dseq.data_ptr(), dqual.data_ptr())
        torch.cuda.synchronize()
    eng = pkg.Engine(w.plan, device=0, table_ptr=table.data_ptr() if w.plan.table_entries else None)
    qptr = dqual.data_ptr() if w.min_quality > 0 else None
    if w.plan.random_barcode:
        eng.clear_keys()
    eng.submit_device(dseq.data_ptr(), qptr, n, R, R)  # compiles / loads the specialised kernel
    eng.sync()
    variants.append((name, w, eng, qptr, eng.counters()))
    print("%-24s %s  %s" % (name, eng.kernel_name(), {k: v for k, v in variants[-1][4].items() if v}), flush=True)
ref = variants[0][4]
for name, _, _, _, c in variants:
    assert ":" in name or c == ref, (name, c, ref)  # every variant counts the same (ablated ones aside)
rows = {name: [] for name, *_ in variants}
for r in range(args.rounds):
    order = variants if r % 2 == 0 else variants[::-1]
    for name, w, eng, qptr, _ in order:
        eng.reset()
        eng.sync()
        eng.timing(True)
        for _ in range(args.steps):
            if w.plan.random_barcode:
                eng.clear_keys()
            if args.fresh:
                eng.timing(False)
                eng.reset()
                eng.sync()
                eng.timing(True)
            eng.submit_device(dseq.data_ptr(), qptr, n, w.read_len, w.read_len)
        ms, k = eng.kernel_ms()
        eng.timing(False)
        rows[name].append(ms / k)
print("\nms per launch (%s, %d reads), one column per round:" % (args.config, n))
for name, v in rows.items():
    s = sorted(v)
    print("%-24s min %.3f  med %.3f  | %s" % (name, s[0], s[len(s) // 2], " ".join("%.3f" % x for x in v)))
