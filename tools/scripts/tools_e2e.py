#!/usr/bin/env python3
"""perf-debug / DESIGN.md numbers: (1) PCIe-inclusive rate of bc_engine_submit_host, (2) end-to-end
rate of the barcode-count command line on a synthetic FASTQ file (ingest + GPU + writers)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ngs_barcode_count_amd as pkg
import workloads

w = workloads.make("config3")
n = int(os.environ.get("E2E_READS", 8_000_000))
t = time.time(); seq, qual = w.synth.generate_host(0, n); print("generate_host %.1fs" % (time.time() - t), flush=True)
eng = pkg.Engine(w.plan, device=0)
for rep in range(3):
    eng.reset(); eng.sync()
    t = time.time(); eng.submit_host(seq, qual, 100, 100); eng.sync(); dt = time.time() - t
    print("submit_host: %d reads in %.3f s = %.1f M reads/s (%.1f GB/s host->device)" % (n, dt, n / dt / 1e6, n * 200 / dt / 1e9), flush=True)
c = eng.counters()
eng.close()
tmp = "/tmp/bc_e2e"; os.makedirs(tmp, exist_ok=True)
fq = os.path.join(tmp, "reads.fastq")
t = time.time()
with open(fq, "wb") as f:
    s2 = seq.reshape(n, 100); q2 = qual.reshape(n, 100)
    step = 500000
    for a in range(0, n, step):
        b = min(n, a + step)
        rec = np.empty((b - a, 4 + 1 + 100 + 1 + 2 + 100 + 1), dtype=np.uint8)
        rec[:, 0:4] = np.frombuffer(b"@r x", dtype=np.uint8); rec[:, 4] = 10
        rec[:, 5:105] = s2[a:b]; rec[:, 105] = 10; rec[:, 106] = ord("+"); rec[:, 107] = 10
        rec[:, 108:208] = q2[a:b]; rec[:, 208] = 10
        f.write(rec.tobytes())
print("wrote %s (%.2f GB) in %.1fs" % (fq, os.path.getsize(fq) / 1e9, time.time() - t), flush=True)
open(os.path.join(tmp, "scheme.txt"), "w").write(w.scheme + "\n")
open(os.path.join(tmp, "samples.csv"), "w").write("Barcode,Sample_ID\n" + "".join("%s,sample_%d\n" % (s, i) for i, s in enumerate(w.samples)))
open(os.path.join(tmp, "counted.csv"), "w").write("Barcode,ID,N\n" + "".join("%s,bb%d_%d,%d\n" % (s, b + 1, i, b + 1) for b, refs in enumerate(w.counted) for i, s in enumerate(refs)))
cli = os.path.join(ROOT, "ngs-barcode-count_amd", "csrc", "barcode-count")
t = time.time()
res = subprocess.run([cli, "-f", fq, "-q", os.path.join(tmp, "scheme.txt"), "-s", os.path.join(tmp, "samples.csv"), "-c", os.path.join(tmp, "counted.csv"), "-o", tmp, "-p", "e2e", "--min-quality", "20"], capture_output=True, text=True)
dt = time.time() - t
import re; print(re.sub(r"Barcodes counted: [\d,]+\r?\n?", "", res.stdout)[-1500:]); print(res.stderr[-300:])
print("CLI end to end: %d reads in %.2f s = %.2f M reads/s (whole program incl. writing %s rows)" % (n, dt, n / dt / 1e6, c["matched"]))
