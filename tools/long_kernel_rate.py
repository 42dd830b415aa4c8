"""reads/s of the wave-per-read kernel on a Barcode-seq plan with a 40-base raw capture (wide keys), device-resident reads:
    python tools/long_kernel_rate.py [reads]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [root, os.path.join(root, "tests")]
import numpy as np
import torch
import ngs_barcode_count_amd as pkg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
rng = np.random.default_rng(1)
pool = ["".join(rng.choice(list("ACGT"), 40)) for _ in range(50_000)]
R = 100
flank = rng.integers(0, 4, (n, R))
reads = np.frombuffer(b"ACGT", dtype=np.uint8)[flank].copy()
pool_arr = np.frombuffer("".join(pool).encode(), dtype=np.uint8).reshape(len(pool), 40)
pick = rng.integers(0, len(pool), n)
off = rng.integers(0, R - 58 - 1, n)
cons_a, cons_b = np.frombuffer(b"GTACCAGTC", dtype=np.uint8), np.frombuffer(b"TGCATGGAC", dtype=np.uint8)
idx = np.arange(n)
for k in range(9):
    reads[idx, off + k] = cons_a[k]
    reads[idx, off + 49 + k] = cons_b[k]
for k in range(40):
    reads[idx, off + 9 + k] = pool_arr[pick, k]
plan = pkg.Plan("GTACCAGTC{40}TGCATGGAC")
eng = pkg.Engine(plan, device=0)
d = torch.from_numpy(reads.reshape(-1)).cuda()
torch.cuda.synchronize()
eng.submit_device(d.data_ptr(), None, n, R, R)
eng.sync()
eng.reset()
eng.sync()
eng.timing(True)
t0 = time.perf_counter()
eng.submit_device(d.data_ptr(), None, n, R, R)
eng.sync()
dt = time.perf_counter() - t0
ms, k = eng.kernel_ms()
c = eng.counters()
print("kernel %s: %d reads in %.2f ms (kernel %.2f ms) = %.1f M reads/s; matched %d, rows %d, key words %d" % (
    eng.kernel_name(), n, dt * 1e3, ms, n / (ms * 1e-3) / 1e6, c["matched"], eng.finish(), eng._lib.bc_engine_key_words(eng._e)))
