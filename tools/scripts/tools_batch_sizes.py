"""throughput of one submit as a function of the batch size (config 3, resident reads)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ngs_barcode_count_amd as pkg
import workloads
w = workloads.make("config3")
R, nmax = w.read_len, 100_000_000
dseq = torch.empty(nmax * R, dtype=torch.uint8, device="cuda"); dqual = torch.empty(nmax * R, dtype=torch.uint8, device="cuda")
w.synth.generate_device(0, None, 0, nmax, dseq.data_ptr(), dqual.data_ptr()); torch.cuda.synchronize()
eng = pkg.Engine(w.plan, device=0)
eng.submit_device(dseq.data_ptr(), dqual.data_ptr(), nmax, R, R); eng.sync()
for n in (65_536, 262_144, 1_000_000, 4_000_000, 16_000_000, 100_000_000):
    reps = max(3, min(200, 200_000_000 // n))
    eng.sync(); t = time.time()
    for _ in range(reps):
        eng.submit_device(dseq.data_ptr(), dqual.data_ptr(), n, R, R)
    eng.sync(); dt = (time.time() - t) / reps
    print("%11d reads per submit: %8.3f ms, %6.2f G reads/s (%s)" % (n, dt * 1e3, n / dt / 1e9, eng.kernel_name()), flush=True)
