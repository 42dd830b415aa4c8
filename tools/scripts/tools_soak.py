"""soak: N full-size steps of a config; counters must be exactly N x those of one step (hang / race detector)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ngs_barcode_count_amd as pkg
import workloads
name = sys.argv[1] if len(sys.argv) > 1 else "config3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
n = 50_000_000
w = workloads.make(name)
R = w.read_len
dseq = torch.empty(n * R, dtype=torch.uint8, device="cuda"); dqual = torch.empty(n * R, dtype=torch.uint8, device="cuda")
w.synth.generate_device(0, None, 0, n, dseq.data_ptr(), dqual.data_ptr()); torch.cuda.synchronize()
qptr = dqual.data_ptr() if w.min_quality > 0 else None
eng = pkg.Engine(w.plan, device=0)
eng.submit_device(dseq.data_ptr(), qptr, n, R, R)
one = eng.counters()
eng.reset()
t = time.time()
for i in range(steps):
    if w.plan.random_barcode:
        eng.clear_keys()
    eng.submit_device(dseq.data_ptr(), qptr, n, R, R)
    if i % 100 == 99:
        eng.sync(); print("step", i + 1, "%.1f s" % (time.time() - t), flush=True)
got = eng.counters()
for k, v in one.items():
    exp = v * steps
    assert got[k] == exp, (k, got[k], exp)
print("soak ok:", name, steps, "steps,", eng.kernel_name(), "%.2f G reads/s sustained" % (n * steps / (time.time() - t) / 1e9))
