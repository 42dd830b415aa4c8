"""config-3 error model on 150-base reads (the common Illumina length): generic vs specialised kernel.
run on the GPU box: python tools/scripts/tools_len150.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

def run(jit, n=40_000_000, R=150):
    os.environ["BC_JIT"] = jit
    os.environ["BC_JIT_CACHE"] = os.path.join(ROOT, "gpurun_out", "jitc")
    import importlib
    import ngs_barcode_count_amd as pkg
    import workloads
    w = workloads.make("config3", read_len=R)
    dseq = torch.empty(n * R, dtype=torch.uint8, device="cuda")
    dqual = torch.empty(n * R, dtype=torch.uint8, device="cuda")
    w.synth.generate_device(0, None, 0, n, dseq.data_ptr(), dqual.data_ptr())
    torch.cuda.synchronize()
    eng = pkg.Engine(w.plan, device=0)
    for _ in range(2):
        eng.submit_device(dseq.data_ptr(), dqual.data_ptr(), n, R, R)
    eng.sync(); eng.reset(); eng.timing(True)
    for _ in range(6):
        eng.submit_device(dseq.data_ptr(), dqual.data_ptr(), n, R, R)
    ms, launches = eng.kernel_ms()
    c = eng.counters()
    print("R=%d BC_JIT=%s %s: %.3f ms per %d M reads = %.2f G reads/s, %.0f GB/s; matched %.3f" % (
        R, jit, eng.kernel_name(), ms / launches, n // 1_000_000, n / (ms / launches * 1e-3) / 1e9,
        n * (2 * R + 8 * c["matched"] / c["total_reads"]) / (ms / launches * 1e-3) / 1e9, c["matched"] / c["total_reads"]))
    eng.close()
    return c

a = run("0")
b = run("force")
assert a == b, (a, b)
