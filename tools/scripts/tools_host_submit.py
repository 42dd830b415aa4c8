"""PCIe-inclusive rate of bc_engine_submit_host (pageable host arrays -> pinned staging -> H2D -> kernel)
for a few staging-thread counts.  python tools/scripts/tools_host_submit.py"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1:
    import ngs_barcode_count_amd as pkg
    import workloads
    w = workloads.make("config3")
    n = 16_000_000
    seq, qual = w.synth.generate_host(0, n)
    eng = pkg.Engine(w.plan, device=0)
    best = 0.0
    for rep in range(4):
        eng.reset(); eng.sync()
        t = time.time(); eng.submit_host(seq, qual, 100, 100); eng.sync(); dt = time.time() - t
        best = max(best, n / dt)
    print("BC_STAGE_THREADS=%s: %.1f M reads/s (%.1f GB/s host->device), kernel %s" % (
        os.environ.get("BC_STAGE_THREADS"), best / 1e6, best * 200 / 1e9, eng.kernel_name()), flush=True)
else:
    for t in ("1", "2", "4", "8"):
        subprocess.call([sys.executable, __file__, "child"], env=dict(os.environ, BC_STAGE_THREADS=t))
