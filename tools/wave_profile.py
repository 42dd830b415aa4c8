"""Where a wavefront's wall-clock goes, phase by phase (needs a BC_PROFILE build: BC_LIB=build_variants/libprof.so).
usage: BC_LIB=... [BC_JIT=0|force] python tools/wave_profile.py [config] [reads]"""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import torch
import ngs_barcode_count_amd as bc
import workloads

name = sys.argv[1] if len(sys.argv) > 1 else "config3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
w = workloads.make(name, read_len=int(sys.argv[3]) if len(sys.argv) > 3 else 100)
R = w.read_len
dseq = torch.empty(n * R, dtype=torch.uint8, device="cuda")
dqual = torch.empty(n * R, dtype=torch.uint8, device="cuda")
w.synth.generate_device(0, None, 0, n, dseq.data_ptr(), dqual.data_ptr())
torch.cuda.synchronize()
eng = bc.Engine(w.plan, device=0)
lib = eng._lib
lib.bc_internal_profile_read.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
qptr = dqual.data_ptr() if w.min_quality > 0 else None
buf = (ctypes.c_ulonglong * 16)()
for _ in range(2):
    eng.submit_device(dseq.data_ptr(), qptr, n, R, R)
eng.sync()
lib.bc_internal_profile_read(eng._e, buf)
before = list(buf)
steps = 5
for _ in range(steps):
    eng.submit_device(dseq.data_ptr(), qptr, n, R, R)
eng.sync()
lib.bc_internal_profile_read(eng._e, buf)
d = [buf[k] - before[k] for k in range(12)]
names = {1: "wait sequence tile", 2: "pack planes", 3: "locate (anchor/repair)", 4: "wait quality tile",
         5: "quality filter", 6: "captures + LDS lookup + issue gathers", 7: "wait gathers", 8: "verdicts / rest of groups",
         9: "set insert, counters, next fetch, table add"}
tot = sum(d)
print("label: %s jit=%s pipe=%s lhash=%s" % (name, os.environ.get("BC_JIT"), os.environ.get("BC_PIPE"), os.environ.get("BC_LHASH")))
for k in range(1, 10):
    print("  %-45s %6.2f %%" % (names[k], 100.0 * d[k] / max(tot, 1)))
tiles = steps * ((n + 63) // 64)
print("  ticks per wave-tile: %.0f" % (tot / tiles))
