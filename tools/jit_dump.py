"""Compiles the scheme-specialised kernel of a workload's plan for gfx950 (no GPU needed), stores it in
the kernel cache next to the library and optionally writes the code object:
    python tools/jit_dump.py config3 [out.co [stride read_len]]"""
import ctypes, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import ngs_barcode_count_amd as bc
import workloads

name = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
stride = int(sys.argv[3]) if len(sys.argv) > 3 else 100
read_len = int(sys.argv[4]) if len(sys.argv) > 4 else stride
w = workloads.make(name)
if out:
    f = w.plan._lib.bc_internal_jit_compile
    f.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p]
    assert f(w.plan._p, stride, read_len, 0, None, out.encode()) == 0, bc._lib.last_error(w.plan._lib)
else:
    bc.precompile(w.plan, stride, read_len)
print("compiled", name, stride, read_len, out or "")
