"""Builds and binds tests/emu/libbc_emu.so: the host emulation of the kernel's lane code
(csrc/bc_lane.h) plus the host-only plan code (csrc/bc_plan.cpp).  TEST-ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ngs-barcode-count_amd", "csrc")
SO = os.path.join(ROOT, "tests", "emu", "libbc_emu.so")
# "static": the lane code compiled with the counting form of the scheme-specialised kernels (carry-save accumulation of
# directly extracted windows, bc_lane.h count_mismatches_static), which otherwise only a GPU build would exercise
VARIANTS = {"generic": (SO, []), "static": (SO.replace(".so", "_static.so"), ["-DBC_EMU_STATIC_COUNT=1"])}
SRCS = [os.path.join(ROOT, "tests", "emu", "emu.cpp"), os.path.join(CSRC, "bc_plan.cpp")]
DEPS = SRCS + [os.path.join(CSRC, f) for f in ("bc_lane.h", "bc_intrin.h", "bc_device_plan.h", "bc_plan.hpp")]

_libs = {}


def lib(variant="generic"):
    if variant not in _libs:
        so, flags = VARIANTS[variant]
        if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in DEPS):
            subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wno-unknown-pragmas", "-fPIC", "-shared",
                                   "-o", so] + flags + SRCS)
        import ngs_barcode_count_amd as pkg
        L = C.CDLL(so)
        pkg._lib.declare(L, pkg._lib.PLAN_API)
        L.emu_plan_create.restype = C.c_void_p
        L.emu_plan_create.argtypes = [C.c_void_p]
        L.emu_plan_destroy.argtypes = [C.c_void_p]
        L.emu_table_entries.restype = C.c_uint64
        L.emu_table_entries.argtypes = [C.c_void_p]
        L.emu_discard_counts.argtypes = [C.c_void_p]
        L.emu_process.argtypes = [C.c_void_p] * 4 + [C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.emu_process2.argtypes = [C.c_void_p] * 5 + [C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.emu_rspace.restype = C.c_uint64
        L.emu_rspace.argtypes = [C.c_void_p]
        _libs[variant] = L
    return _libs[variant]


def make_plan(case, variant="generic"):
    """Plan (host-only functions, served by the emu library) for a cases.build_case() dict"""
    import ngs_barcode_count_amd as pkg
    p = pkg.Plan(case["scheme"], lib=lib(variant))
    if case.get("samples"):
        for s, i in case["samples"].items():
            p.add_sample(s, i)
    if case.get("counted"):
        for bi, refs in enumerate(case["counted"]):
            for r in refs:
                p.add_counted(bi, r, r)
    kw = case.get("kwargs", {})
    p.set_max_errors(kw.get("max_sample"), kw.get("max_barcode"), kw.get("max_constant"))
    p.set_min_quality(kw.get("min_quality", 0.0))
    return p


def emulate(plan, seq, qual, lens, stride, read_len, with_random=False, qlens=None):
    """-> (outcomes, dense idx, table entries, discard flag); with_random adds (rcode, rspace): in
    random-barcode mode outcome 0 means "passed every test" -- set membership is the caller's job"""
    L = plan._lib  # the emu variant the plan was made with
    e = L.emu_plan_create(plan._p)
    if not e:
        raise RuntimeError(L.bc_last_error().decode())
    n = seq.size // stride
    outc = np.zeros(n, dtype=np.uint8)
    idx = np.zeros(n, dtype=np.uint64)
    rcode = np.zeros(n, dtype=np.uint64)
    rc = L.emu_process2(e, seq.ctypes.data, qual.ctypes.data if qual is not None else None,
                        lens.ctypes.data if lens is not None else None, qlens.ctypes.data if qlens is not None else None,
                        stride, read_len, n, outc.ctypes.data, idx.ctypes.data, rcode.ctypes.data)
    entries = L.emu_table_entries(e)
    discard = L.emu_discard_counts(e)
    rspace = L.emu_rspace(e)
    L.emu_plan_destroy(e)
    assert rc == 0
    if with_random:
        return outc, idx, entries, discard, rcode, rspace
    return outc, idx, entries, discard
