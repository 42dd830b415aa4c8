"""ctypes declarations for include/barcode_count_hip.h (the C ABI of the gfx950 engine)."""
import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(CSRC_DIR, "libbarcode_count_hip.so")

BC_OK = 0
BC_ERR_INVALID, BC_ERR_UNSUPPORTED, BC_ERR_HIP, BC_ERR_NOMEM, BC_ERR_STATE, BC_ERR_COMM = -1, -2, -3, -4, -5, -6
BC_COMM_ID_BYTES = 128
COUNTER_NAMES = ["matched", "constant_region", "sample_barcode", "barcode", "duplicates", "low_quality",
                 "total_reads", "unsupported_reads"]


ROWS_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_uint64, C.c_void_p)  # bc_rows_fn


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("read_len", C.c_uint32), ("p_sub", C.c_uint32), ("p_n", C.c_uint32),
                ("p_lowq", C.c_uint32), ("phred_lo", C.c_uint8), ("phred_hi", C.c_uint8), ("lowq_lo", C.c_uint8),
                ("lowq_hi", C.c_uint8), ("n_molecules", C.c_uint64), ("zipf", C.c_uint32), ("reserved", C.c_uint32), ("geo_total", C.c_uint64)]


_vp, _cp, _u32, _u64, _i32, _int, _sz = C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_int, C.c_size_t

# name -> (restype, argtypes); one entry per function declared in the header
PLAN_API = {
    "bc_version": (_cp, []),
    "bc_last_error": (_cp, []),
    "bc_plan_create": (_vp, [_cp, _sz]),
    "bc_plan_destroy": (None, [_vp]),
    "bc_plan_format_string": (_cp, [_vp]),
    "bc_plan_regions_string": (_cp, [_vp]),
    "bc_plan_regex_string": (_cp, [_vp]),
    "bc_plan_length": (_u32, [_vp]),
    "bc_plan_constant_region_length": (_u32, [_vp]),
    "bc_plan_barcode_num": (_u32, [_vp]),
    "bc_plan_barcode_length": (_u32, [_vp, _u32]),
    "bc_plan_sample_length": (_i32, [_vp]),
    "bc_plan_has_random": (_int, [_vp]),
    "bc_plan_has_sample": (_int, [_vp]),
    "bc_plan_load_sample_csv": (_int, [_vp, _cp, _sz]),
    "bc_plan_load_counted_csv": (_int, [_vp, _cp, _sz]),
    "bc_plan_add_sample": (_int, [_vp, _cp, _cp]),
    "bc_plan_add_counted": (_int, [_vp, _u32, _cp, _cp]),
    "bc_plan_n_samples": (_u32, [_vp]),
    "bc_plan_sample_seq": (_cp, [_vp, _u32]),
    "bc_plan_sample_id": (_cp, [_vp, _u32]),
    "bc_plan_n_counted": (_u32, [_vp, _u32]),
    "bc_plan_counted_seq": (_cp, [_vp, _u32, _u32]),
    "bc_plan_counted_id": (_cp, [_vp, _u32, _u32]),
    "bc_plan_set_max_errors": (_int, [_vp, _int, _int, _int]),
    "bc_plan_max_constant_errors": (_u32, [_vp]),
    "bc_plan_max_sample_errors": (_u32, [_vp]),
    "bc_plan_max_barcode_errors": (_u32, [_vp, _u32]),
    "bc_max_seq_errors": (None, [_int, _int, _int, C.POINTER(C.c_uint16), _u32, _int, C.c_uint16,
                                 C.POINTER(C.c_uint16)]),
    "bc_plan_set_min_quality": (_int, [_vp, C.c_float]),
    "bc_plan_quality_threshold": (_u32, [_vp, _u32]),
    "bc_plan_table_entries": (_u64, [_vp]),
    "bc_plan_mode": (_int, [_vp]),
}
ENGINE_API = {
    "bc_engine_create": (_vp, [_vp, _int, _vp, _vp]),
    "bc_engine_destroy": (None, [_vp]),
    "bc_engine_submit_device": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u64]),
    "bc_engine_submit_device_q": (_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u64]),
    "bc_engine_hip_stream": (_vp, [_vp]),
    "bc_engine_device": (_int, [_vp]),
    "bc_engine_submit_host": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u64]),
    "bc_engine_sync": (_int, [_vp]),
    "bc_engine_reset": (_int, [_vp]),
    "bc_engine_reset_results": (_int, [_vp]),
    "bc_engine_counters": (_int, [_vp, C.POINTER(C.c_uint64)]),
    "bc_engine_table_ptr": (_vp, [_vp]),
    "bc_engine_counters_ptr": (_vp, [_vp]),
    "bc_engine_table_entries": (_u64, [_vp]),
    "bc_engine_finish": (_int, [_vp, C.POINTER(C.c_uint64)]),
    "bc_engine_rows": (_int, [_vp, _u64, _u64, _vp, _vp, _vp]),
    "bc_engine_finish_stream": (_int, [_vp, _vp, _vp, C.POINTER(C.c_uint64)]),
    "bc_engine_nonzero_entries": (_int, [_vp, C.POINTER(C.c_uint64)]),
    "bc_engine_decode_index": (_int, [_vp, _u64, C.POINTER(_u32), C.POINTER(_u32)]),
    "bc_engine_timing": (_int, [_vp, _int]),
    "bc_engine_kernel_ms": (_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "bc_engine_kernel_ms_each": (_int, [_vp, C.POINTER(C.c_double), _u64, C.POINTER(C.c_uint64)]),
    "bc_table_pack_u8": (_int, [_vp, _u64, _vp, _vp, _vp, _u64, C.POINTER(_u64), _int, _vp]),
    "bc_table_sum_u8": (_int, [_vp, _u32, _u64, _vp, _int, _vp]),
    "bc_engine_kernel_name": (_cp, [_vp]),
    "bc_engine_sclk_mhz": (_int, [_vp, C.POINTER(C.c_double)]),
    "bc_plan_precompile": (_int, [_vp, _u32, _u32, _int, _cp]),  # lives with the engine: it drives the device compiler
    "bc_engine_trace": (_int, [_vp, _vp, _vp]),
    "bc_engine_row_text": (_int, [_vp, _u64, _cp, _sz, _cp, _sz, C.POINTER(C.c_uint64)]),
    "bc_engine_key_words": (_u32, [_vp]),
    "bc_engine_key_count": (_int, [_vp, C.POINTER(C.c_uint64)]),
    "bc_engine_export_keys": (_int, [_vp, _vp, _u64, C.POINTER(C.c_uint64)]),
    "bc_engine_import_keys": (_int, [_vp, _vp, _u64, C.POINTER(C.c_uint64)]),
    "bc_engine_clear_keys": (_int, [_vp]),
    "bc_probe_atomic_rate": (_int, [_int, _vp, _u64, _u64, C.POINTER(C.c_double)]),
    "bc_engine_materialize_table": (_int, [_vp]),
    "bc_engine_export_counts": (_int, [_vp, _vp, _vp, _u64, C.POINTER(_u64)]),
    "bc_engine_import_counts": (_int, [_vp, _vp, _vp, _u64]),
    "bc_comm_unique_id": (_int, [_vp]),
    "bc_comm_create": (_vp, [_vp, _int, _int, _int]),
    "bc_comm_create_host": (_vp, [_cp, _int, _int]),
    "bc_comm_destroy": (None, [_vp]),
    "bc_comm_rank": (_int, [_vp]),
    "bc_comm_world": (_int, [_vp]),
    "bc_comm_barrier": (_int, [_vp]),
    "bc_engine_reduce_all": (_int, [_vp, _vp, _int, C.POINTER(C.c_uint64)]),
    "bc_engine_finish_all": (_int, [_vp, _vp, _int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "bc_engine_plan": (_vp, [_vp]),
    "bc_fix_error": (C.c_int64, [_cp, C.POINTER(_cp), _u64, C.c_uint16, _int]),
    "bc_fastq_count": (_int, [_vp, _cp, C.POINTER(C.c_uint64), _vp, _vp]),
    "bc_fastq_count_shard": (_int, [_vp, _cp, _u32, _u32, C.POINTER(C.c_uint64), _vp, _vp]),
    "bc_fastq_record_start": (_int, [_cp, _u64, C.POINTER(C.c_uint64)]),
    "bc_comm_sum_u64": (_int, [_vp, C.POINTER(C.c_uint64), _int, _int]),
    "bc_synth_create": (_vp, [_vp, C.POINTER(SynthParams)]),
    "bc_synth_destroy": (None, [_vp]),
    "bc_synth_generate_host": (_int, [_vp, _u64, _u64, _vp, _vp, _u32]),
    "bc_synth_generate_device": (_int, [_vp, _int, _vp, _u64, _u64, _vp, _vp, _u32]),
    "bc_synth_make_set": (_int, [_u64, _u32, _u32, _u32, C.c_char_p]),
}


def declare(lib, api):
    for name, (res, args) in api.items():
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = args
    return lib


_lib = None


def load(path=None):
    """Loads the HIP engine library.  There is no CPU fallback: a missing library is an error."""
    global _lib
    if path is None and _lib is not None:
        return _lib
    p = path or os.environ.get("BC_LIB") or LIB_PATH  # BC_LIB: perf experiments with variant builds
    if not os.path.exists(p):
        raise RuntimeError("HIP engine library not built: %s (run `python -c 'import __graft_entry__ as g; "
                           "g.build()'` or `make -C %s`)" % (p, CSRC_DIR))
    # PyTorch-ROCm ships its own copy of the HIP runtime; whichever copy a process loads SECOND finds no GPU.  This
    # library links the system's (/opt/rocm): when torch is installed but not imported yet, import it first, so that the
    # one runtime both use is torch's.  (A process that never imports torch is unaffected; BC_NO_TORCH_PRELOAD=1 skips.)
    import sys
    if "torch" not in sys.modules and not os.environ.get("BC_NO_TORCH_PRELOAD"):
        import importlib.util
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    lib = C.CDLL(p)
    declare(lib, PLAN_API)
    declare(lib, ENGINE_API)
    if path is None:
        _lib = lib
    return lib


def last_error(lib):
    return lib.bc_last_error().decode(errors="replace")
