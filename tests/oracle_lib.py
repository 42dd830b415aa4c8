"""ctypes binding of oracle/liboracle.so (the CPU oracle; test infrastructure only)."""
import ctypes as C
import os
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ODIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ODIR, "liboracle.so")

NAMES = ["matched", "constant_region", "sample_barcode", "barcode", "duplicates", "low_quality"]


def build():
    src = [os.path.join(_ODIR, f) for f in ("oracle.c", "oracle.h")]
    if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _ODIR, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_new.restype = C.c_void_p
        L.orc_new.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.orc_free.argtypes = [C.c_void_p]
        for f in ("orc_format_string", "orc_regions_string", "orc_regex_string"):
            getattr(L, f).restype = C.c_char_p
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("orc_length", "orc_constant_region_length", "orc_barcode_num", "orc_max_constant_errors",
                  "orc_max_sample_errors"):
            getattr(L, f).restype = C.c_uint32
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_barcode_length.restype = C.c_uint32
        L.orc_barcode_length.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_max_barcode_errors.restype = C.c_uint32
        L.orc_max_barcode_errors.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_sample_length.restype = C.c_int32
        L.orc_sample_length.argtypes = [C.c_void_p]
        L.orc_has_random.argtypes = [C.c_void_p]
        L.orc_has_sample.argtypes = [C.c_void_p]
        L.orc_load_sample_csv.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.orc_load_counted_csv.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.orc_add_sample.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.orc_add_counted.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p]
        L.orc_set_max_errors.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_set_min_quality.argtypes = [C.c_void_p, C.c_float]
        L.orc_max_seq_errors.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint16), C.c_uint32, C.c_int,
                                         C.c_uint16, C.POINTER(C.c_uint16)]
        L.orc_begin.argtypes = [C.c_void_p]
        L.orc_process_read.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.orc_process_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                        C.c_uint64]
        L.orc_process_batch_outcomes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                                 C.c_uint32, C.c_uint64, C.c_void_p]
        L.orc_run_reference_threads.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_uint32, C.c_uint32, C.c_uint64]
        L.orc_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.orc_undefined_reads.restype = C.c_uint64
        L.orc_undefined_reads.argtypes = [C.c_void_p]
        L.orc_result_rows.restype = C.c_uint64
        L.orc_result_rows.argtypes = [C.c_void_p]
        L.orc_result_row.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                     C.POINTER(C.c_uint64)]
        L.orc_result_samples.restype = C.c_uint64
        L.orc_result_samples.argtypes = [C.c_void_p]
        L.orc_result_sample.restype = C.c_char_p
        L.orc_result_sample.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_fix_error.restype = C.c_int64
        L.orc_fix_error.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_uint64, C.c_uint16]
        L.orc_sample_id.restype = C.c_char_p
        L.orc_sample_id.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_counted_id.restype = C.c_char_p
        L.orc_counted_id.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p]
        _lib = L
    return _lib


def run_reference_threads(workers, shared, seq, qual, stride, read_len):
    """the reads through the reference's thread structure: 1 reader (this thread) + len(workers) worker threads popping
    a mutex-guarded deque, one mutex-guarded Results (`shared`'s).  Returns the summed outcome counters."""
    L = lib()
    n = seq.size // stride
    arr = (C.c_void_p * len(workers))(*[w._c for w in workers])
    rc = L.orc_run_reference_threads(arr, len(workers), shared._c, seq.ctypes.data,
                                     qual.ctypes.data if qual is not None else None, stride, read_len, n)
    if rc != 0:
        raise RuntimeError("orc_run_reference_threads: could not start the worker threads")
    total = dict.fromkeys(NAMES, 0)
    for w in workers:
        for k, v in w.counters.items():
            total[k] += v
    return total


def fix_error(query, candidates, max_mismatches):
    L = lib()
    arr = (C.c_char_p * len(candidates))(*[c.encode() for c in candidates])
    r = L.orc_fix_error(query.encode(), arr, len(candidates), max_mismatches)
    return None if r < 0 else candidates[r]


def max_seq_errors(sample_errors, sample_size, barcode_errors, barcode_sizes, constant_errors, constant_size):
    L = lib()
    n = len(barcode_sizes)
    sizes = (C.c_uint16 * max(n, 1))(*barcode_sizes)
    out = (C.c_uint16 * (2 + max(n, 1)))()
    opt = lambda v: -1 if v is None else v
    L.orc_max_seq_errors(opt(sample_errors), opt(sample_size), opt(barcode_errors), sizes, n, opt(constant_errors),
                         constant_size, out)
    return out[0], out[1], [out[2 + i] for i in range(n)]


class Oracle:
    """One run of the reference hot path on the CPU oracle."""

    def __init__(self, scheme_text, samples=None, counted=None, max_sample=None, max_barcode=None, max_constant=None,
                 min_quality=0.0, sample_csv=None, counted_csv=None):
        L = lib()
        err = C.create_string_buffer(256)
        t = scheme_text.encode()
        self._c = L.orc_new(t, len(t), err, 256)
        if not self._c:
            raise ValueError(err.value.decode())
        if sample_csv is not None:
            b = sample_csv.encode()
            L.orc_load_sample_csv(self._c, b, len(b))
        if counted_csv is not None:
            b = counted_csv.encode()
            rc = L.orc_load_counted_csv(self._c, b, len(b), err, 256)
            if rc != 0:
                raise ValueError(err.value.decode())
        if samples:
            for s, i in samples.items():
                L.orc_add_sample(self._c, s.encode(), i.encode())
        if counted:
            for bi, d in enumerate(counted):
                for s, i in (d.items() if isinstance(d, dict) else ((x, x) for x in d)):
                    L.orc_add_counted(self._c, bi, s.encode(), i.encode())
        opt = lambda v: -1 if v is None else v
        L.orc_set_max_errors(self._c, opt(max_sample), opt(max_barcode), opt(max_constant))
        L.orc_set_min_quality(self._c, min_quality)
        L.orc_begin(self._c)

    def __del__(self):
        if getattr(self, "_c", None):
            lib().orc_free(self._c)
            self._c = None

    # SequenceFormat fields
    @property
    def format_string(self):
        return lib().orc_format_string(self._c).decode()

    @property
    def regions_string(self):
        return lib().orc_regions_string(self._c).decode()

    @property
    def regex_string(self):
        return lib().orc_regex_string(self._c).decode()

    @property
    def length(self):
        return lib().orc_length(self._c)

    @property
    def constant_region_length(self):
        return lib().orc_constant_region_length(self._c)

    @property
    def barcode_num(self):
        return lib().orc_barcode_num(self._c)

    @property
    def barcode_lengths(self):
        return [lib().orc_barcode_length(self._c, i) for i in range(self.barcode_num)]

    @property
    def sample_length(self):
        v = lib().orc_sample_length(self._c)
        return None if v < 0 else v

    @property
    def budgets(self):
        L = lib()
        return (L.orc_max_constant_errors(self._c), L.orc_max_sample_errors(self._c),
                [L.orc_max_barcode_errors(self._c, i) for i in range(self.barcode_num)])

    def process(self, seq, qual=""):
        s, q = seq.encode(), qual.encode()
        return NAMES[lib().orc_process_read(self._c, s, len(s), q, len(q))]

    def process_batch(self, seq, qual, stride, read_len, lens=None):
        """seq/qual: contiguous uint8 numpy arrays of n*stride bytes"""
        n = seq.size // stride
        lib().orc_process_batch(self._c, seq.ctypes.data, qual.ctypes.data if qual is not None else None,
                                lens.ctypes.data if lens is not None else None, stride, read_len, n)

    def process_batch_outcomes(self, seq, qual, stride, read_len, lens=None, qlens=None):
        """process_batch that also returns the per-read outcome codes (uint8 array, indices into NAMES)"""
        import numpy as np
        n = seq.size // stride
        out = np.zeros(n, dtype=np.uint8)
        lib().orc_process_batch_outcomes(self._c, seq.ctypes.data, qual.ctypes.data if qual is not None else None,
                                         lens.ctypes.data if lens is not None else None,
                                         qlens.ctypes.data if qlens is not None else None, stride, read_len, n,
                                         out.ctypes.data)
        return out

    @property
    def counters(self):
        out = (C.c_uint64 * 6)()
        lib().orc_counters(self._c, out)
        return dict(zip(NAMES, list(out)))

    @property
    def undefined_reads(self):
        return lib().orc_undefined_reads(self._c)

    def rows(self):
        L = lib()
        n = L.orc_result_rows(self._c)
        s, t, c = C.c_char_p(), C.c_char_p(), C.c_uint64()
        out = []
        for i in range(n):
            L.orc_result_row(self._c, i, C.byref(s), C.byref(t), C.byref(c))
            out.append((s.value.decode(), t.value.decode(), c.value))
        return sorted(out)

    def sample_keys(self):
        L = lib()
        return sorted(L.orc_result_sample(self._c, i).decode() for i in range(L.orc_result_samples(self._c)))
