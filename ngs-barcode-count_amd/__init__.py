"""ngs-barcode-count_amd -- MI355X (gfx950) engine for the per-read match/count path of
Roco-scientist/NGS-Barcode-Count, behind the C ABI of include/barcode_count_hip.h.

Import name: `ngs_barcode_count_amd` (the shim module at the repo root maps it onto this
directory, whose name is not a Python identifier).

Two layers:
  * Plan / Engine / Synth -- thin ctypes wrappers of the C ABI;
  * SequenceFormat, BarcodeConversions, MaxSeqErrors, SequenceErrors, Results, SequenceParser --
    host-side mirror of the reference's own types for this path (src/info.rs, src/parse.rs), same
    names and argument meaning, so callers and tests read like the reference.
There is no CPU implementation in this package: without the HIP library everything raises.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import COUNTER_NAMES, SynthParams

__all__ = ["Plan", "Engine", "Comm", "Synth", "SequenceFormat", "BarcodeConversions", "MaxSeqErrors", "SequenceErrors",
           "Results", "SequenceParser", "fix_error", "BarcodeCountError", "COUNTER_NAMES"]


class BarcodeCountError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (status %d)" % (msg, code))
        self.code = code


def _check(lib, rc):
    if rc != 0:
        raise BarcodeCountError(rc, _lib.last_error(lib))


def _opt(v):
    return -1 if v is None else int(v)


class Plan:
    """Compiled scheme + known barcode sets + error budgets (bc_plan)."""

    def __init__(self, scheme_text, lib=None):
        self._lib = lib or _lib.load()
        b = scheme_text.encode()
        self._p = self._lib.bc_plan_create(b, len(b))
        if not self._p:
            raise BarcodeCountError(_lib.BC_ERR_INVALID, _lib.last_error(self._lib))

    def __del__(self):
        if getattr(self, "_p", None):
            self._lib.bc_plan_destroy(self._p)
            self._p = None

    # SequenceFormat fields (info.rs:176-187)
    format_string = property(lambda s: s._lib.bc_plan_format_string(s._p).decode())
    regions_string = property(lambda s: s._lib.bc_plan_regions_string(s._p).decode())
    regex_string = property(lambda s: s._lib.bc_plan_regex_string(s._p).decode())
    length = property(lambda s: s._lib.bc_plan_length(s._p))
    constant_region_length = property(lambda s: s._lib.bc_plan_constant_region_length(s._p))
    barcode_num = property(lambda s: s._lib.bc_plan_barcode_num(s._p))
    barcode_lengths = property(lambda s: [s._lib.bc_plan_barcode_length(s._p, i) for i in range(s.barcode_num)])
    random_barcode = property(lambda s: bool(s._lib.bc_plan_has_random(s._p)))
    sample_barcode = property(lambda s: bool(s._lib.bc_plan_has_sample(s._p)))

    @property
    def sample_length_option(self):
        v = self._lib.bc_plan_sample_length(self._p)
        return None if v < 0 else v

    def load_sample_csv(self, text):
        b = text.encode()
        _check(self._lib, self._lib.bc_plan_load_sample_csv(self._p, b, len(b)))

    def load_counted_csv(self, text):
        b = text.encode()
        _check(self._lib, self._lib.bc_plan_load_counted_csv(self._p, b, len(b)))

    def add_sample(self, seq, sample_id):
        _check(self._lib, self._lib.bc_plan_add_sample(self._p, seq.encode(), sample_id.encode()))

    def add_counted(self, barcode_index, seq, barcode_id):
        _check(self._lib, self._lib.bc_plan_add_counted(self._p, barcode_index, seq.encode(), barcode_id.encode()))

    def samples(self):
        """[(sequence, id)] in engine index order"""
        L, p = self._lib, self._p
        return [(L.bc_plan_sample_seq(p, i).decode(), L.bc_plan_sample_id(p, i).decode())
                for i in range(L.bc_plan_n_samples(p))]

    def counted(self, barcode_index):
        L, p = self._lib, self._p
        return [(L.bc_plan_counted_seq(p, barcode_index, i).decode(), L.bc_plan_counted_id(p, barcode_index, i).decode())
                for i in range(L.bc_plan_n_counted(p, barcode_index))]

    def set_max_errors(self, sample_errors=None, barcode_errors=None, constant_errors=None):
        _check(self._lib, self._lib.bc_plan_set_max_errors(self._p, _opt(sample_errors), _opt(barcode_errors),
                                                           _opt(constant_errors)))

    def set_min_quality(self, q):
        _check(self._lib, self._lib.bc_plan_set_min_quality(self._p, float(q)))

    max_constant_errors = property(lambda s: s._lib.bc_plan_max_constant_errors(s._p))
    max_sample_errors = property(lambda s: s._lib.bc_plan_max_sample_errors(s._p))
    max_barcode_errors = property(lambda s: [s._lib.bc_plan_max_barcode_errors(s._p, i)
                                             for i in range(s.barcode_num)])

    def quality_threshold(self, run_len):
        return self._lib.bc_plan_quality_threshold(self._p, run_len)

    @property
    def mode(self):
        """"dense" (counter table) or "sparse" (hash map of tuple keys: some barcode is kept raw)"""
        m = self._lib.bc_plan_mode(self._p)
        if m == 0:
            raise BarcodeCountError(_lib.BC_ERR_UNSUPPORTED, _lib.last_error(self._lib))
        return "dense" if m == 1 else "sparse"

    @property
    def table_entries(self):
        if self.mode == "sparse":
            return 0
        return self._lib.bc_plan_table_entries(self._p)


class Engine:
    """One run on one GPU (bc_engine).  `table` may be a caller-owned zeroed device buffer of
    plan.table_entries u32 (e.g. a torch tensor's data_ptr()) so it can be reduced with RCCL."""

    def __init__(self, plan, device=0, stream=None, table_ptr=None):
        self._lib = plan._lib
        self.plan = plan
        self._e = self._lib.bc_engine_create(plan._p, int(device), stream, table_ptr)
        if not self._e:
            raise BarcodeCountError(_lib.BC_ERR_HIP, _lib.last_error(self._lib))

    def close(self):
        if getattr(self, "_e", None):
            self._lib.bc_engine_destroy(self._e)
            self._e = None

    __del__ = close

    def submit_device(self, d_seq, d_qual, n_reads, stride, read_len, d_lens=None):
        _check(self._lib, self._lib.bc_engine_submit_device(self._e, d_seq, d_qual, d_lens, stride, read_len, n_reads))

    def submit_device_q(self, d_seq, d_qual, n_reads, stride, d_lens, d_qlens):
        """records whose quality line differs in length from the sequence line: both length arrays (u16, device)"""
        _check(self._lib, self._lib.bc_engine_submit_device_q(self._e, d_seq, d_qual, d_lens, d_qlens, stride, n_reads))

    def submit_host(self, seq, qual, stride, read_len, lens=None):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        n = seq.size // stride
        qp = None
        if qual is not None:
            qual = np.ascontiguousarray(qual, dtype=np.uint8)
            qp = qual.ctypes.data
        lp = None
        if lens is not None:
            lens = np.ascontiguousarray(lens, dtype=np.uint16)
            lp = lens.ctypes.data
        _check(self._lib, self._lib.bc_engine_submit_host(self._e, seq.ctypes.data, qp, lp, stride, read_len, n))

    def count_fastq(self, path, shard=0, n_shards=1):
        """input::read_fastq + the workers (input.rs:24-89, parse.rs:53-76): reads a .fastq / .fastq.gz
        file and counts it; returns the reference's "Total sequences" value"""
        n = C.c_uint64()
        if n_shards == 1:
            _check(self._lib, self._lib.bc_fastq_count(self._e, str(path).encode(), C.byref(n), None, None))
        else:  # this GPU's share of the file's records (bc_fastq_count_shard)
            _check(self._lib, self._lib.bc_fastq_count_shard(self._e, str(path).encode(), shard, n_shards, C.byref(n), None, None))
        return n.value

    def sync(self):
        _check(self._lib, self._lib.bc_engine_sync(self._e))

    def reset(self):
        _check(self._lib, self._lib.bc_engine_reset(self._e))

    def reset_results(self):
        """a fresh Results (table / bit map / keys); the outcome counters go on"""
        _check(self._lib, self._lib.bc_engine_reset_results(self._e))

    def counters(self):
        out = (C.c_uint64 * 8)()
        _check(self._lib, self._lib.bc_engine_counters(self._e, out))
        return dict(zip(COUNTER_NAMES, [int(x) for x in out]))

    table_ptr = property(lambda s: s._lib.bc_engine_table_ptr(s._e))
    counters_ptr = property(lambda s: s._lib.bc_engine_counters_ptr(s._e))
    table_entries = property(lambda s: s._lib.bc_engine_table_entries(s._e))

    def trace(self, d_outcome_u8, d_index_u64):
        """per-read outcome / dense index written by the next submits (tests); None disables"""
        _check(self._lib, self._lib.bc_engine_trace(self._e, d_outcome_u8, d_index_u64))

    # random-barcode mode: the set of (tuple, random barcode) keys
    def key_count(self):
        n = C.c_uint64()
        _check(self._lib, self._lib.bc_engine_key_count(self._e, C.byref(n)))
        return n.value

    def export_keys(self, d_keys, capacity):
        n = C.c_uint64()
        _check(self._lib, self._lib.bc_engine_export_keys(self._e, d_keys, capacity, C.byref(n)))
        return n.value

    def import_keys(self, d_keys, n):
        new = C.c_uint64()
        _check(self._lib, self._lib.bc_engine_import_keys(self._e, d_keys, n, C.byref(new)))
        return new.value

    def clear_keys(self):
        _check(self._lib, self._lib.bc_engine_clear_keys(self._e))

    def materialize_table(self):
        """random-barcode plans with a dense table: per-tuple distinct counts of the current key set -> the table"""
        _check(self._lib, self._lib.bc_engine_materialize_table(self._e))

    def export_counts(self, d_keys, d_counts, capacity):
        """(key, count) pairs of a raw-key plan's map -> device buffers; None buffers: only their number"""
        n = C.c_uint64()
        _check(self._lib, self._lib.bc_engine_export_counts(self._e, d_keys, d_counts, capacity, C.byref(n)))
        return n.value

    def import_counts(self, d_keys, d_counts, n):
        _check(self._lib, self._lib.bc_engine_import_counts(self._e, d_keys, d_counts, n))

    def timing(self, enable=True):
        _check(self._lib, self._lib.bc_engine_timing(self._e, 1 if enable else 0))

    def kernel_name(self):
        return self._lib.bc_engine_kernel_name(self._e).decode()

    def sclk_mhz(self):
        """shader clock right now (0.3 ms probe kernel)"""
        v = C.c_double()
        _check(self._lib, self._lib.bc_engine_sclk_mhz(self._e, C.byref(v)))
        return v.value

    def kernel_ms(self):
        ms, n = C.c_double(), C.c_uint64()
        _check(self._lib, self._lib.bc_engine_kernel_ms(self._e, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_ms_each(self):
        """per-launch times (ms) of the match kernel since the last kernel_ms(), in launch order"""
        n = C.c_uint64()
        _check(self._lib, self._lib.bc_engine_kernel_ms_each(self._e, None, 0, C.byref(n)))
        out = (C.c_double * max(n.value, 1))()
        _check(self._lib, self._lib.bc_engine_kernel_ms_each(self._e, out, n.value, C.byref(n)))
        return [float(out[i]) for i in range(n.value)]

    def nonzero_entries(self):
        """rows finish() would produce now (dense plans, no random barcode): one sweep of the table"""
        n = C.c_uint64()
        _check(self._lib, self._lib.bc_engine_nonzero_entries(self._e, C.byref(n)))
        return n.value

    def finish_stream(self, on_rows):
        """bc_engine_finish_stream: on_rows(keys uint64 array, counts uint32 array) per chunk (copies of the chunk);
        returns the number of rows.  Host and device memory stay bounded whatever the table holds."""
        err = []

        def cb(kp, cp, n, _user):
            try:
                on_rows(np.ctypeslib.as_array(kp, shape=(n,)).copy(), np.ctypeslib.as_array(cp, shape=(n,)).copy())
                return 0
            except Exception as ex:  # never let an exception cross the C frames
                err.append(ex)
                return 1

        fn = _lib.ROWS_FN(cb)
        n = C.c_uint64()
        rc = self._lib.bc_engine_finish_stream(self._e, fn, None, C.byref(n))
        if err:
            raise err[0]
        _check(self._lib, rc)
        return n.value

    def reduce_all(self, comm, root=0):
        """bc_engine_reduce_all: the job's one exchange (collective over comm); -> the job's counters on the root,
        zeros elsewhere.  Afterwards the root's engine holds the job's table / key set / key map."""
        out = (C.c_uint64 * 8)()
        _check(self._lib, self._lib.bc_engine_reduce_all(self._e, comm._c if comm is not None else None, root, out))
        return dict(zip(COUNTER_NAMES, [int(x) for x in out]))

    def finish_all(self, comm, root=0):
        """bc_engine_finish_all = reduce_all + finish on the root; -> (counters, number of rows)"""
        out = (C.c_uint64 * 8)()
        n = C.c_uint64()
        _check(self._lib, self._lib.bc_engine_finish_all(self._e, comm._c if comm is not None else None, root, out, C.byref(n)))
        return dict(zip(COUNTER_NAMES, [int(x) for x in out])), n.value

    def finish(self):
        """bc_engine_finish: compacts the results into sparse rows; returns their number"""
        n = C.c_uint64()
        _check(self._lib, self._lib.bc_engine_finish(self._e, C.byref(n)))
        return n.value

    def rows(self):
        """-> (sample_idx[n], barcode_idx[n, barcode_num], count[n]) of the non-zero table entries"""
        n = C.c_uint64()
        _check(self._lib, self._lib.bc_engine_finish(self._e, C.byref(n)))
        n = n.value
        nb = max(self.plan.barcode_num, 1)
        s = np.zeros(n, dtype=np.uint32)
        b = np.zeros((n, nb), dtype=np.uint32)
        c = np.zeros(n, dtype=np.uint64)
        if n:
            _check(self._lib, self._lib.bc_engine_rows(self._e, 0, n, s.ctypes.data, b.ctypes.data, c.ctypes.data))
        return s, b[:, :self.plan.barcode_num], c

    def result_rows(self):
        """sorted [(sample key, "b1,b2,..", count)] with sequences as keys, like Results (info.rs:661-665)"""
        if self.plan.mode == "sparse":
            n = C.c_uint64()
            _check(self._lib, self._lib.bc_engine_finish(self._e, C.byref(n)))
            sb, tb, cnt = C.create_string_buffer(64), C.create_string_buffer(1024), C.c_uint64()
            out = []
            for i in range(n.value):
                _check(self._lib, self._lib.bc_engine_row_text(self._e, i, sb, 64, tb, 1024, C.byref(cnt)))
                out.append((sb.value.decode(), tb.value.decode(), int(cnt.value)))
            return sorted(out)
        s, b, c = self.rows()
        samples = [x for x, _ in self.plan.samples()] if self.plan.sample_barcode else ["barcode"]
        sets = [[x for x, _ in self.plan.counted(i)] for i in range(self.plan.barcode_num)]
        out = []
        for i in range(len(c)):
            out.append((samples[s[i]], ",".join(sets[j][b[i, j]] for j in range(len(sets))), int(c[i])))
        return sorted(out)


class Comm:
    """The communicator of a multi-GPU job (bc_comm): one rank per GPU, ONE exchange at the job's end.
    Comm.rccl(id, rank, world, device): RCCL over xGMI -- rank 0 makes `id` with Comm.unique_id() and hands the bytes to
    the others (a file, a pipe, torch.distributed ...).  Comm.host(dir, rank, world): message files in a directory every
    rank can write to (any processes of one machine, several ranks on one GPU included)."""

    def __init__(self, handle, lib):
        self._c, self._lib = handle, lib

    @staticmethod
    def unique_id(lib=None):
        lib = lib or _lib.load()
        buf = C.create_string_buffer(_lib.BC_COMM_ID_BYTES)
        _check(lib, lib.bc_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def rccl(cls, unique_id, rank, world, device, lib=None):
        lib = lib or _lib.load()
        assert len(unique_id) == _lib.BC_COMM_ID_BYTES
        h = lib.bc_comm_create(C.create_string_buffer(bytes(unique_id), _lib.BC_COMM_ID_BYTES), rank, world, int(device))
        if not h:
            raise BarcodeCountError(_lib.BC_ERR_COMM, _lib.last_error(lib))
        return cls(h, lib)

    @classmethod
    def host(cls, directory, rank, world, lib=None):
        lib = lib or _lib.load()
        h = lib.bc_comm_create_host(str(directory).encode(), rank, world)
        if not h:
            raise BarcodeCountError(_lib.BC_ERR_COMM, _lib.last_error(lib))
        return cls(h, lib)

    rank = property(lambda s: s._lib.bc_comm_rank(s._c))
    world = property(lambda s: s._lib.bc_comm_world(s._c))

    def barrier(self):
        _check(self._lib, self._lib.bc_comm_barrier(self._c))

    def sum_u64(self, values, root=0):
        """element-wise sum of a list of ints onto the root (the other ranks get their own values back)"""
        arr = (C.c_uint64 * len(values))(*values)
        _check(self._lib, self._lib.bc_comm_sum_u64(self._c, arr, len(values), root))
        return [int(x) for x in arr]

    def close(self):
        if getattr(self, "_c", None):
            self._lib.bc_comm_destroy(self._c)
            self._c = None

    __del__ = close


class Synth:
    """Counter-based synthetic read generator; identical on host and device."""

    def __init__(self, plan, seed, read_len=100, p_sub=0.0, p_n=0.0, p_lowq=0.0, phred=(30, 40), lowq=(2, 15),
                 n_molecules=0, zipf=False, geo_total=0):
        self._lib = plan._lib
        self.plan = plan
        f = lambda p: min(int(round(p * 4294967296.0)), 4294967295)
        self.params = SynthParams(seed, read_len, f(p_sub), f(p_n), f(p_lowq), phred[0], phred[1], lowq[0], lowq[1],
                                  n_molecules, 1 if zipf else 0, 0, geo_total)
        self._s = self._lib.bc_synth_create(plan._p, C.byref(self.params))
        if not self._s:
            raise BarcodeCountError(_lib.BC_ERR_INVALID, _lib.last_error(self._lib))

    def __del__(self):
        if getattr(self, "_s", None):
            self._lib.bc_synth_destroy(self._s)
            self._s = None

    def generate_host(self, first, n, stride=None):
        stride = stride or self.params.read_len
        seq = np.empty(n * stride, dtype=np.uint8)
        qual = np.empty(n * stride, dtype=np.uint8)
        _check(self._lib, self._lib.bc_synth_generate_host(self._s, first, n, seq.ctypes.data, qual.ctypes.data, stride))
        return seq, qual

    def generate_device(self, device, stream, first, n, d_seq, d_qual, stride=None):
        stride = stride or self.params.read_len
        _check(self._lib, self._lib.bc_synth_generate_device(self._s, device, stream, first, n, d_seq, d_qual, stride))


def make_set(seed, n, k, min_dist=1, lib=None):
    lib = lib or _lib.load()
    buf = C.create_string_buffer(n * (k + 1))
    rc = lib.bc_synth_make_set(seed, n, k, min_dist, buf)
    _check(lib, rc)
    raw = buf.raw
    return [raw[i * (k + 1):i * (k + 1) + k].decode() for i in range(n)]


def probe_atomic_rate(device, table_ptr, entries, n_atomics, lib=None):
    """random no-return atomic adds of 0 per second over a u32 table on this device (box diagnostic)"""
    lib = lib or _lib.load()
    v = C.c_double()
    _check(lib, lib.bc_probe_atomic_rate(int(device), table_ptr, entries, n_atomics, C.byref(v)))
    return v.value


def precompile(plan, stride=100, read_len=None, lens=False, cache_dir=None):
    """Builds the scheme-specialised match kernel of a plan for one batch shape ahead of time (no GPU needed) into
    the kernel cache (default: jit_cache/ next to the library), so that engines of this plan find it ready.
    stride / read_len as they will be passed to submit_*; lens = per-read lengths will be passed."""
    lib = plan._lib
    _check(lib, lib.bc_plan_precompile(plan._p, stride, stride if read_len is None else read_len, 1 if lens else 0,
                                       cache_dir.encode() if cache_dir else None))


def fix_error(mismatch_seq, possible_seqs, mismatches, device=0, lib=None):
    """fix_error (parse.rs:553-593) on the GPU: the unique nearest candidate or None."""
    lib = lib or _lib.load()
    possible_seqs = list(possible_seqs)
    arr = (C.c_char_p * max(len(possible_seqs), 1))(*[s.encode() for s in possible_seqs])
    r = lib.bc_fix_error(mismatch_seq.encode(), arr, len(possible_seqs), mismatches, device)
    if r < -1:
        raise BarcodeCountError(int(r) + 1, _lib.last_error(lib))
    return None if r < 0 else possible_seqs[r]


# ------------------------------------------------------------------------------------------------
# host-side mirror of the reference's types for this path
# ------------------------------------------------------------------------------------------------
class SequenceFormat:
    """info.rs:176-310"""

    def __init__(self, plan):
        self.plan = plan
        for f in ("format_string", "regions_string", "length", "constant_region_length", "barcode_num",
                  "barcode_lengths", "sample_length_option", "random_barcode", "sample_barcode"):
            setattr(self, f, getattr(plan, f))

    @classmethod
    def parse_format_file(cls, format_path):
        with open(format_path) as fh:
            return cls(Plan(fh.read()))

    @classmethod
    def from_text(cls, text):
        return cls(Plan(text))

    def __str__(self):  # Display, info.rs:313-335
        key, seen = "", set()
        names = {"S": "\nS: Sample barcode", "B": "\nB: Counted barcode", "C": "\nC: Constant region",
                 "R": "\nR: Random barcode"}
        for ch in self.regions_string:
            if ch not in seen:
                seen.add(ch)
                key += names.get(ch, "")
        return "-FORMAT-\n%s\n%s%s" % (self.format_string, self.regions_string, key)


class BarcodeConversions:
    """info.rs:338-457; the sets live in the plan of the SequenceFormat they belong to"""

    def __init__(self, sequence_format):
        self.plan = sequence_format.plan

    def sample_barcode_file_conversion(self, barcode_path):
        with open(barcode_path) as fh:
            self.plan.load_sample_csv(fh.read())

    def barcode_file_conversion(self, barcode_path, barcode_num=None):
        with open(barcode_path) as fh:
            self.plan.load_counted_csv(fh.read())

    samples_barcode_hash = property(lambda s: dict(s.plan.samples()))
    sample_seqs = property(lambda s: {x for x, _ in s.plan.samples()})
    counted_barcodes_hash = property(lambda s: [dict(s.plan.counted(i)) for i in range(s.plan.barcode_num)])
    counted_barcode_seqs = property(lambda s: [{x for x, _ in s.plan.counted(i)} for i in range(s.plan.barcode_num)])


class MaxSeqErrors:
    """info.rs:461-616"""

    def __init__(self, sample_errors_option, sample_barcode_size_option, barcode_errors_option, barcode_sizes,
                 constant_errors_option, constant_region_size, min_quality, lib=None):
        lib = lib or _lib.load()
        n = len(barcode_sizes)
        sizes = (C.c_uint16 * max(n, 1))(*barcode_sizes)
        out = (C.c_uint16 * (2 + max(n, 1)))()
        lib.bc_max_seq_errors(_opt(sample_errors_option), _opt(sample_barcode_size_option), _opt(barcode_errors_option),
                              sizes, n, _opt(constant_errors_option), constant_region_size, out)
        self._c, self._s, self._b = out[0], out[1], [out[2 + i] for i in range(n)]
        self.options = (sample_errors_option, barcode_errors_option, constant_errors_option)
        self.min_quality = min_quality

    def max_constant_errors(self):
        return self._c

    def max_sample_errors(self):
        return self._s

    def max_barcode_errors(self):
        return list(self._b)


class SequenceErrors:
    """info.rs:16-172; filled from the engine's device counters"""
    FIELDS = ["matched", "constant_region", "sample_barcode", "barcode", "duplicates", "low_quality"]

    def __init__(self):
        for f in self.FIELDS:
            setattr(self, f, 0)

    def update(self, counters):
        for f in self.FIELDS:
            setattr(self, f, counters[f] & 0xFFFFFFFF)  # AtomicU32 (info.rs:17-22)

    def __str__(self):  # Display, info.rs:141-172
        g = lambda v: "{:,}".format(v)
        return ("Correctly matched sequences: %s\nConstant region mismatches:  %s\nSample barcode mismatches:   %s\n"
                "Counted barcode mismatches:  %s\nDuplicates:                  %s\nLow quality barcodes:        %s" %
                (g(self.matched), g(self.constant_region), g(self.sample_barcode), g(self.barcode), g(self.duplicates),
                 g(self.low_quality)))


class Results:
    """info.rs:669-808: results_hashmap[sample][\"b1,b2,..\"] = count"""

    def __init__(self, samples_barcode_hash, random_barcode, sample_barcode):
        self.results_hashmap = {}
        if samples_barcode_hash:
            for s in samples_barcode_hash:
                self.results_hashmap[s] = {}
        elif not sample_barcode:
            self.results_hashmap["barcode"] = {}

    def fill(self, rows):
        for sample, tup, count in rows:
            self.results_hashmap.setdefault(sample, {})[tup] = count


class SequenceParser:
    """parse.rs:15-163.  The worker pool of main.rs:93-120 becomes one engine per GPU; the queue of
    packed records (SharedMutData.seq) becomes batches of sequence / quality lines."""

    def __init__(self, results, sequence_errors, sequence_format, max_errors, min_quality_score, device=0):
        self.results = results
        self.sequence_errors = sequence_errors
        self.plan = sequence_format.plan
        so, bo, co = max_errors.options
        self.plan.set_max_errors(so, bo, co)
        self.plan.set_min_quality(min_quality_score)
        self.engine = Engine(self.plan, device)

    def parse(self, seq, qual, stride, read_len, lens=None):
        """one batch of reads from host arrays; call finish() after the last batch"""
        self.engine.submit_host(seq, qual, stride, read_len, lens)

    def finish(self):
        self.sequence_errors.update(self.engine.counters())
        self.results.fill(self.engine.result_rows())
