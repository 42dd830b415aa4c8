/*
 * barcode_count_hip.h -- C ABI of the MI355X (gfx950) barcode match/count engine.
 *
 * Drop-in boundary for ONE path of Roco-scientist/NGS-Barcode-Count (crate
 * barcode-count v0.11.1): the per-read match/count loop of src/parse.rs
 * (SequenceParser::parse, parse.rs:53-76, and everything it calls) together
 * with Results::add_count (info.rs:735-808) and the six SequenceErrors
 * counters (info.rs:16-139).  The reference exposes no FFI of its own; these
 * entry points are what a Rust `extern "C"` block in the reference's
 * src/main.rs would bind in place of the rayon worker fan-out
 * (main.rs:93-120) -- see INTEGRATION.md for that binding.
 *
 * Plain pointers and sizes only; no C++ or torch types cross this boundary.
 * All functions return BC_OK (0) or a negative status; bc_last_error() gives
 * the message of the calling thread's last failure.  Nothing here falls back
 * to a CPU implementation: an engine cannot be created without a HIP device.
 */
#ifndef BARCODE_COUNT_HIP_H
#define BARCODE_COUNT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BC_OK 0
#define BC_ERR_INVALID (-1)     /* the reference would return Err / panic on this input */
#define BC_ERR_UNSUPPORTED (-2) /* valid for the reference, not handled by this engine (message says what) */
#define BC_ERR_HIP (-3)         /* HIP runtime failure */
#define BC_ERR_NOMEM (-4)
#define BC_ERR_STATE (-5)       /* call order violated */
#define BC_ERR_COMM (-6)        /* the exchange between the ranks of a multi-GPU job failed (a peer is gone, I/O) */

/* outcome counters: SequenceErrors (info.rs:16-23) in Display order (info.rs:141-172),
 * then two engine-side extras */
enum {
  BC_MATCHED = 0,
  BC_CONSTANT_REGION = 1,
  BC_SAMPLE_BARCODE = 2,
  BC_BARCODE = 3,
  BC_DUPLICATES = 4,
  BC_LOW_QUALITY = 5,
  BC_TOTAL_READS = 6, /* reads submitted (input.rs:128-130 total_reads, minus the gz off-by-one) */
  BC_UNSUPPORTED_READS = 7, /* reads the engine cannot score exactly (non-ASCII bytes); always 0 for FASTQ */
  BC_NCOUNTERS = 8
};

typedef struct bc_plan bc_plan;     /* compiled scheme + known barcode sets + error budgets (host) */
typedef struct bc_engine bc_engine; /* one run on one GPU: device tables, stream, results */
typedef struct bc_synth bc_synth;   /* counter-based synthetic read generator */

const char *bc_version(void);
const char *bc_last_error(void);

/* ---- static run description ------------------------------------------------------------ */

/* SequenceFormat::parse_format_file (info.rs:215-310); `text` = content of the scheme file. */
bc_plan *bc_plan_create(const char *scheme_text, size_t len);
void bc_plan_destroy(bc_plan *p);
/* SequenceFormat fields (info.rs:176-187) */
const char *bc_plan_format_string(const bc_plan *p);
const char *bc_plan_regions_string(const bc_plan *p);
const char *bc_plan_regex_string(const bc_plan *p);
uint32_t bc_plan_length(const bc_plan *p);
uint32_t bc_plan_constant_region_length(const bc_plan *p);
uint32_t bc_plan_barcode_num(const bc_plan *p);
uint32_t bc_plan_barcode_length(const bc_plan *p, uint32_t i);
int32_t bc_plan_sample_length(const bc_plan *p); /* -1 = None */
int bc_plan_has_random(const bc_plan *p);
int bc_plan_has_sample(const bc_plan *p);

/* BarcodeConversions::sample_barcode_file_conversion + get_sample_seqs (info.rs:364-381, 435-441) */
int bc_plan_load_sample_csv(bc_plan *p, const char *csv_text, size_t len);
/* BarcodeConversions::barcode_file_conversion + get_barcode_seqs (info.rs:390-433, 444-456) */
int bc_plan_load_counted_csv(bc_plan *p, const char *csv_text, size_t len);
/* what the loaders insert, one entry at a time (a later duplicate sequence replaces the ID) */
int bc_plan_add_sample(bc_plan *p, const char *seq, const char *id);
int bc_plan_add_counted(bc_plan *p, uint32_t barcode_index, const char *seq, const char *id);
/* known sets in engine index order (index = position in first-insertion order) */
uint32_t bc_plan_n_samples(const bc_plan *p);
const char *bc_plan_sample_seq(const bc_plan *p, uint32_t i);
const char *bc_plan_sample_id(const bc_plan *p, uint32_t i);
uint32_t bc_plan_n_counted(const bc_plan *p, uint32_t barcode_index);
const char *bc_plan_counted_seq(const bc_plan *p, uint32_t barcode_index, uint32_t i);
const char *bc_plan_counted_id(const bc_plan *p, uint32_t barcode_index, uint32_t i);

/* MaxSeqErrors::new (info.rs:490-543); -1 = None (use 20 % of the length) */
int bc_plan_set_max_errors(bc_plan *p, int sample_errors, int barcode_errors, int constant_errors);
uint32_t bc_plan_max_constant_errors(const bc_plan *p);          /* info.rs:565 */
uint32_t bc_plan_max_sample_errors(const bc_plan *p);            /* info.rs:589 */
uint32_t bc_plan_max_barcode_errors(const bc_plan *p, uint32_t i); /* info.rs:613 */
/* free-standing form for the reference's doctest values: out[0]=constant, out[1]=sample, out[2..]=barcodes */
void bc_max_seq_errors(int sample_errors, int sample_size, int barcode_errors, const uint16_t *barcode_sizes,
                       uint32_t n_barcodes, int constant_errors, uint16_t constant_region_size, uint16_t *out);
/* --min-quality (arguments.rs:111-118); 0 = filter off (parse.rs:98) */
int bc_plan_set_min_quality(bc_plan *p, float min_quality);
/* integer score-sum threshold equivalent to the f32 mean test of parse.rs:352-355 for a run of n
 * bases: low quality <=> sum(scores) < T_n */
uint32_t bc_plan_quality_threshold(const bc_plan *p, uint32_t run_len);

/* ---- engine ---------------------------------------------------------------------------- */

/* Number of u32 entries of the dense per-(sample, barcode tuple) counter table the plan needs
 * (product of the set sizes); 0 when the plan cannot use a dense table. */
uint64_t bc_plan_table_entries(const bc_plan *p);

/* Replaces SequenceParser::new (parse.rs:28-52) for all workers of one device.  `hip_stream`
 * (a hipStream_t) may be NULL: the engine then owns a stream.  `table_mem` may point to
 * caller-owned device memory of bc_plan_table_entries()*4 bytes (zeroed by the caller), so the
 * caller can reduce it across devices with RCCL; NULL lets the engine allocate it. */
bc_engine *bc_engine_create(const bc_plan *p, int device_id, void *hip_stream, void *table_mem);
void bc_engine_destroy(bc_engine *e);

/* One batch of reads = the records the reference's workers pop from SharedMutData.seq
 * (parse.rs:78-86), already split into the sequence line and the quality line.
 * Layout: read i occupies bytes [i*stride, i*stride+len_i) of `seq` and of `qual`
 * (ASCII, as in the FASTQ); len_i = lens[i] or read_len when lens is NULL.  qual may be NULL
 * only when the quality filter is off.  Both buffers must be 16-byte aligned.
 * _device: pointers are device memory; the call only enqueues work on the engine's stream.
 * _host: pointers are host memory; the engine stages them through pinned buffers with
 * hipMemcpyAsync on a side stream and returns when the host buffers may be reused. */
int bc_engine_submit_device(bc_engine *e, const void *d_seq, const void *d_qual, const void *d_lens, uint32_t stride,
                            uint32_t read_len, uint64_t n_reads);
/* The same for records whose quality line is not as long as their sequence line (trimmed or damaged files): the
 * reference zips the scores with the regions (parse.rs:340-345), so what counts is the quality line's own length.
 * d_lens / d_qlens: u16 per read, both required; the first min(qlen, stride) quality bytes are stored at the stride. */
int bc_engine_submit_device_q(bc_engine *e, const void *d_seq, const void *d_qual, const void *d_lens, const void *d_qlens,
                              uint32_t stride, uint64_t n_reads);
/* the engine's stream (a hipStream_t) and device, for callers that produce batches on the device themselves */
void *bc_engine_hip_stream(bc_engine *e);
int bc_engine_device(const bc_engine *e);
int bc_engine_submit_host(bc_engine *e, const void *seq, const void *qual, const uint16_t *lens, uint32_t stride,
                          uint32_t read_len, uint64_t n_reads);
int bc_engine_sync(bc_engine *e);
/* zero the table and the counters (a fresh Results::new, info.rs:678) */
int bc_engine_reset(bc_engine *e);
/* the same without the outcome counters: a fresh Results (table, bit map, key set / map) for the next sample of a run
 * whose SequenceErrors go on counting.  Large engine-owned tables are reset by the blocks that were touched (they are
 * almost all zeros after a job: first occurrences live in the bit map), not by a 16 GB memset. */
int bc_engine_reset_results(bc_engine *e);

/* SequenceErrors values (u64; the reference wraps at 2^32, info.rs:17-22) */
int bc_engine_counters(bc_engine *e, uint64_t out[BC_NCOUNTERS]);
/* device pointers, for cross-device reduction by the caller (RCCL sum over u32 / u64).  The table holds plain u32
 * counts when this returns and again after every later bc_engine_sync (large tables count on two levels in between,
 * bc_kernel.h: taking the pointer makes every sync fold them, as for a caller-owned table). */
void *bc_engine_table_ptr(bc_engine *e);
void *bc_engine_counters_ptr(bc_engine *e);
uint64_t bc_engine_table_entries(const bc_engine *e);

/* Compacts the non-zero table entries on the device; rows are then readable in any order.
 * A row = (sample index, barcode index per counted barcode, count): the (sample, tuple, count)
 * triples Results holds (info.rs:661-665), with indices into the plan's known sets.
 * Device memory is bounded whatever the table holds: the table is compacted range by range through two staging
 * buffers of BC_FINISH_CHUNK_ROWS rows (default 2^22; 12 bytes per row on the device and pinned).  bc_engine_finish
 * keeps all rows on the host (12 bytes per row; BC_ERR_NOMEM when that does not fit). */
int bc_engine_finish(bc_engine *e, uint64_t *n_rows);
int bc_engine_rows(bc_engine *e, uint64_t first, uint64_t n, uint32_t *sample_idx, uint32_t *barcode_idx,
                   uint64_t *count);
/* The same compaction with the rows handed to `fn` chunk by chunk as they leave the device (nothing is kept: host
 * memory is bounded too).  A chunk is n (key, count) pairs valid during the call; key = the dense table index
 * (bc_engine_decode_index turns it into set indices) or, for plans that keep raw captures, the tuple key.  fn returns 0
 * to go on; anything else stops the compaction (BC_ERR_STATE). */
typedef int (*bc_rows_fn)(const uint64_t *key, const uint32_t *count, uint64_t n, void *user);
int bc_engine_finish_stream(bc_engine *e, bc_rows_fn fn, void *user, uint64_t *n_rows);
int bc_engine_decode_index(const bc_engine *e, uint64_t dense_index, uint32_t *sample_idx, uint32_t *barcode_idx);
/* number of rows bc_engine_finish would produce now (dense plans without a random barcode): one sweep of the table,
 * nothing is moved */
int bc_engine_nonzero_entries(bc_engine *e, uint64_t *n);

/* Row i as the reference's Results holds it (info.rs:661-665): the sample key (a sample barcode
 * sequence, or "barcode" without a sample group) and the counted barcodes "b1,b2,.." as sequences.
 * Works for every plan, including those that keep raw captures (no sample / counted-barcode
 * file: README.md "Barcode-seq"), whose rows have no index form. */
int bc_engine_row_text(bc_engine *e, uint64_t row, char *sample, size_t sample_cap, char *tuple, size_t tuple_cap,
                       uint64_t *count);
/* 0: the engine cannot run the plan (bc_last_error says why); 1: dense counter table;
 * 2: hash map of tuple keys (some barcode is kept raw because no conversion file names it) */
int bc_plan_mode(const bc_plan *p);

/* Random-barcode schemes (PCR-duplicate collapse, Results::add_count info.rs:770-802): the engine
 * keeps the set of distinct (sample, barcode tuple, random barcode) keys in a device hash set; a
 * read whose key is already present counts as BC_DUPLICATES (parse.rs:65-69) and the count of a
 * tuple is the number of its distinct random barcodes (output.rs:265-270; bc_engine_finish turns
 * the set into the dense table).  A key is tuple_index * 5^len + base-5 code of the random barcode
 * (A,C,T,G,N = 0..4).  Across GPUs a sum of tables would be wrong: export the
 * keys, exchange them so that every key has one owner, import, then reduce.  Device pointers. */
/* Plans whose captures kept raw (no conversion file) or whose random barcode do not fit that 64-bit key -- more than 27
 * bases, or several captures that overflow it together -- count under WIDE keys of bc_engine_key_words() u64 each (word
 * 0 a fingerprint of the rest, then the captures as bit planes; csrc/bc_long.h): everywhere below a key is then that
 * many consecutive u64, and a buffer of n keys holds n * bc_engine_key_words() of them.  Such plans run on the
 * wave-per-read kernel and hand their rows out as text (bc_engine_row_text). */
uint32_t bc_engine_key_words(const bc_engine *e);
int bc_engine_key_count(bc_engine *e, uint64_t *n);
int bc_engine_export_keys(bc_engine *e, void *d_keys, uint64_t capacity, uint64_t *n);
int bc_engine_import_keys(bc_engine *e, const void *d_keys, uint64_t n, uint64_t *n_new);
int bc_engine_clear_keys(bc_engine *e);
/* Dense table + random barcode, several GPUs: after the key exchange every rank turns ITS (owned) keys into per-tuple
 * distinct counts in its table with bc_engine_materialize_table; the tables are then summed onto the root
 * (distributed.reduce_table), whose bc_engine_finish compacts the summed table as it stands.  (On one GPU
 * bc_engine_finish materializes by itself.)  Any later submit / import / clear invalidates the materialized table. */
int bc_engine_materialize_table(bc_engine *e);
/* Plans that keep raw captures and have NO random barcode count in a (key -> count) map.  Across GPUs
 * the maps are merged by sending every (key, count) pair to the key's owner (or all of them to the
 * root), where bc_engine_import_counts adds them in.  export with NULL buffers only counts the pairs.
 * Device pointers: keys u64, counts u32. */
int bc_engine_export_counts(bc_engine *e, void *d_keys, void *d_counts, uint64_t capacity, uint64_t *n);
int bc_engine_import_counts(bc_engine *e, const void *d_keys, const void *d_counts, uint64_t n);

/* Dense counts for the wire.  out[i] = table[i] where that fits a byte, else 0 with (i, table[i]) appended to the
 * overflow list; *n_ovf = entries the list needs (call again with a larger one if it exceeds ovf_capacity).  The
 * receiver adds the bytes up and then the list entries -- exact for any counts, a quarter of the bytes when counts
 * are small (ngs-barcode-count_amd/distributed.py reduce_table).  Device pointers; runs on hip_stream and waits. */
int bc_table_pack_u8(const void *d_table_u32, uint64_t n, void *d_out_u8, void *d_ovf_idx_u64, void *d_ovf_val_u32,
                     uint64_t ovf_capacity, uint64_t *n_ovf, int device_id, void *hip_stream);

/* ... and the receiving side of the exchange: out[i] = sum of the n_rows byte slices rows[r * n + i], i < n
 * (n a multiple of 4). */
int bc_table_sum_u8(const void *d_rows_u8, uint32_t n_rows, uint64_t n, void *d_out_u32, int device_id, void *hip_stream);

/* ---- several GPUs: one process (or thread) per GPU, one exchange at the end of the job ----------
 * The reference is one process whose workers share one Results map (main.rs:93-120).  Across GPUs every rank counts
 * its own shard of the reads into its own engine -- no communication on the data path -- and the job ends with ONE
 * collective call on every rank, after which the root's engine holds the job's result:
 *   - dense table, no random barcode: the u32 tables are summed onto the root (all-to-all of byte-packed slices, local
 *     sums, slices to the root: one xGMI link per peer, never a ring);
 *   - random barcode: the (tuple, random) keys first go to one owner rank each, where duplicates across ranks collapse
 *     (a sum of set sizes would be wrong, output.rs:265-270); the owners' per-tuple distinct counts then add up;
 *   - captures kept raw: every rank's (key, count) pairs / keys go to the root's map.
 * A communicator is either RCCL over xGMI (bc_comm_create: ncclSend / ncclRecv on the engine's stream; rank 0 makes the
 * id with bc_comm_unique_id and hands its BC_COMM_ID_BYTES bytes to the other ranks by whatever means the caller has --
 * a file, a pipe, MPI) or message files in a directory all ranks can write to (bc_comm_create_host: works between any
 * processes of one machine, several ranks on ONE GPU included; device buffers are staged through host memory).
 * counters (may be NULL): the job's outcome counters on the root, zeros elsewhere.  The tables of the other ranks are
 * left in an unspecified state.  Tables must be 16-byte aligned (engine-owned ones are). */
typedef struct bc_comm bc_comm;
#define BC_COMM_ID_BYTES 128
int bc_comm_unique_id(void *id_out);
bc_comm *bc_comm_create(const void *id, int rank, int world, int device_id);
bc_comm *bc_comm_create_host(const char *dir, int rank, int world);
void bc_comm_destroy(bc_comm *c);
int bc_comm_rank(const bc_comm *c);
int bc_comm_world(const bc_comm *c);
int bc_comm_barrier(bc_comm *c);
/* element-wise sum of n u64 onto the root (e.g. the shards' "Total sequences"); the other ranks' values stay */
int bc_comm_sum_u64(bc_comm *c, uint64_t *vals, int n, int root);
/* the exchange alone: afterwards the root's table / key set / key map is the job's */
int bc_engine_reduce_all(bc_engine *e, bc_comm *c, int root, uint64_t counters[BC_NCOUNTERS]);
/* ... followed by bc_engine_finish on the root (rows readable there; *n_rows = 0 on the other ranks).  c = NULL or a
 * communicator of one rank: the same as bc_engine_counters + bc_engine_finish. */
int bc_engine_finish_all(bc_engine *e, bc_comm *c, int root, uint64_t counters[BC_NCOUNTERS], uint64_t *n_rows);
/* the plan an engine was created from */
const bc_plan *bc_engine_plan(const bc_engine *e);

/* Debug / parity-test hook: the next submits also write, for read i of the submit, its outcome
 * (BC_* counter index; BC_MATCHED = passed every test) to d_outcome_u8[i] and its dense table index
 * to d_index_u64[i].  Both device pointers; NULL switches tracing off. */
int bc_engine_trace(bc_engine *e, void *d_outcome_u8, void *d_index_u64);

/* HIP-event timing of the match/count kernel on the engine's stream (for the roofline) */
int bc_engine_timing(bc_engine *e, int enable);
int bc_engine_kernel_ms(bc_engine *e, double *total_ms, uint64_t *launches);   /* sum since the last call; resets */
/* the same launches one by one, in launch order (up to `capacity` of them; *launches = how many there are); does not
 * reset: call it before bc_engine_kernel_ms */
int bc_engine_kernel_ms_each(bc_engine *e, double *ms_out, uint64_t capacity, uint64_t *launches);
/* Which kernel the last submit launched: "match_count_kernel<NW,NWW>" (the generic one, any plan) or
 * "bc_jit_match_count<NW,NWW>" (the one specialised to this plan's scheme); "" before the first submit. */
const char *bc_engine_kernel_name(bc_engine *e);
/* Shader clock right now (MHz), measured by a 0.3 ms probe kernel on the engine's stream against the
 * 100 MHz reference counter.  Call it straight after the work of interest: the clock sags under power
 * and thermal limits, and the match kernel's time follows it. */
int bc_engine_sclk_mhz(bc_engine *e, double *mhz);

/* Box diagnostic for benchmarks: issues ~n_atomics no-return atomic adds of +1 at random entries of a u32 table, then
 * the same adds of -1 at the same entries (the table ends as it was, but is transiently different: the caller must be
 * idle on it), and reports the sustained rate of the first pass -- what the memory system of THIS device gives the match
 * kernel's counting alone (it differs between boxes by tens of percent).  Runs on the NULL stream and waits for the
 * device; it is not ordered against work on a non-blocking stream, so sync the engine first. */
int bc_probe_atomic_rate(int device_id, void *d_table_u32, uint64_t entries, uint64_t n_atomics, double *atomics_per_s);

/* Scheme-specialised kernels.  Next to the generic kernel (any plan, plan read from memory) an engine
 * can run one compiled for its plan's scheme (offsets, shift programs, thresholds and set sizes as
 * immediates; ~15-25 % faster).  The code object is looked up by content hash in jit_cache/ next to
 * the library and in $BC_JIT_CACHE: a hit is used from the first submit on.  On a miss the engine
 * keeps counting with the generic kernel and, once 2^20 reads have been submitted, compiles the
 * specialised one on a worker thread (ROCm's hipcc as a child process, else hiprtc in-process), stores
 * it in the cache and switches over at the next submit after it is ready.  Results are identical
 * either way.  If no compiler is available the engine says so once on stderr.
 * Environment: BC_JIT=0 (never) | 1 (default) | force (compile synchronously at the first submit) |
 * cached (cache hits only).
 * A specialised kernel is also specific to the batch shape: the stride and, for fixed-length batches, the read
 * length are compile-time constants of it (one kernel per shape; a FASTQ file has one or a few).
 * bc_plan_precompile builds the kernel of one shape ahead of time without touching a GPU: stride / read_len as
 * they will be passed to bc_engine_submit_* (reads of up to 256 bases; longer ones stay on the generic kernel),
 * with_lens = per-read lengths will be passed (read_len is then ignored), cache_dir NULL = next to the library. */
int bc_plan_precompile(const bc_plan *p, uint32_t stride, uint32_t read_len, int with_lens, const char *cache_dir);

/* fix_error (parse.rs:553-593) on the device: nearest unique candidate under Hamming distance
 * with 'N' wildcards, common-prefix compare.  Returns the candidate index, -1 for None,
 * or < -1 on error (BC_ERR_* - 1). */
int64_t bc_fix_error(const char *mismatch_seq, const char *const *possible_seqs, uint64_t n, uint16_t mismatches,
                     int device_id);

/* ---- FASTQ ingest (replaces input::read_fastq, input.rs:24-149) --------------- */

/* Reads a whole *.fastq / *.fastq.gz file, frames it into 4-line records exactly as
 * FastqLineReader does (input.rs:115-148), and submits the sequence / quality lines to the engine in
 * batches through bc_engine_submit_host (pinned buffers, copies overlapped with the kernel).
 * *total_reads receives the reference's "Total sequences" value, including its quirks: a trailing
 * partial record is counted but never processed, and a .gz input counts one read more
 * (input.rs:69-73, 128-130).  progress (may be NULL) is called with the running total about every
 * million reads.  Errors the reference raises come back as BC_ERR_INVALID with its message:
 * wrong extension (input.rs:36-39), unreadable file, first record not FASTQ (parse.rs:377-394). */
typedef void (*bc_progress_fn)(uint64_t total_reads, void *user);
int bc_fastq_count(bc_engine *e, const char *fastq_path, uint64_t *total_reads, bc_progress_fn progress, void *user);
/* One shard of n_shards (one per GPU of a job): the records that start inside this shard's share of the file's bytes.
 * Shard boundaries are moved to the next record start ('@' line whose second-next line begins with '+'), so the shards
 * tile the file's records exactly and their totals add up to bc_fastq_count's -- for a file whose lines come in fours;
 * one that does not is refused (BC_ERR_INVALID: the reference frames from the file's first line, a shard cannot).  The
 * first-record check is the first shard's, the trailing partial record the last shard's.  A .gz stream cannot be entered
 * in the middle: shard 0 reads all of it, the others nothing. */
int bc_fastq_count_shard(bc_engine *e, const char *fastq_path, uint32_t shard, uint32_t n_shards, uint64_t *total_reads,
                         bc_progress_fn progress, void *user);
/* where a shard that nominally begins at byte `offset` of a plain FASTQ file really begins: the first record start at
 * or after it (the file's size when there is none).  Host logic only: no engine, no GPU. */
int bc_fastq_record_start(const char *fastq_path, uint64_t offset, uint64_t *start);

/* ---- synthetic workloads (bench + full-size parity) -------------------------------------- */

typedef struct bc_synth_params {
  uint64_t seed;
  uint32_t read_len;     /* R; construct start uniform in [0, R-L-1] */
  uint32_t p_sub;        /* per-base substitution probability * 2^32 */
  uint32_t p_n;          /* per-base 'N' probability * 2^32 */
  uint32_t p_lowq;       /* probability * 2^32 that one barcode of a read gets low qualities */
  uint8_t phred_lo, phred_hi;   /* default Phred range, inclusive */
  uint8_t lowq_lo, lowq_hi;     /* low-quality Phred range, inclusive */
  uint64_t n_molecules;  /* 0: every read is its own molecule; else reads are draws from this many
                            molecules (PCR duplicates) */
  uint32_t zipf;         /* 1: counted-barcode indices follow a Zipf-like law with exponent 1 (P(rank k) ~ 1/k; the
                            hot-spot variant of SURVEY.md 8(d)) instead of the uniform one */
  uint32_t reserved;
  uint64_t geo_total;    /* > 0: PCR copies per molecule geometric with mean 2, scattered over this many reads (the
                            job's total) by a fixed permutation; n_molecules is then ignored */
} bc_synth_params;

bc_synth *bc_synth_create(const bc_plan *p, const bc_synth_params *params);
void bc_synth_destroy(bc_synth *s);
int bc_synth_generate_host(bc_synth *s, uint64_t first_read, uint64_t n_reads, void *seq, void *qual, uint32_t stride);
int bc_synth_generate_device(bc_synth *s, int device_id, void *hip_stream, uint64_t first_read, uint64_t n_reads,
                             void *d_seq, void *d_qual, uint32_t stride);
/* deterministic reference set: n distinct k-mers, pairwise Hamming distance >= min_dist;
 * out receives n*(k+1) bytes (NUL-terminated strings) */
int bc_synth_make_set(uint64_t seed, uint32_t n, uint32_t k, uint32_t min_dist, char *out);

#ifdef __cplusplus
}
#endif
#endif
