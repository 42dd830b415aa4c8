/*
 * oracle.h -- CPU restatement of the per-read match/count path of
 * Roco-scientist/NGS-Barcode-Count (crate barcode-count v0.11.1).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 *
 * Pinning status: the reference is Rust and no Rust toolchain exists in this
 * image, so the reference binary cannot be executed here.  The oracle is
 * pinned by the only known answers the reference holds for this path --
 * the fix_error doctest (src/parse.rs:540-551) and the MaxSeqErrors doctests
 * (src/info.rs:479-611) -- see tests/test_oracle_kat.py.  For the anchor
 * search, constant-region repair, quality filter, SequenceMatchResult and
 * Results::add_count the reference has no tests: PARITY UNPINNED by the
 * reference for those; they are cross-checked against an independent Python
 * restatement (tests/pyref.py) and hand-derived vectors (tests/golden/).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference root, src/...).
 */
#ifndef BC_ORACLE_H
#define BC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

/* indices into the counter array: the six SequenceErrors fields, src/info.rs:16-23 */
enum {
  ORC_MATCHED = 0,
  ORC_CONSTANT_REGION = 1,
  ORC_SAMPLE_BARCODE = 2,
  ORC_BARCODE = 3,
  ORC_DUPLICATES = 4,
  ORC_LOW_QUALITY = 5,
  ORC_NCOUNTERS = 6
};

/* SequenceFormat::parse_format_file, src/info.rs:215-310.  `text` is the
 * content of the scheme file.  Returns NULL (and fills err) when the
 * reference would fail (duplicate group name -> regex compile error). */
orc_ctx *orc_new(const char *text, size_t len, char *err, size_t errlen);
void orc_free(orc_ctx *c);

/* SequenceFormat fields, src/info.rs:176-187 */
const char *orc_format_string(const orc_ctx *c);
const char *orc_regions_string(const orc_ctx *c);
const char *orc_regex_string(const orc_ctx *c);
uint32_t orc_length(const orc_ctx *c);
uint32_t orc_constant_region_length(const orc_ctx *c);
uint32_t orc_barcode_num(const orc_ctx *c);
uint32_t orc_barcode_length(const orc_ctx *c, uint32_t i);
int32_t orc_sample_length(const orc_ctx *c); /* -1 = None */
int orc_has_random(const orc_ctx *c);
int orc_has_sample(const orc_ctx *c);

/* BarcodeConversions::sample_barcode_file_conversion + get_sample_seqs,
 * src/info.rs:364-381, 435-441 (CSV text in). */
int orc_load_sample_csv(orc_ctx *c, const char *text, size_t len);
/* BarcodeConversions::barcode_file_conversion + get_barcode_seqs,
 * src/info.rs:390-433, 444-456.  Returns 0 ok, <0 error (message in err). */
int orc_load_counted_csv(orc_ctx *c, const char *text, size_t len, char *err, size_t errlen);
/* direct set construction (what the loaders produce) */
int orc_add_sample(orc_ctx *c, const char *seq, const char *id);
int orc_add_counted(orc_ctx *c, uint32_t barcode_index, const char *seq, const char *id);

/* MaxSeqErrors::new, src/info.rs:490-543.  Pass -1 for None. */
void orc_set_max_errors(orc_ctx *c, int sample_errors, int barcode_errors, int constant_errors);
void orc_set_min_quality(orc_ctx *c, float min_quality);
uint32_t orc_max_constant_errors(const orc_ctx *c);
uint32_t orc_max_sample_errors(const orc_ctx *c);
uint32_t orc_max_barcode_errors(const orc_ctx *c, uint32_t i);

/* Free-standing MaxSeqErrors::new for the doctest known answers.
 * out[0]=constant, out[1]=sample, out[2..2+n)=barcodes. */
void orc_max_seq_errors(int sample_errors, int sample_size, int barcode_errors, const uint16_t *barcode_sizes,
                        uint32_t n_barcodes, int constant_errors, uint16_t constant_region_size, uint16_t *out);

/* Must be called after sets / budgets are in place and before reads:
 * Results::new, src/info.rs:678-732. */
void orc_begin(orc_ctx *c);

/* One iteration of SequenceParser::parse, src/parse.rs:53-76, on one read
 * (sequence line and quality line, no terminators).  Returns the outcome
 * counter index that was incremented. */
int orc_process_read(orc_ctx *c, const char *seq, size_t seqlen, const char *qual, size_t quallen);
/* n reads at fixed stride; lens==NULL -> every read is read_len long. */
void orc_process_batch(orc_ctx *c, const uint8_t *seq, const uint8_t *qual, const uint16_t *lens, uint32_t stride,
                       uint32_t read_len, uint64_t n);

/* orc_process_batch that also reports, per read, the counter index it incremented; qlens (may be NULL) gives quality
 * lines a length of their own */
void orc_process_batch_outcomes(orc_ctx *c, const uint8_t *seq, const uint8_t *qual, const uint16_t *lens,
                                const uint16_t *qlens, uint32_t stride, uint32_t read_len, uint64_t n, uint8_t *outcomes);

/* The same reads through the reference's own thread structure (bench.py's cpu_baseline): the calling thread is the
 * reader posting packed 4-line records to a mutex-guarded deque (10,000-record back-pressure spin,
 * src/input.rs:115-148), n_workers threads pop and match (busy spin while empty, src/parse.rs:53-86) on their own
 * clone of the static inputs (src/main.rs:95-102) and add into ONE Results -- `shared`'s -- under one mutex
 * (src/parse.rs:60-64).  Outcome counters stay per worker context: sum them. */
int orc_run_reference_threads(orc_ctx **workers, uint32_t n_workers, orc_ctx *shared, const uint8_t *seq,
                              const uint8_t *qual, uint32_t stride, uint32_t read_len, uint64_t n);

void orc_counters(const orc_ctx *c, uint64_t out[ORC_NCOUNTERS]);
/* reads whose handling is undefined in the reference (len < format length:
 * usize underflow at src/parse.rs:291); counted as constant-region errors */
uint64_t orc_undefined_reads(const orc_ctx *c);

/* result rows: (sample key, "b1,b2,.." tuple, count); count = set size in
 * random-barcode mode (src/output.rs:265-270).  Order is unspecified. */
uint64_t orc_result_rows(orc_ctx *c);
int orc_result_row(orc_ctx *c, uint64_t i, const char **sample, const char **tuple, uint64_t *count);
/* number of sample keys present in the results map (incl. empty ones,
 * src/info.rs:698-709) and the i-th key */
uint64_t orc_result_samples(orc_ctx *c);
const char *orc_result_sample(orc_ctx *c, uint64_t i);

/* fix_error, src/parse.rs:553-593.  Returns the index of the chosen
 * candidate or -1 for None. */
int64_t orc_fix_error(const char *mismatch_seq, const char *const *possible, uint64_t n, uint16_t mismatches);

/* ID conversion tables kept by the loaders (last duplicate wins,
 * src/info.rs:378, 418) */
const char *orc_sample_id(const orc_ctx *c, const char *seq);
const char *orc_counted_id(const orc_ctx *c, uint32_t barcode_index, const char *seq);

#ifdef __cplusplus
}
#endif
#endif
