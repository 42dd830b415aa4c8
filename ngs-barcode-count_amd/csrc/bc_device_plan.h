// bc_device_plan.h -- the compiled scheme as the GPU sees it (plain data, uploaded once per engine).
//
// Device-side form of what every reference worker clones from main.rs:95-102:
// SequenceFormat (info.rs:176-187), MaxSeqErrors (info.rs:461-472) and the known-barcode
// sets of BarcodeConversions (info.rs:338-343).
#pragma once
#include "bc_intrin.h"

#ifdef BC_JIT_TU
// A scheme-specialised kernel embeds the plan as a constant but takes the table addresses from this
// array, which the engine fills after loading the code object: the compiled code then depends only
// on the scheme, the budgets and the set sizes, and can be cached across runs.
extern "C" __constant__ uint64_t bc_jit_addr[];
#define BC_ADDR(field, k) bc_jit_addr[index * kAddrPerGroup + (k)]
#define BC_PLAN_ADDR(field, k) bc_jit_addr[kMaxGroups * kAddrPerGroup + (k)]
// ... of a table only rare paths touch: read where it is used (a volatile load is not hoisted out of the loop over the
// tiles, where it would sit in a register the hot path is short of for the whole launch)
#define BC_ADDR_COLD(field, k) (*reinterpret_cast<const volatile uint64_t*>(&bc_jit_addr[index * kAddrPerGroup + (k)]))
#else
#define BC_ADDR_COLD(field, k) field
#define BC_ADDR(field, k) field
#define BC_PLAN_ADDR(field, k) field
#endif

namespace bc {

constexpr int kMaxGroups = 18;
constexpr int kAddrPerGroup = 17;  // address fields of a DevGroup, in declaration order
constexpr int kPlanAddrs = 1;      // ... of the DevPlan itself    // sample + 16 counted barcodes + random
constexpr int kMaxEntries = 160;  // program entries per position class (up to three positions each)
constexpr int kMaxRuns = 40;      // quality runs of regions_string
constexpr int kClasses = 5;       // A, C, T, G constants + scheme-N ([AGCT]) positions
constexpr int kMaxNW = 10;        // 32-base words per read: reads up to 320 bases
constexpr uint32_t kFail = 0xFFFFFFFFu;
constexpr uint32_t kDeferred = 0xFFFFFFFEu;  // Ops::nearest: the verdict comes later (bc_kernel.h: search queue)
constexpr uint16_t kFail16 = 0xFFFFu;
constexpr uint32_t kLhashMul1 = 0x9E3779u;  // 24-bit multipliers: keys are below 2^20 (len <= 10), so the
constexpr uint32_t kLhashMul2 = 0x85EBCBu;  // product fits the full-rate 24-bit multiply
constexpr uint32_t kLhashMaxVec = 2048;  // 32 KiB of LDS per workgroup at most
// bucket of a key: 16 well-mixed bits of key * mul, scaled to [0, nb) -- nb need not be a power of two,
// which lets a table be sized to its load instead of the next power of two (LDS is what limits
// how many wavefronts stay resident).  Three full-rate 24-bit multiplies / shifts.
BC_HD uint32_t lhash_bucket(uint32_t key, uint32_t mul, uint32_t nb) {
  return mul24((mul24(key, mul) >> 8) & 0xFFFFu, nb) >> 16;
}

// letter code of an ASCII base: (c >> 1) & 3  ->  A=0 C=1 T=2 G=3
enum { kCodeA = 0, kCodeC = 1, kCodeT = 2, kCodeG = 3, kClassFmtN = 4 };

enum GroupType : uint32_t { kGroupSample = 0, kGroupBarcode = 1, kGroupRandom = 2 };
enum SetMode : uint32_t {
  kSetNone = 0,    // no known set: the capture itself is the key (raw-key mode)
  kSetDirect = 1,  // 4^len-entry correction table (len <= 10): one gather per barcode
  kSetHash = 2,    // exact hash lookup, then wave-cooperative Hamming search
  kSetScan = 3     // wave-cooperative Hamming search only
};

struct DevGroup {
  uint32_t type;       // GroupType
  uint32_t off;        // offset of the capture inside a match
  uint32_t len;        // capture length (<= 32)
  uint32_t mode;       // SetMode
  uint32_t n_refs;
  uint32_t max_err;    // MaxSeqErrors budget of this group
  uint32_t hmask;      // hash slots - 1
  uint32_t has_odd;    // set holds refs with 'N' or of a different length (no single-N shortcut)
  uint64_t table_stride;  // multiplier of this group's value (reference index, or base-5 code of a raw
                          // capture) in the mixed-radix tuple key = dense counter index
  // Device addresses are kept as integers so that a plan is plain bytes (a scheme-specialised
  // kernel embeds it as a compile-time constant); the accessors below turn them into pointers.
  // kSetDirect: one u32 per N-free capture q1 | q2 << len:
  //   bits 0-15 fix_error's verdict (reference index, kFail16 = None)
  //   bits 16-23 distance of the nearest reference (capped at 255), bit 24 set when only one reference is that near
  uint64_t dtable_a;
  uint64_t r1_a;     // reference bit planes (bit i = base i): ASCII bit 1
  uint64_t r2_a;     //                                          ASCII bit 2
  uint64_t rn_a;     // 'N' positions of the reference
  uint64_t rlen_a;   // reference length (u8)
  uint64_t hkeys_a;  // kSetHash: q1 | q2 << 32 of references usable for exact lookup
  uint64_t hvals_a;  // index or kFail for an empty slot
  // kSetHash, large sets: pigeonhole seeds.  The capture is cut into seed_nb = max_err + 1 blocks of
  // seed_blen bases; a reference within max_err mismatches equals the capture on at least one block,
  // so only the references filed under the capture's block values need scoring.
  uint32_t seed_nb;       // 0: no seed index, score every reference
  uint32_t seed_blen;
  uint32_t n_idx;         // plain references (no 'N', length == len) in the index
  uint32_t n_odd;         // the others, always scored
  uint64_t seed_off_a;    // [seed_nb][4^seed_blen + 1] bucket starts into seed_list rows
  uint64_t seed_list_a;   // [seed_nb][n_idx] entries {r1, r2, index, 0} (16 B) ordered by block value
  uint64_t odd_list_a;    // [n_odd]
  // kSetHash, first tier of the search (budget >= 1): two disjoint blocks of tier_blen bases at offsets 0
  // and tier_stride.  A reference within ONE mismatch of a capture equals it on one of them, and with
  // 4^8 buckets per block a bucket holds a handful of references at most: every LANE looks its own
  // capture up (bc_lane.h tier_lookup).  Nothing within one mismatch -> the pigeonhole search above.
  // ... and a coarser pigeonhole index in front of the full one: seed2_nb = 3 longer blocks catch every
  // reference within two mismatches with buckets an order of magnitude shorter
  uint32_t seed2_nb;      // 0: none
  uint32_t seed2_blen;
  uint64_t seed2_off_a;
  uint64_t seed2_list_a;
  uint32_t tier_blen;     // 0: no tier index
  uint32_t tier_stride;
  uint64_t tier_off_a;    // [2][4^tier_blen + 1] bucket starts
  uint64_t tier_list_a;   // [2][n_idx] entries {r1, r2, index, 0}
  uint64_t tier_bkt_a;    // [2][4^tier_blen][4] the first four entries of every bucket, one 64-byte line each:
                          // {r1, r2, index, references in the bucket}; the rest of a longer bucket is in tier_list
  // kSetDirect, small sets: the plain references also sit in an LDS-resident exact-match table of
  // 4-entry buckets, entry = capture key << ibits | reference index with ibits = 32 - 2 * len.  A key
  // lives in one of its two buckets (two multiplicative hashes), so a lookup is two 16-byte LDS reads.
  // A capture that is a reference is answered from LDS; a capture that is not falls through to the
  // dtable gather (or, when no mismatch is allowed, fails right away).  Unused slots hold copies of
  // a real entry.  lhash_complete: every plain reference found a slot, so "not in the table" proves
  // "not a reference" (it always does unless the builder ran out of moves).
  uint32_t lhash_nb;        // buckets of this group's table (lhash_bucket below); 0: no LDS table for this group
  uint32_t lhash_off;       // first bucket of this group in the workgroup's table area (16-byte units)
  uint32_t lhash_complete;
  uint32_t index;           // position in DevPlan::groups
  // tier_bkt in its compact form: four 8-byte entries per bucket (32 bytes: the whole first tier of 100 k 20-nt
  // guides is 4 MiB and mostly stays in the XCD's L2) -- entry = r1 | r2 << len | index << 2 len, the bucket's
  // reference count (capped at 7) in the top three bits of its first entry.  0: sixteen-byte entries.
  uint32_t tier_compact;
  uint32_t pad_;
  // ... and a middle one between the coarse index and the full one (budgets of four mismatches and more): seed3_nb = 4
  // blocks catch every reference within three mismatches.  One capture in a thousand needs it, but the full index's
  // buckets (one reference in 256 each, a dozen dependent round trips for 100 k references) held its whole wavefront.
  uint32_t seed3_nb;      // 0: none
  uint32_t seed3_blen;
  uint64_t seed3_off_a;
  uint64_t seed3_list_a;

  BC_HD const BC_GLOBAL uint32_t* dtable() const { return reinterpret_cast<const BC_GLOBAL uint32_t*>(BC_ADDR(dtable_a, 0)); }
  BC_HD const BC_GLOBAL uint32_t* r1() const { return reinterpret_cast<const BC_GLOBAL uint32_t*>(BC_ADDR(r1_a, 1)); }
  BC_HD const BC_GLOBAL uint32_t* r2() const { return reinterpret_cast<const BC_GLOBAL uint32_t*>(BC_ADDR(r2_a, 2)); }
  BC_HD const BC_GLOBAL uint32_t* rn() const { return reinterpret_cast<const BC_GLOBAL uint32_t*>(BC_ADDR(rn_a, 3)); }
  BC_HD const BC_GLOBAL uint8_t* rlen() const { return reinterpret_cast<const BC_GLOBAL uint8_t*>(BC_ADDR(rlen_a, 4)); }
  BC_HD const BC_GLOBAL uint64_t* hkeys() const { return reinterpret_cast<const BC_GLOBAL uint64_t*>(BC_ADDR(hkeys_a, 5)); }
  BC_HD const BC_GLOBAL uint32_t* hvals() const { return reinterpret_cast<const BC_GLOBAL uint32_t*>(BC_ADDR(hvals_a, 6)); }
  BC_HD const BC_GLOBAL uint32_t* seed_off() const { return reinterpret_cast<const BC_GLOBAL uint32_t*>(BC_ADDR(seed_off_a, 7)); }
  BC_HD const BC_GLOBAL uint32_t* seed_list() const { return reinterpret_cast<const BC_GLOBAL uint32_t*>(BC_ADDR(seed_list_a, 8)); }
  BC_HD const BC_GLOBAL uint32_t* odd_list() const { return reinterpret_cast<const BC_GLOBAL uint32_t*>(BC_ADDR(odd_list_a, 9)); }
  BC_HD const BC_GLOBAL uint32_t* seed2_off() const { return reinterpret_cast<const BC_GLOBAL uint32_t*>(BC_ADDR(seed2_off_a, 12)); }
  BC_HD const BC_GLOBAL uint32_t* seed2_list() const { return reinterpret_cast<const BC_GLOBAL uint32_t*>(BC_ADDR(seed2_list_a, 13)); }
  BC_HD const BC_GLOBAL uint32_t* seed3_off() const { return reinterpret_cast<const BC_GLOBAL uint32_t*>(BC_ADDR_COLD(seed3_off_a, 15)); }
  BC_HD const BC_GLOBAL uint32_t* seed3_list() const { return reinterpret_cast<const BC_GLOBAL uint32_t*>(BC_ADDR_COLD(seed3_list_a, 16)); }
  BC_HD const BC_GLOBAL uint32_t* tier_bkt() const { return reinterpret_cast<const BC_GLOBAL uint32_t*>(BC_ADDR(tier_bkt_a, 14)); }
  BC_HD const BC_GLOBAL uint32_t* tier_off() const { return reinterpret_cast<const BC_GLOBAL uint32_t*>(BC_ADDR(tier_off_a, 10)); }
  BC_HD const BC_GLOBAL uint32_t* tier_list() const { return reinterpret_cast<const BC_GLOBAL uint32_t*>(BC_ADDR(tier_list_a, 11)); }
};

struct DevPlan {
  uint32_t L;           // format length = bytes one match spans
  uint32_t RL;          // strlen(regions_string) (= L minus scheme-N positions, Appendix A Q9)
  uint32_t n_groups;
  uint32_t max_const;   // MaxSeqErrors::max_constant_errors
  uint32_t nb;          // bits of the bit-sliced mismatch counters: values 0..2^nb-1, then overflow
  uint32_t quality_on;  // min_quality > 0 (parse.rs:98)
  uint32_t n_runs;
  uint32_t discard_counts;  // sample file given but no sample group: add_count hits a temporary (info.rs:762-766)
  uint32_t has_fmtn;
  uint32_t no_repair;   // the scheme has lower-case constants: fix_constant_region can only fail (bc_plan.cpp)
  // random barcode (PCR-duplicate collapse, info.rs:770-802): the capture is kept raw; its base-5
  // code (A,C,T,G,N -> 0,1,2,3,4) and the dense tuple index form one 64-bit key of a device hash set
  uint32_t has_random, rnd_off, rnd_len;
  uint32_t sparse;      // some group has no known set: its capture's base-5 code is part of the key and
                        // the (then astronomically large) key space is held in a hash map, not a table
  uint64_t rspace;      // 5^rnd_len: key = dense_idx * rspace + code
  uint64_t dirty_off;   // two-level counting (large dense tables) with an engine-owned table: u32 words from the start
                        // of the bit map to a byte map with one flag per 64 table entries, set by whoever adds to the
                        // table -- after a short job the table is almost all zeros (first occurrences live in the
                        // bit map), and bc_engine_reset zeroes only the flagged 256-byte blocks.  0: no such map
  uint32_t lhash_vec;   // uint4s of the LDS exact-match area (0: none); image of it at lhash_a
  uint64_t lhash_a;
  uint32_t ablate;      // perf-debug only, honoured by -DBC_EXPERIMENT builds alone (`make experiment-lib`): bit mask of
                        // phases to skip; results are then wrong.  The release library ignores the field (abl() == 0).
  // Per position class a program that walks the class's format positions in ascending order,
  // shifting the class vector right by the distance to the next position (0..31 per shift).
  //   prog_mode 0: prog[0 .. n3) are full triples s1 | s2 << 8 | s3 << 16 (three positions each, a
  //                branch-free loop), prog[n3] = k << 24 | s1 | s2 << 8 holds the k (0..2) positions left;
  //   prog_mode 1: (some gap exceeds 31) prog[0 .. n3) are single steps k << 24 | s1, k = 0 meaning
  //                "shift only".
  // One spare dword follows every program so that a prefetch stays inside the array.
  uint32_t n3[kClasses];
  uint32_t prog_mode[kClasses];
  uint32_t n_pos[kClasses];  // positions in the class (0: nothing to do)
  uint32_t prog[kClasses][kMaxEntries + 2];
  uint32_t cmask[kMaxNW];  // bit p set: format position p is a constant base
  uint32_t run_off[kMaxRuns];   // quality runs in regions_string coordinates
  uint32_t run_len[kMaxRuns];
  uint32_t run_thr[kMaxRuns];   // low <=> sum(scores) < thr   (f32-exact, Appendix A Q10)
  DevGroup groups[kMaxGroups];  // sample group first (if any), then counted barcodes in order, then random

  BC_HD uint64_t lhash_image() const { return BC_PLAN_ADDR(lhash_a, 0); }
  // One known barcode group and nothing else (CRISPR guides): a read's verdict IS its barcode's, so a capture that
  // needs the search beyond one mismatch may wait in the wave's queue until four of them share one coarse-index probe
  // (bc_kernel.h).
  BC_HD bool defer_search() const {
    return n_groups == 1u && !has_random && !sparse && groups[0].mode == kSetHash && groups[0].seed_nb != 0u &&
           groups[0].seed2_nb != 0u && groups[0].seed2_nb <= 3u && groups[0].n_odd == 0u;
  }
#ifdef BC_EXPERIMENT
  BC_HD uint32_t abl() const { return ablate; }
#else
  BC_HD constexpr uint32_t abl() const { return 0u; }
#endif
};

constexpr int kNCounters = 8;      // BC_NCOUNTERS
constexpr int kTotalReads = 6;     // BC_TOTAL_READS

// outcome of one read, in counter order (barcode_count_hip.h BC_*)
enum Outcome : uint32_t {
  kMatched = 0, kConstantRegion = 1, kSampleBarcode = 2, kBarcode = 3, kDuplicate = 4, kLowQuality = 5,
  kPending = 6,  // (not an outcome: the read's one barcode waits in the wave's search queue, bc_kernel.h; the slot is
                 // BC_TOTAL_READS' in the counter array, which is never counted by outcome)
  kUnsupported = 7
};

}  // namespace bc
