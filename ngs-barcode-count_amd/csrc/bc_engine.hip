// bc_engine.hip -- gfx950 kernels and the engine half of the C ABI (include/barcode_count_hip.h).
//
// Replaces, per GPU, the worker pool of the reference (main.rs:93-120: threads-1 clones of
// SequenceParser popping packed records from a mutex-guarded queue, parse.rs:53-86) by one
// kernel launch per batch: one LANE per read (64 reads per wavefront), reads staged through
// LDS with 16-byte coalesced loads, outcome counters reduced per workgroup, one no-return
// global atomic per matched read into the dense (sample, barcode tuple) counter table that
// stands in for Results' nested HashMap (info.rs:661-665).
#include <hip/hip_runtime.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/barcode_count_hip.h"
#include "bc_lane.h"
#include "bc_plan.hpp"
#include "bc_synth.h"

namespace bc {

constexpr int kTPB = 256;          // reads (lanes) per workgroup
#ifndef BC_MIN_WAVES
#define BC_MIN_WAVES 4  // occupancy floor (waves per SIMD) the register allocator must honour
#endif

// ------------------------------------------------------------------------------------------------
// wave-level pieces
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t t = (uint32_t)__shfl_xor((int)v, o);
    v = t < v ? t : v;
  }
  return v;
}

// combines the lanes' partial unique-minimum searches: wave min-reduce of the key, then a vote
__device__ __forceinline__ uint32_t wave_verdict(const Nearest& s, uint32_t max_err) {
  const uint32_t kmin = wave_min_u32(s.key);
  const unsigned long long holders = __ballot(s.key == kmin);
  const int first = __ffsll(holders) - 1;
  const uint32_t cnt = (uint32_t)__shfl((int)s.count, first);
  const uint32_t idx = (uint32_t)__shfl((int)s.idx, first);
  const bool unique = __popcll(holders) == 1 && kmin != 0xFFFFFFFFu;
  return unique ? nearest_result(kmin, idx, cnt, max_err) : kFail;
}

// fix_error (parse.rs:553-593) for ONE capture by the whole wavefront: every lane scores the
// references j = lane, lane+64, ...; a wavefront min-reduce plus a vote decides best / ambiguous.
// use_exact: a reference that IS the capture wins outright (AHashSet::contains, parse.rs:457/489).
__device__ __forceinline__ uint32_t wave_fix_error(const DevGroup& G, uint32_t q1, uint32_t q2, uint32_t qn, uint32_t qx,
                                                   bool use_exact) {
  const uint32_t lane = __lane_id();
  Nearest s;
  nearest_init(s);
  for (uint32_t j = lane; j < G.n_refs; j += 64) {
    bool ex;
    const uint32_t d = ref_distance(q1, q2, qn, qx, G.len, G.r1[j], G.r2[j], G.rn[j], G.rlen[j], ex);
    nearest_add(s, d, j, ex && use_exact);
  }
  return wave_verdict(s, G.max_err);
}

// The same verdict for a large set through the pigeonhole seed index: only references that equal
// the capture on one of its max_err+1 blocks can be within the budget, and every reference within
// the budget is among them, so best / ambiguous come out exactly as from the full scan.
// The whole wavefront walks the capture's buckets, 256 entries (four 1-KiB loads) in flight; the
// entries carry the reference planes, so there is one memory round trip per batch.
// The capture must be free of 'N' / foreign bytes.
// Returns the smallest key (distance + 1, 0 for the capture itself), whether exactly one reference
// has it, and that reference.
__device__ __forceinline__ void wave_seeded_min(const DevGroup& G, uint32_t q1, uint32_t q2, uint32_t& kmin_out,
                                                bool& unique_out, uint32_t& idx_out) {
  const uint32_t nb = G.seed_nb, blen = G.seed_blen, bm = (1u << blen) - 1u;
  const uint32_t nbk = 1u << (2 * blen);
  const uint32_t lane = __lane_id();
  const uint4* entries = reinterpret_cast<const uint4*>(G.seed_list);
  Nearest s;
  nearest_init(s);
  for (uint32_t b = 0; b < nb; ++b) {
    const uint32_t val = ((q1 >> (b * blen)) & bm) | (((q2 >> (b * blen)) & bm) << blen);
    const uint32_t* off = G.seed_off + (size_t)b * (nbk + 1);
    const uint32_t beg = off[val], end = off[val + 1];
    const uint4* list = entries + (size_t)b * G.n_idx;
    // blocks before b on which a reference may not equal the capture (it was scored there)
    for (uint32_t i0 = beg; i0 < end; i0 += 256) {
      uint4 e[4];
      bool on[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t i = i0 + 64u * k + lane;
        on[k] = i < end;
        e[k] = list[on[k] ? i : beg];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t diff = (q1 ^ e[k].x) | (q2 ^ e[k].y);
        bool earlier = false;
        for (uint32_t p = 0; p < b; ++p) earlier = earlier || ((diff >> (p * blen)) & bm) == 0u;
        if (on[k] && !earlier) nearest_add(s, popc(diff), e[k].z, diff == 0u);
      }
    }
    // A reference with d mismatches has at most d spoiled blocks, so it has shown up by the time
    // blocks 0..d are done: once the best distance so far is <= b nothing nearer or equally near
    // can still be hiding in the later blocks.
    const uint32_t kmin = wave_min_u32(s.key);
    if (kmin != 0xFFFFFFFFu && kmin <= b + 1u) break;
  }
  for (uint32_t i = lane; i < G.n_odd; i += 64) {
    const uint32_t j = G.odd_list[i];
    bool ex;
    const uint32_t d = ref_distance(q1, q2, 0u, 0u, G.len, G.r1[j], G.r2[j], G.rn[j], G.rlen[j], ex);
    nearest_add(s, d, j, ex);
  }
  const uint32_t kmin = wave_min_u32(s.key);
  const unsigned long long holders = __ballot(s.key == kmin);
  const int first = __ffsll(holders) - 1;
  kmin_out = kmin;
  unique_out = __popcll(holders) == 1 && (uint32_t)__shfl((int)s.count, first) == 1u && kmin != 0xFFFFFFFFu;
  idx_out = (uint32_t)__shfl((int)s.idx, first);
}

// A capture with up to two 'N's against plain references: 'N' is free (parse.rs:569), so its
// distance to a reference is that of the capture with each N replaced by the reference's base
// there.  The nearest references of the capture are therefore those of its 4 (16) substitutions
// at the smallest of their minimum distances, and the match is unique iff exactly one substitution
// reaches that distance and does so uniquely (the argument of single_n_lookup, bc_lane.h).
__device__ __forceinline__ uint32_t wave_fix_error_seeded(const DevGroup& G, uint32_t q1, uint32_t q2, uint32_t qn) {
  const uint32_t n_n = popc(qn);
  const uint32_t p0 = n_n ? ctz(qn) : 0u;
  const uint32_t p1 = n_n > 1 ? ctz(qn & (qn - 1u)) : 0u;
  const uint32_t combos = 1u << (2u * n_n);
  const uint32_t b1 = q1 & ~qn, b2 = q2 & ~qn;
  uint32_t best = 0xFFFFFFFFu, res = kFail;
  bool ok = false;
  for (uint32_t c = 0; c < combos; ++c) {
    uint32_t s1 = b1, s2 = b2;
    if (n_n > 0) {
      s1 |= (c & 1u) << p0;
      s2 |= ((c >> 1) & 1u) << p0;
    }
    if (n_n > 1) {
      s1 |= ((c >> 2) & 1u) << p1;
      s2 |= ((c >> 3) & 1u) << p1;
    }
    uint32_t k, idx;
    bool uniq;
    wave_seeded_min(G, s1, s2, k, uniq, idx);
    if (k < best) {
      best = k;
      ok = uniq;
      res = idx;
    } else if (k == best) {
      ok = false;
    }
  }
  return (ok && best != 0xFFFFFFFFu && (best == 0u || best - 1u <= G.max_err)) ? res : kFail;
}

// Wave-private LDS tile: each wavefront stages the 64 reads it owns (64*stride contiguous bytes of
// the batch, 16 B per lane and instruction) and never needs a workgroup barrier to use them: LDS
// executes one wave's instructions in order, so only the compiler has to be told not to reorder.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void stage_tile(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t bytes,
                                           uint32_t region, uint8_t pad, uint32_t lane) {
  const uint4* g = reinterpret_cast<const uint4*>(src);
  uint4* d = reinterpret_cast<uint4*>(dst);
  const uint32_t n16 = bytes >> 4;
  for (uint32_t i0 = 0; i0 < n16; i0 += 256) {  // four independent 1-KiB wave loads in flight
    const uint32_t ia = i0 + lane, ib = ia + 64u, ic = ia + 128u, id = ia + 192u;
    const uint32_t last = n16 - 1u;
    const uint4 ra = g[ia < n16 ? ia : last];
    const uint4 rb = g[ib < n16 ? ib : last];
    const uint4 rc = g[ic < n16 ? ic : last];
    const uint4 rd = g[id < n16 ? id : last];
    if (ia < n16) d[ia] = ra;
    if (ib < n16) d[ib] = rb;
    if (ic < n16) d[ic] = rc;
    if (id < n16) d[id] = rd;
  }
  for (uint32_t i = (n16 << 4) + lane; i < bytes; i += 64) dst[i] = src[i];
  // bytes past the reads that the lane code may touch: plain values
  for (uint32_t i = bytes + lane; i < region; i += 64) dst[i] = pad;
}

// ---- device hash set of 64-bit keys (the AHashSet<String> per tuple of info.rs:663, flattened) ----
constexpr unsigned long long kEmptyKey = ~0ull;

// true when the key was not in the set before
__device__ __forceinline__ bool set_insert(unsigned long long* __restrict__ slots, uint64_t mask, uint64_t key) {
  uint64_t h = hash64(key) & mask;
  for (;;) {
    const unsigned long long old = atomicCAS(&slots[h], kEmptyKey, (unsigned long long)key);
    if (old == kEmptyKey) return true;
    if (old == key) return false;
    h = (h + 1) & mask;
  }
}

// slot of the key in a hash map, inserting it when absent
__device__ __forceinline__ uint64_t map_slot(unsigned long long* __restrict__ slots, uint64_t mask, uint64_t key) {
  uint64_t h = hash64(key) & mask;
  for (;;) {
    const unsigned long long old = atomicCAS(&slots[h], kEmptyKey, (unsigned long long)key);
    if (old == kEmptyKey || old == key) return h;
    h = (h + 1) & mask;
  }
}

struct DeviceOps {
  uint8_t* tile;          // this wave's LDS region
  const uint8_t* qsrc;    // this wave's quality lines in global memory
  uint32_t bytes, region, lane;

  __device__ __forceinline__ bool any(bool c) const { return __any(c) != 0; }
  // the sequence bytes are dead once the planes are built: reuse the tile for the quality lines
  __device__ __forceinline__ const uint32_t* stage_quality() const {
    wave_lds_fence();
    stage_tile(tile, qsrc, bytes, region, (uint8_t)'I', lane);
    wave_lds_fence();
    return reinterpret_cast<const uint32_t*>(tile);
  }
  // every lane calls this; lanes with `need` get their capture resolved one after the other
  __device__ __forceinline__ uint32_t nearest(const DevGroup& G, uint32_t q1, uint32_t q2, uint32_t qn, uint32_t qx,
                                              bool need) const {
    uint32_t out = kFail;
    unsigned long long todo = __ballot(need);
    while (todo) {
      const int src = __ffsll(todo) - 1;
      todo &= todo - 1;
      const uint32_t b1 = (uint32_t)__shfl((int)q1, src);
      const uint32_t b2 = (uint32_t)__shfl((int)q2, src);
      const uint32_t bn = (uint32_t)__shfl((int)qn, src);
      const uint32_t bx = (uint32_t)__shfl((int)qx, src);
      // the seed index answers captures with at most two 'N's when every reference is plain
      const bool seeded = G.seed_nb && bx == 0u && G.n_odd == 0u && __popc(bn) <= 2;
      const uint32_t r = seeded ? wave_fix_error_seeded(G, b1, b2, bn) : wave_fix_error(G, b1, b2, bn, bx, true);
      if (lane == (uint32_t)src) out = r;
    }
    return out;
  }
};

// ------------------------------------------------------------------------------------------------
// the hot kernel: SequenceParser::parse, 64 reads per wavefront, 4 wavefronts per workgroup
// ------------------------------------------------------------------------------------------------
template <int NW, int NWW>
__global__ __launch_bounds__(kTPB, ((NW <= 4 && NWW <= 2) ? BC_MIN_WAVES : 1)) void match_count_kernel(const DevPlan* __restrict__ plp,
                                                           const uint8_t* __restrict__ seq,
                                                           const uint8_t* __restrict__ qual,
                                                           const uint16_t* __restrict__ lens, uint32_t stride,
                                                           uint32_t read_len, uint32_t nd, uint64_t n_reads,
                                                           uint32_t region, uint32_t* __restrict__ table,
                                                           unsigned long long* __restrict__ slots, uint32_t* __restrict__ vals,
                                                           uint64_t smask,
                                                           unsigned long long* __restrict__ counters,
                                                           uint8_t* __restrict__ trace_outcome,
                                                           uint64_t* __restrict__ trace_idx) {
  extern __shared__ uint4 smem[];
  __shared__ uint32_t s_cnt[BC_NCOUNTERS];
  const DevPlan& pl = *plp;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  if (tid < BC_NCOUNTERS) s_cnt[tid] = 0;

  DeviceOps ops;
  ops.tile = reinterpret_cast<uint8_t*>(smem) + wave * region;
  ops.region = region;
  ops.lane = lane;

  // Persistent wavefronts: wave-tile t = 64 consecutive reads; this wave takes tiles
  // gid, gid + G, gid + 2G, ...  No workgroup barrier inside the loop; the outcome counters stay in
  // (scalar) registers until the very end, so the six global counters see one add per workgroup.
  const uint64_t n_tiles = (n_reads + 63u) >> 6;
  const uint64_t n_waves = (uint64_t)gridDim.x * (kTPB / 64);
  uint32_t acc_cnt[BC_NCOUNTERS];
#pragma unroll
  for (int k = 0; k < BC_NCOUNTERS; ++k) acc_cnt[k] = 0;
  for (uint64_t t = (uint64_t)blockIdx.x * (kTPB / 64) + wave; t < n_tiles; t += n_waves) {
    const uint64_t wfirst = t << 6;
    const uint32_t n_w = n_reads - wfirst < 64u ? (uint32_t)(n_reads - wfirst) : 64u;
    const uint64_t goff = wfirst * stride;  // multiple of 64 bytes: 16-byte aligned
    ops.qsrc = qual + goff;
    ops.bytes = n_w * stride;
    wave_lds_fence();  // the previous tile's last LDS reads are done before the tile is overwritten
    if (!(pl.ablate & 0x80u)) stage_tile(ops.tile, seq + goff, ops.bytes, region, (uint8_t)'A', lane);
    wave_lds_fence();

    const bool active = lane < n_w;
    uint32_t len = 0;
    if (active) len = lens ? (uint32_t)lens[wfirst + lane] : read_len;
    const uint32_t base = (active ? lane : 0u) * stride;
    const ReadResult r =
        process_read<DeviceOps, NW, NWW>(pl, ops, reinterpret_cast<const uint32_t*>(ops.tile), base, len, nd, active);

    uint32_t outcome = r.outcome;
    if (pl.has_random) {
      // Results::add_count with a random barcode (info.rs:770-802): insert (tuple, random) into the
      // set; an element already present makes the read a duplicate (parse.rs:65-69)
      if (active && outcome == kMatched && !set_insert(slots, smask, r.dense_idx * pl.rspace + r.rcode))
        outcome = kDuplicate;
    } else if (active && outcome == kMatched && !pl.discard_counts && !(pl.ablate & 0x4u)) {
      // Results::add_count (info.rs:761-767): one no-return atomic into the dense counter table, or,
      // when captures are kept raw, into the hash map slot of the tuple key
      if (pl.sparse)
        atomicAdd(&vals[map_slot(slots, smask, r.dense_idx)], 1u);
      else
        atomicAdd(&table[r.dense_idx], 1u);
    }
    // outcome counters (SequenceErrors, info.rs:16-139)
#pragma unroll
    for (uint32_t k = 0; k < BC_NCOUNTERS; ++k) {
      if (k == BC_TOTAL_READS) continue;
      acc_cnt[k] += (uint32_t)__popcll(__ballot(active && outcome == k));
    }
    acc_cnt[BC_TOTAL_READS] += n_w;
    if (trace_outcome && active) {
      trace_outcome[wfirst + lane] = (uint8_t)outcome;
      trace_idx[wfirst + lane] = pl.has_random ? r.dense_idx * pl.rspace + r.rcode : r.dense_idx;
    }
  }

  __syncthreads();  // s_cnt zeroed
  if (lane == 0) {
#pragma unroll
    for (uint32_t k = 0; k < BC_NCOUNTERS; ++k)
      if (acc_cnt[k]) atomicAdd(&s_cnt[k], acc_cnt[k]);
  }
  __syncthreads();
  if (tid < BC_NCOUNTERS) {
    const uint32_t v = s_cnt[tid];
    if (v) atomicAdd(&counters[tid], (unsigned long long)v);
  }
}

// ------------------------------------------------------------------------------------------------
// plan-time kernels
// ------------------------------------------------------------------------------------------------
// correction table of a short barcode: fix_error's verdict for every N-free capture
__global__ void build_dtable_kernel(const DevPlan* __restrict__ plp, uint32_t g, uint32_t* __restrict__ out) {
  const DevGroup& G = plp->groups[g];
  const uint32_t nq = 1u << (2 * G.len);
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  out[q] = dtable_entry(G, q);
}

// bc_fix_error: one wavefront, one query, plain fix_error semantics (no exact-member shortcut)
__global__ void fix_error_kernel(DevGroup G, uint32_t q1, uint32_t q2, uint32_t qn, uint32_t qx, uint32_t* out) {
  const uint32_t r = wave_fix_error(G, q1, q2, qn, qx, false);
  if (threadIdx.x == 0) *out = r;
}

__global__ void set_fill_kernel(unsigned long long* slots, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) slots[i] = kEmptyKey;
}

// re-inserts every key of `src` (n slots, or n plain keys when `dense`) into dst; counts the new ones
__global__ void set_insert_kernel(const unsigned long long* __restrict__ src, uint64_t n, unsigned long long* dst,
                                  uint64_t dmask, unsigned long long* n_new) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  uint32_t c = 0;
  for (; i < n; i += step) {
    const unsigned long long k = src[i];
    if (k != kEmptyKey && set_insert(dst, dmask, k)) ++c;
  }
  if (n_new) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
    if (__lane_id() == 0 && c) atomicAdd(n_new, (unsigned long long)c);
  }
}

// moves (key, value) pairs into a larger map
__global__ void map_rehash_kernel(const unsigned long long* __restrict__ src, const uint32_t* __restrict__ src_vals,
                                  uint64_t n, unsigned long long* dst, uint32_t* dst_vals, uint64_t dmask) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const unsigned long long k = src[i];
    if (k != kEmptyKey) atomicAdd(&dst_vals[map_slot(dst, dmask, k)], src_vals[i]);
  }
}

// raw captures + random barcode: count of a tuple = number of its distinct (tuple, random) keys
__global__ void set_to_map_kernel(const unsigned long long* __restrict__ slots, uint64_t n, uint64_t rspace,
                                  unsigned long long* dst, uint32_t* dst_vals, uint64_t dmask) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const unsigned long long k = slots[i];
    if (k != kEmptyKey) atomicAdd(&dst_vals[map_slot(dst, dmask, k / rspace)], 1u);
  }
}

__global__ void map_export_kernel(const unsigned long long* __restrict__ slots, const uint32_t* __restrict__ vals,
                                  uint64_t n, unsigned long long* cursor, uint64_t* __restrict__ out_key,
                                  uint32_t* __restrict__ out_cnt) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const unsigned long long k = slots[i];
    if (k != kEmptyKey) {
      const unsigned long long p = atomicAdd(cursor, 1ull);
      out_key[p] = k;
      out_cnt[p] = vals[i];
    }
  }
}

// compacts the keys of a set into out[0 .. *cursor)
__global__ void set_export_kernel(const unsigned long long* __restrict__ slots, uint64_t n, unsigned long long* cursor,
                                  unsigned long long* __restrict__ out, uint64_t capacity) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const unsigned long long k = slots[i];
    if (k != kEmptyKey) {
      const unsigned long long p = atomicAdd(cursor, 1ull);
      if (p < capacity) out[p] = k;
    }
  }
}

// final counts with a random barcode = number of distinct random barcodes per tuple (output.rs:265-270)
__global__ void set_to_table_kernel(const unsigned long long* __restrict__ slots, uint64_t n, uint64_t rspace,
                                    uint32_t* __restrict__ table) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const unsigned long long k = slots[i];
    if (k != kEmptyKey) atomicAdd(&table[k / rspace], 1u);
  }
}

__global__ void count_nonzero_kernel(const uint32_t* __restrict__ table, uint64_t n, unsigned long long* total) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  uint32_t c = 0;
  for (; i < n; i += step) c += table[i] != 0u;
  // wave sum
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
  if (__lane_id() == 0 && c) atomicAdd(total, (unsigned long long)c);
}

__global__ void compact_kernel(const uint32_t* __restrict__ table, uint64_t n, unsigned long long* cursor,
                               uint64_t* __restrict__ out_idx, uint32_t* __restrict__ out_cnt) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const uint32_t v = table[i];
    if (v) {
      const unsigned long long p = atomicAdd(cursor, 1ull);
      out_idx[p] = i;
      out_cnt[p] = v;
    }
  }
}

__global__ void synth_kernel(SynthDev S, uint64_t first, uint64_t n, uint8_t* __restrict__ seq,
                             uint8_t* __restrict__ qual, uint32_t stride) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint64_t i = first + t;
  SynthRead R;
  synth_begin(S, i, R);
  uint8_t* so = seq + t * stride;
  uint8_t* qo = qual ? qual + t * stride : nullptr;
  for (uint32_t p = 0; p < S.read_len; ++p) {
    uint8_t b, q;
    synth_byte(S, i, R, p, b, q);
    so[p] = b;
    if (qo) qo[p] = q;
  }
  for (uint32_t p = S.read_len; p < stride; ++p) {
    so[p] = '\n';
    if (qo) qo[p] = '\n';
  }
}

}  // namespace bc

using namespace bc;

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                        \
      return BC_ERR_HIP;                                                                   \
    }                                                                                      \
  } while (0)

struct bc_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipStream_t copy_stream = nullptr;
  HostDevPlan h;
  DevPlan* d_plan = nullptr;
  std::vector<void*> allocs;
  uint32_t* d_table = nullptr;
  bool own_table = false;
  uint64_t table_entries = 0;
  unsigned long long* d_counters = nullptr;
  uint32_t barcode_num = 0;
  uint32_t n_sets[kMaxGroups] = {0};
  std::vector<std::vector<std::string>> set_seqs;  // per group: the known sequences in index order
  bool has_sample_group = false;
  uint8_t* trace_outcome = nullptr;
  uint64_t* trace_idx = nullptr;
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  double ms_total = 0.0;
  uint64_t launches = 0;
  // compacted results
  std::vector<uint64_t> row_idx;
  std::vector<uint32_t> row_cnt;
  // host staging (bc_engine_submit_host)
  static constexpr int kStages = 2;
  uint8_t* pin[kStages] = {nullptr, nullptr};
  uint8_t* dev[kStages] = {nullptr, nullptr};
  hipEvent_t copied[kStages] = {nullptr, nullptr};
  hipEvent_t consumed[kStages] = {nullptr, nullptr};
  size_t stage_bytes = 0;
  uint32_t lds_limit = 0;
  uint32_t n_cus = 0;
  // random-barcode mode: the hash set of (tuple, random barcode) keys
  unsigned long long* d_slots = nullptr;
  uint32_t* d_vals = nullptr;  // sparse plans without a random barcode: the count of each key
  uint64_t n_slots = 0;
  uint64_t key_bound = 0;  // upper bound on the keys held: reads submitted / keys imported so far
};

static int upload(bc_engine* e, const void* src, size_t bytes, void** out) {
  void* d = nullptr;
  HIP_TRY(hipMalloc(&d, bytes ? bytes : 16));
  e->allocs.push_back(d);
  if (bytes) HIP_TRY(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
  *out = d;
  return BC_OK;
}

static void engine_free(bc_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  for (auto& ev : e->events) {
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  for (int s = 0; s < bc_engine::kStages; ++s) {
    if (e->pin[s]) (void)hipHostFree(e->pin[s]);
    if (e->dev[s]) (void)hipFree(e->dev[s]);
    if (e->copied[s]) (void)hipEventDestroy(e->copied[s]);
    if (e->consumed[s]) (void)hipEventDestroy(e->consumed[s]);
  }
  for (void* p : e->allocs) (void)hipFree(p);
  if (e->own_table && e->d_table) (void)hipFree(e->d_table);
  if (e->d_slots) (void)hipFree(e->d_slots);
  if (e->d_vals) (void)hipFree(e->d_vals);
  if (e->d_counters) (void)hipFree(e->d_counters);
  if (e->d_plan) (void)hipFree(e->d_plan);
  if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
  if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

static int engine_init(bc_engine* e, const bc_plan* p, int device_id, void* hip_stream, void* table_mem) {
  if (!p->lower(e->h)) return BC_ERR_UNSUPPORTED;
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev) {
    set_error("bc_engine_create: no HIP device " + std::to_string(device_id) + " (the engine has no CPU path)");
    return BC_ERR_HIP;
  }
  e->device = device_id;
  HIP_TRY(hipSetDevice(device_id));
  if (hip_stream) {
    e->stream = (hipStream_t)hip_stream;
  } else {
    HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    e->own_stream = true;
  }
  HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_id));
  e->lds_limit = (uint32_t)prop.maxSharedMemoryPerMultiProcessor;
  e->n_cus = (uint32_t)prop.multiProcessorCount;
  e->barcode_num = p->barcode_num;
  e->has_sample_group = p->sample_barcode;
  {
    // groups are ordered sample first, then counted barcodes (bc_plan::lower)
    if (p->sample_barcode) e->set_seqs.push_back(p->samples.seqs);
    for (uint32_t b = 0; b < p->barcode_num; ++b) e->set_seqs.push_back(p->counted[b].seqs);
  }

  DevPlan& P = e->h.plan;
  if (const char* ab = getenv("BC_ABLATE")) P.ablate = (uint32_t)strtoul(ab, nullptr, 0);
  for (uint32_t g = 0; g < P.n_groups; ++g) {
    DevGroup& G = P.groups[g];
    HostSet& H = e->h.sets[g];
    e->n_sets[g] = G.n_refs;
    int rc;
    if ((rc = upload(e, H.r1.data(), H.r1.size() * 4, (void**)&G.r1))) return rc;
    if ((rc = upload(e, H.r2.data(), H.r2.size() * 4, (void**)&G.r2))) return rc;
    if ((rc = upload(e, H.rn.data(), H.rn.size() * 4, (void**)&G.rn))) return rc;
    if ((rc = upload(e, H.rlen.data(), H.rlen.size(), (void**)&G.rlen))) return rc;
    if (G.mode == kSetHash) {
      if ((rc = upload(e, H.hkeys.data(), H.hkeys.size() * 8, (void**)&G.hkeys))) return rc;
      if ((rc = upload(e, H.hvals.data(), H.hvals.size() * 4, (void**)&G.hvals))) return rc;
      if (G.seed_nb) {
        if ((rc = upload(e, H.seed_off.data(), H.seed_off.size() * 4, (void**)&G.seed_off))) return rc;
        if ((rc = upload(e, H.seed_list.data(), H.seed_list.size() * 4, (void**)&G.seed_list))) return rc;
        if ((rc = upload(e, H.odd_list.data(), H.odd_list.size() * 4, (void**)&G.odd_list))) return rc;
      }
    }
    if (G.mode == kSetDirect) {
      void* d = nullptr;
      HIP_TRY(hipMalloc(&d, (size_t)4 << (2 * G.len)));
      e->allocs.push_back(d);
      G.dtable = (const uint32_t*)d;
    }
  }
  HIP_TRY(hipMalloc((void**)&e->d_plan, sizeof(DevPlan)));
  HIP_TRY(hipMemcpy(e->d_plan, &P, sizeof(DevPlan), hipMemcpyHostToDevice));
  for (uint32_t g = 0; g < P.n_groups; ++g) {
    const DevGroup& G = P.groups[g];
    if (G.mode != kSetDirect) continue;
    const uint32_t nq = 1u << (2 * G.len);
    hipLaunchKernelGGL(build_dtable_kernel, dim3((nq + 255) / 256), dim3(256), 0, e->stream, e->d_plan, g,
                       const_cast<uint32_t*>(G.dtable));
    HIP_TRY(hipGetLastError());
  }
  e->table_entries = P.sparse ? 0 : e->h.table_entries;
  if (P.sparse) {
    if (table_mem) {
      set_error("bc_engine_create: the plan keeps raw captures (no dense table); pass table_mem = NULL");
      return BC_ERR_INVALID;
    }
  } else if (table_mem) {
    e->d_table = (uint32_t*)table_mem;
  } else {
    HIP_TRY(hipMalloc((void**)&e->d_table, e->table_entries * 4));
    e->own_table = true;
    HIP_TRY(hipMemsetAsync(e->d_table, 0, e->table_entries * 4, e->stream));
  }
  HIP_TRY(hipMalloc((void**)&e->d_counters, BC_NCOUNTERS * 8));
  HIP_TRY(hipMemsetAsync(e->d_counters, 0, BC_NCOUNTERS * 8, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return BC_OK;
}

template <int NW, int NWW>
static int launch_match(bc_engine* e, const void* d_seq, const void* d_qual, const void* d_lens, uint32_t stride,
                        uint32_t read_len, uint32_t nd, uint64_t n_reads, uint64_t trace_off) {
  // per-wave LDS region: 64 reads + the few bytes past them the lane code may touch
  const uint32_t L = e->h.plan.L;
  const uint32_t slack = (uint32_t)NW * 32u + 16u + (L > stride ? L - stride : 0u);
  const uint32_t tile_alloc = (64u * stride + slack + 15u) & ~15u;
  const uint32_t lds = tile_alloc * (kTPB / 64);
  if (lds + 64 > e->lds_limit) {
    set_error("read stride too large for one LDS tile");
    return BC_ERR_UNSUPPORTED;
  }
  auto kern = match_count_kernel<NW, NWW>;
  if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // persistent grid: as many workgroups as the chip holds at once (or fewer for a small batch)
  int per_cu = 0;
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, kTPB, lds));
  if (per_cu < 1) per_cu = 1;
  const uint64_t resident = (uint64_t)per_cu * e->n_cus;
  const uint64_t blocks = std::min<uint64_t>((n_reads + kTPB - 1) / kTPB, resident);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (e->timing) {
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, e->stream));
  }
  hipLaunchKernelGGL(kern, dim3((uint32_t)blocks), dim3(kTPB), lds, e->stream, e->d_plan, (const uint8_t*)d_seq,
                     (const uint8_t*)d_qual, (const uint16_t*)d_lens, stride, read_len, nd, n_reads, tile_alloc,
                     e->d_table, e->d_slots, e->d_vals, e->n_slots ? e->n_slots - 1 : 0, e->d_counters, e->trace_outcome ? e->trace_outcome + trace_off : nullptr,
                     e->trace_idx ? e->trace_idx + trace_off : nullptr);
  HIP_TRY(hipGetLastError());
  if (e->timing) {
    HIP_TRY(hipEventRecord(e1, e->stream));
    e->events.emplace_back(e0, e1);
  }
  return BC_OK;
}

static uint32_t grid_for(uint64_t n) { return (uint32_t)std::min<uint64_t>((n + 255) / 256, 256ull * 32); }

// Random-barcode mode: keeps the hash set at most half full for `more` further keys.  Growing
// allocates a set of twice the size (or more) and re-inserts the old keys on the device.
static int set_reserve(bc_engine* e, uint64_t more) {
  const DevPlan& P = e->h.plan;
  if (!P.has_random && !P.sparse) return BC_OK;
  const bool with_vals = P.sparse && !P.has_random;
  e->key_bound += more;
  uint64_t want = 1ull << 20;
  while (want < 2 * e->key_bound) want <<= 1;
  if (want <= e->n_slots) return BC_OK;
  unsigned long long* fresh = nullptr;
  uint32_t* fresh_vals = nullptr;
  HIP_TRY(hipMalloc((void**)&fresh, want * 8));
  hipLaunchKernelGGL(set_fill_kernel, dim3(grid_for(want)), dim3(256), 0, e->stream, fresh, want);
  if (with_vals) {
    HIP_TRY(hipMalloc((void**)&fresh_vals, want * 4));
    HIP_TRY(hipMemsetAsync(fresh_vals, 0, want * 4, e->stream));
  }
  if (e->d_slots) {
    if (with_vals)
      hipLaunchKernelGGL(map_rehash_kernel, dim3(grid_for(e->n_slots)), dim3(256), 0, e->stream, e->d_slots, e->d_vals,
                         e->n_slots, fresh, fresh_vals, want - 1);
    else
      hipLaunchKernelGGL(set_insert_kernel, dim3(grid_for(e->n_slots)), dim3(256), 0, e->stream, e->d_slots, e->n_slots,
                         fresh, want - 1, (unsigned long long*)nullptr);
    HIP_TRY(hipStreamSynchronize(e->stream));
    (void)hipFree(e->d_slots);
    if (e->d_vals) (void)hipFree(e->d_vals);
  }
  HIP_TRY(hipGetLastError());
  e->d_slots = fresh;
  e->d_vals = fresh_vals;
  e->n_slots = want;
  return BC_OK;
}

static int submit_device_impl(bc_engine* e, const void* d_seq, const void* d_qual, const void* d_lens, uint32_t stride,
                              uint32_t read_len, uint64_t n_reads, uint64_t trace_off) {
  if (n_reads == 0) return BC_OK;
  if (!d_seq || stride == 0) {
    set_error("submit: null sequence buffer or zero stride");
    return BC_ERR_INVALID;
  }
  if (e->h.plan.quality_on && !d_qual) {
    set_error("submit: the quality filter is on (--min-quality > 0) but no quality buffer was given");
    return BC_ERR_INVALID;
  }
  if (((uintptr_t)d_seq & 15) || ((uintptr_t)d_qual & 15)) {
    set_error("submit: buffers must be 16-byte aligned");
    return BC_ERR_INVALID;
  }
  const uint32_t maxlen = d_lens ? stride : read_len;
  if (maxlen > stride) {
    set_error("submit: read_len exceeds stride");
    return BC_ERR_INVALID;
  }
  const uint32_t nd = (maxlen + 3) / 4;
  HIP_TRY(hipSetDevice(e->device));
  {
    const int rc = set_reserve(e, n_reads);
    if (rc != BC_OK) return rc;
  }
  // candidate offsets 0 .. maxlen-L: how many 32-bit words the anchor / repair vectors need
  const uint32_t L = e->h.plan.L;
  const uint32_t nww = maxlen >= L ? (maxlen - L + 1 + 31) / 32 : 1;
#define BC_LAUNCH(NW_, NWW_) return launch_match<NW_, NWW_>(e, d_seq, d_qual, d_lens, stride, read_len, nd, n_reads, trace_off)
  if (maxlen <= 128) {
    if (nww <= 1) BC_LAUNCH(4, 1);
    if (nww <= 2) BC_LAUNCH(4, 2);
    BC_LAUNCH(4, 4);
  }
  if (maxlen <= 256) {
    if (nww <= 2) BC_LAUNCH(8, 2);
    if (nww <= 4) BC_LAUNCH(8, 4);
    BC_LAUNCH(8, 8);
  }
  if (maxlen <= 320) {
    if (nww <= 4) BC_LAUNCH(10, 4);
    BC_LAUNCH(10, 10);
  }
#undef BC_LAUNCH
  set_error("submit: reads longer than 320 bases are not supported");
  return BC_ERR_UNSUPPORTED;
}

extern "C" {

bc_engine* bc_engine_create(const bc_plan* p, int device_id, void* hip_stream, void* table_mem) {
  if (!p) {
    set_error("bc_engine_create: null plan");
    return nullptr;
  }
  bc_engine* e = new bc_engine();
  const int rc = engine_init(e, p, device_id, hip_stream, table_mem);
  if (rc != BC_OK) {
    const std::string msg = get_error();
    engine_free(e);
    set_error(msg);
    return nullptr;
  }
  return e;
}

void bc_engine_destroy(bc_engine* e) { engine_free(e); }

int bc_engine_submit_device(bc_engine* e, const void* d_seq, const void* d_qual, const void* d_lens, uint32_t stride,
                            uint32_t read_len, uint64_t n_reads) {
  return submit_device_impl(e, d_seq, d_qual, d_lens, stride, read_len, n_reads, 0);
}

int bc_engine_submit_host(bc_engine* e, const void* seq, const void* qual, const uint16_t* lens, uint32_t stride,
                          uint32_t read_len, uint64_t n_reads) {
  if (n_reads == 0) return BC_OK;
  if (!seq || stride == 0) {
    set_error("submit: null sequence buffer or zero stride");
    return BC_ERR_INVALID;
  }
  const bool with_qual = e->h.plan.quality_on != 0;
  if (with_qual && !qual) {
    set_error("submit: the quality filter is on (--min-quality > 0) but no quality buffer was given");
    return BC_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(e->device));
  // chunk = a multiple of 256 reads so that every chunk's arrays stay 16-byte aligned
  const uint64_t chunk = std::max<uint64_t>(256, ((64ull << 20) / (stride * (with_qual ? 2 : 1) + 2)) & ~255ull);
  const size_t seq_bytes = (size_t)chunk * stride;
  const size_t need = seq_bytes * (with_qual ? 2 : 1) + chunk * 2 + 64;
  if (e->stage_bytes < need) {
    for (int s = 0; s < bc_engine::kStages; ++s) {
      if (e->pin[s]) (void)hipHostFree(e->pin[s]);
      if (e->dev[s]) (void)hipFree(e->dev[s]);
      e->pin[s] = nullptr;
      e->dev[s] = nullptr;
      HIP_TRY(hipHostMalloc((void**)&e->pin[s], need, hipHostMallocDefault));
      HIP_TRY(hipMalloc((void**)&e->dev[s], need));
      if (!e->copied[s]) HIP_TRY(hipEventCreateWithFlags(&e->copied[s], hipEventDisableTiming));
      if (!e->consumed[s]) HIP_TRY(hipEventCreateWithFlags(&e->consumed[s], hipEventDisableTiming));
      HIP_TRY(hipEventRecord(e->consumed[s], e->stream));
    }
    e->stage_bytes = need;
  }
  uint64_t done = 0;
  int s = 0;
  while (done < n_reads) {
    const uint64_t n = std::min(chunk, n_reads - done);
    // the pinned buffer is free once its previous H2D copy is done, the device buffer once the
    // kernel that read it has finished
    HIP_TRY(hipEventSynchronize(e->copied[s]));
    uint8_t* hp = e->pin[s];
    const size_t sb = (size_t)n * stride;
    const size_t q_off = with_qual ? ((sb + 15) & ~(size_t)15) : 0;
    const size_t l_off = q_off + (with_qual ? ((sb + 15) & ~(size_t)15) : ((sb + 15) & ~(size_t)15));
    memcpy(hp, (const uint8_t*)seq + done * stride, sb);
    if (with_qual) memcpy(hp + q_off, (const uint8_t*)qual + done * stride, sb);
    if (lens) memcpy(hp + l_off, lens + done, n * 2);
    const size_t total = l_off + (lens ? n * 2 : 0);
    HIP_TRY(hipStreamWaitEvent(e->copy_stream, e->consumed[s], 0));
    HIP_TRY(hipMemcpyAsync(e->dev[s], hp, total, hipMemcpyHostToDevice, e->copy_stream));
    HIP_TRY(hipEventRecord(e->copied[s], e->copy_stream));
    HIP_TRY(hipStreamWaitEvent(e->stream, e->copied[s], 0));
    const int rc = submit_device_impl(e, e->dev[s], with_qual ? e->dev[s] + q_off : nullptr,
                                      lens ? e->dev[s] + l_off : nullptr, stride, read_len, n, done);
    if (rc != BC_OK) return rc;
    HIP_TRY(hipEventRecord(e->consumed[s], e->stream));
    done += n;
    s ^= 1;
  }
  // the caller may reuse its buffers on return: they were copied into pinned memory above
  return BC_OK;
}

int bc_engine_sync(bc_engine* e) {
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->copy_stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return BC_OK;
}

int bc_engine_reset(bc_engine* e) {
  HIP_TRY(hipSetDevice(e->device));
  if (e->d_table) HIP_TRY(hipMemsetAsync(e->d_table, 0, e->table_entries * 4, e->stream));
  HIP_TRY(hipMemsetAsync(e->d_counters, 0, BC_NCOUNTERS * 8, e->stream));
  if (e->d_slots) {
    hipLaunchKernelGGL(set_fill_kernel, dim3(grid_for(e->n_slots)), dim3(256), 0, e->stream, e->d_slots, e->n_slots);
    HIP_TRY(hipGetLastError());
    if (e->d_vals) HIP_TRY(hipMemsetAsync(e->d_vals, 0, e->n_slots * 4, e->stream));
  }
  e->key_bound = 0;
  return BC_OK;
}

int bc_engine_counters(bc_engine* e, uint64_t out[BC_NCOUNTERS]) {
  int rc = bc_engine_sync(e);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(out, e->d_counters, BC_NCOUNTERS * 8, hipMemcpyDeviceToHost));
  return BC_OK;
}

void* bc_engine_table_ptr(bc_engine* e) { return e->d_table; }
void* bc_engine_counters_ptr(bc_engine* e) { return e->d_counters; }
uint64_t bc_engine_table_entries(const bc_engine* e) { return e->table_entries; }

int bc_engine_trace(bc_engine* e, void* d_outcome_u8, void* d_index_u64) {
  e->trace_outcome = (uint8_t*)d_outcome_u8;
  e->trace_idx = (uint64_t*)d_index_u64;
  return BC_OK;
}

// rows of a sparse plan: (tuple key, count) pairs straight out of the hash map
static int finish_sparse(bc_engine* e, uint64_t* n_rows) {
  const DevPlan& P = e->h.plan;
  e->row_idx.clear();
  e->row_cnt.clear();
  if (n_rows) *n_rows = 0;
  if (!e->d_slots) return BC_OK;
  unsigned long long* keys = e->d_slots;
  uint32_t* vals = e->d_vals;
  uint64_t n_slots = e->n_slots;
  unsigned long long* agg_keys = nullptr;
  uint32_t* agg_vals = nullptr;
  if (P.has_random) {
    // count of a tuple = number of its distinct random barcodes (output.rs:265-270)
    HIP_TRY(hipMalloc((void**)&agg_keys, n_slots * 8));
    HIP_TRY(hipMalloc((void**)&agg_vals, n_slots * 4));
    hipLaunchKernelGGL(set_fill_kernel, dim3(grid_for(n_slots)), dim3(256), 0, e->stream, agg_keys, n_slots);
    HIP_TRY(hipMemsetAsync(agg_vals, 0, n_slots * 4, e->stream));
    hipLaunchKernelGGL(set_to_map_kernel, dim3(grid_for(n_slots)), dim3(256), 0, e->stream, e->d_slots, n_slots, P.rspace,
                       agg_keys, agg_vals, n_slots - 1);
    HIP_TRY(hipGetLastError());
    keys = agg_keys;
    vals = agg_vals;
  }
  unsigned long long* d_n = nullptr;
  uint64_t* d_key = nullptr;
  uint32_t* d_cnt = nullptr;
  HIP_TRY(hipMalloc((void**)&d_n, 8));
  HIP_TRY(hipMemsetAsync(d_n, 0, 8, e->stream));
  // upper bound on the rows: the keys held
  const uint64_t cap = std::min<uint64_t>(n_slots, e->key_bound ? e->key_bound : 1);
  HIP_TRY(hipMalloc((void**)&d_key, cap * 8));
  HIP_TRY(hipMalloc((void**)&d_cnt, cap * 4));
  hipLaunchKernelGGL(map_export_kernel, dim3(grid_for(n_slots)), dim3(256), 0, e->stream, keys, vals, n_slots, d_n, d_key,
                     d_cnt);
  unsigned long long n = 0;
  HIP_TRY(hipMemcpyAsync(&n, d_n, 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->row_idx.resize(n);
  e->row_cnt.resize(n);
  if (n) {
    HIP_TRY(hipMemcpy(e->row_idx.data(), d_key, n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(e->row_cnt.data(), d_cnt, n * 4, hipMemcpyDeviceToHost));
  }
  (void)hipFree(d_n);
  (void)hipFree(d_key);
  (void)hipFree(d_cnt);
  if (agg_keys) (void)hipFree(agg_keys);
  if (agg_vals) (void)hipFree(agg_vals);
  if (n_rows) *n_rows = n;
  return BC_OK;
}

int bc_engine_finish(bc_engine* e, uint64_t* n_rows) {
  int rc = bc_engine_sync(e);
  if (rc) return rc;
  if (e->h.plan.sparse) return finish_sparse(e, n_rows);
  if (e->h.plan.has_random) {
    // the count of a tuple is the number of distinct random barcodes seen with it (output.rs:265-270)
    HIP_TRY(hipMemsetAsync(e->d_table, 0, e->table_entries * 4, e->stream));
    if (e->d_slots) {
      hipLaunchKernelGGL(set_to_table_kernel, dim3(grid_for(e->n_slots)), dim3(256), 0, e->stream, e->d_slots,
                         e->n_slots, e->h.plan.rspace, e->d_table);
      HIP_TRY(hipGetLastError());
    }
  }
  unsigned long long* d_n = nullptr;
  HIP_TRY(hipMalloc((void**)&d_n, 16));
  HIP_TRY(hipMemsetAsync(d_n, 0, 16, e->stream));
  const uint32_t grid = (uint32_t)std::min<uint64_t>((e->table_entries + 255) / 256, 256ull * 32);
  hipLaunchKernelGGL(count_nonzero_kernel, dim3(grid), dim3(256), 0, e->stream, e->d_table, e->table_entries, d_n);
  unsigned long long n = 0;
  HIP_TRY(hipMemcpyAsync(&n, d_n, 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->row_idx.resize(n);
  e->row_cnt.resize(n);
  if (n) {
    uint64_t* d_idx = nullptr;
    uint32_t* d_cnt = nullptr;
    HIP_TRY(hipMalloc((void**)&d_idx, n * 8));
    HIP_TRY(hipMalloc((void**)&d_cnt, n * 4));
    hipLaunchKernelGGL(compact_kernel, dim3(grid), dim3(256), 0, e->stream, e->d_table, e->table_entries, d_n + 1,
                       d_idx, d_cnt);
    HIP_TRY(hipMemcpyAsync(e->row_idx.data(), d_idx, n * 8, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(e->row_cnt.data(), d_cnt, n * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    (void)hipFree(d_idx);
    (void)hipFree(d_cnt);
  }
  (void)hipFree(d_n);
  if (n_rows) *n_rows = n;
  return BC_OK;
}

int bc_engine_rows(bc_engine* e, uint64_t first, uint64_t n, uint32_t* sample_idx, uint32_t* barcode_idx,
                   uint64_t* count) {
  if (first + n > e->row_idx.size()) {
    set_error("bc_engine_rows: range outside the compacted rows (call bc_engine_finish first)");
    return BC_ERR_STATE;
  }
  const DevPlan& P = e->h.plan;
  if (P.sparse) {
    set_error("bc_engine_rows: the plan keeps raw captures, which have no index; use bc_engine_row_text");
    return BC_ERR_STATE;
  }
  const uint32_t nb = e->barcode_num ? e->barcode_num : 1;
  const uint32_t g0 = e->has_sample_group ? 1u : 0u;  // groups[0] is the sample group when present
  for (uint64_t r = 0; r < n; ++r) {
    uint64_t di = e->row_idx[first + r];
    for (int b = (int)e->barcode_num - 1; b >= 0; --b) {
      const uint32_t nr = P.groups[g0 + b].n_refs;
      barcode_idx[r * nb + b] = (uint32_t)(di % nr);
      di /= nr;
    }
    sample_idx[r] = (uint32_t)di;
    count[r] = e->row_cnt[first + r];
  }
  return BC_OK;
}

int bc_engine_key_count(bc_engine* e, uint64_t* n) {
  *n = 0;
  if (!e->h.plan.has_random || !e->d_slots) return BC_OK;
  return bc_engine_export_keys(e, nullptr, 0, n);
}

int bc_engine_export_keys(bc_engine* e, void* d_keys, uint64_t capacity, uint64_t* n) {
  *n = 0;
  if (!e->h.plan.has_random) {
    set_error("bc_engine_export_keys: the plan has no random barcode");
    return BC_ERR_STATE;
  }
  int rc = bc_engine_sync(e);
  if (rc) return rc;
  if (!e->d_slots) return BC_OK;
  unsigned long long* d_n = nullptr;
  HIP_TRY(hipMalloc((void**)&d_n, 8));
  HIP_TRY(hipMemsetAsync(d_n, 0, 8, e->stream));
  hipLaunchKernelGGL(set_export_kernel, dim3(grid_for(e->n_slots)), dim3(256), 0, e->stream, e->d_slots, e->n_slots, d_n,
                     (unsigned long long*)d_keys, d_keys ? capacity : 0);
  unsigned long long cnt = 0;
  HIP_TRY(hipMemcpyAsync(&cnt, d_n, 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  (void)hipFree(d_n);
  *n = cnt;
  if (d_keys && cnt > capacity) {
    set_error("bc_engine_export_keys: buffer too small");
    return BC_ERR_INVALID;
  }
  return BC_OK;
}

int bc_engine_import_keys(bc_engine* e, const void* d_keys, uint64_t n, uint64_t* n_new) {
  if (n_new) *n_new = 0;
  if (!e->h.plan.has_random) {
    set_error("bc_engine_import_keys: the plan has no random barcode");
    return BC_ERR_STATE;
  }
  if (n == 0) return BC_OK;
  HIP_TRY(hipSetDevice(e->device));
  int rc = set_reserve(e, n);
  if (rc) return rc;
  unsigned long long* d_n = nullptr;
  HIP_TRY(hipMalloc((void**)&d_n, 8));
  HIP_TRY(hipMemsetAsync(d_n, 0, 8, e->stream));
  hipLaunchKernelGGL(set_insert_kernel, dim3(grid_for(n)), dim3(256), 0, e->stream, (const unsigned long long*)d_keys, n,
                     e->d_slots, e->n_slots - 1, d_n);
  unsigned long long cnt = 0;
  HIP_TRY(hipMemcpyAsync(&cnt, d_n, 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  (void)hipFree(d_n);
  if (n_new) *n_new = cnt;
  return BC_OK;
}

int bc_engine_clear_keys(bc_engine* e) {
  HIP_TRY(hipSetDevice(e->device));
  if (e->d_slots) {
    hipLaunchKernelGGL(set_fill_kernel, dim3(grid_for(e->n_slots)), dim3(256), 0, e->stream, e->d_slots, e->n_slots);
    HIP_TRY(hipGetLastError());
  }
  e->key_bound = 0;
  return BC_OK;
}

int bc_engine_row_text(bc_engine* e, uint64_t row, char* sample, size_t sample_cap, char* tuple, size_t tuple_cap,
                       uint64_t* count) {
  if (row >= e->row_idx.size()) {
    set_error("bc_engine_row_text: row outside the compacted rows (call bc_engine_finish first)");
    return BC_ERR_STATE;
  }
  const DevPlan& P = e->h.plan;
  uint64_t key = e->row_idx[row];
  std::string s_out = "barcode", t_out;  // parse.rs:473: the sample key without a sample group
  std::vector<std::string> parts(P.n_groups);
  for (int g = (int)P.n_groups - 1; g >= 0; --g) {
    const DevGroup& G = P.groups[g];
    std::string v;
    if (G.mode == kSetNone) {
      uint64_t radix = 1;
      for (uint32_t k = 0; k < G.len; ++k) radix *= 5;
      uint64_t code = key % radix;
      key /= radix;
      for (uint32_t k = 0; k < G.len; ++k) {
        v.push_back("ACTGN"[code % 5]);
        code /= 5;
      }
    } else {
      const uint32_t idx = (uint32_t)(key % G.n_refs);
      key /= G.n_refs;
      v = e->set_seqs[g][idx];
    }
    parts[g] = v;
  }
  for (uint32_t g = 0; g < P.n_groups; ++g) {
    if (P.groups[g].type == kGroupSample)
      s_out = parts[g];
    else
      t_out += (t_out.empty() ? "" : ",") + parts[g];
  }
  if (s_out.size() + 1 > sample_cap || t_out.size() + 1 > tuple_cap) {
    set_error("bc_engine_row_text: buffer too small");
    return BC_ERR_INVALID;
  }
  memcpy(sample, s_out.c_str(), s_out.size() + 1);
  memcpy(tuple, t_out.c_str(), t_out.size() + 1);
  if (count) *count = e->row_cnt[row];
  return BC_OK;
}

int bc_engine_timing(bc_engine* e, int enable) {
  e->timing = enable != 0;
  return BC_OK;
}

int bc_engine_kernel_ms(bc_engine* e, double* total_ms, uint64_t* launches) {
  int rc = bc_engine_sync(e);
  if (rc) return rc;
  for (auto& ev : e->events) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
    e->ms_total += ms;
    e->launches++;
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  e->events.clear();
  if (total_ms) *total_ms = e->ms_total;
  if (launches) *launches = e->launches;
  e->ms_total = 0.0;
  e->launches = 0;
  return BC_OK;
}

// classify one ASCII string into planes; returns false if longer than 32
static bool pack_query(const char* s, uint32_t& q1, uint32_t& q2, uint32_t& qn, uint32_t& qx, uint32_t& len) {
  q1 = q2 = qn = qx = 0;
  len = (uint32_t)strlen(s);
  if (len > 32) return false;
  for (uint32_t i = 0; i < len; ++i) {
    const char c = s[i];
    if (c == 'N')
      qn |= 1u << i;
    else if (c == 'A' || c == 'C' || c == 'G' || c == 'T') {
      q1 |= (uint32_t)((c >> 1) & 1) << i;
      q2 |= (uint32_t)((c >> 2) & 1) << i;
    } else {
      qx |= 1u << i;
    }
  }
  return true;
}

int64_t bc_fix_error(const char* mismatch_seq, const char* const* possible_seqs, uint64_t n, uint16_t mismatches,
                     int device_id) {
  uint32_t q1, q2, qn, qx, qlen;
  if (!pack_query(mismatch_seq, q1, q2, qn, qx, qlen)) {
    set_error("bc_fix_error: sequences longer than 32 bases are not supported");
    return BC_ERR_UNSUPPORTED - 1;
  }
  std::vector<uint32_t> r1(n), r2(n), rn(n);
  std::vector<uint8_t> rl(n);
  for (uint64_t j = 0; j < n; ++j) {
    uint32_t x, l;
    if (!pack_query(possible_seqs[j], r1[j], r2[j], rn[j], x, l) || x) {
      set_error("bc_fix_error: candidates must be A,C,G,T,N strings of at most 32 bases");
      return BC_ERR_UNSUPPORTED - 1;
    }
    rl[j] = (uint8_t)l;
  }
  if (hipSetDevice(device_id) != hipSuccess) {
    set_error("bc_fix_error: no HIP device (the engine has no CPU path)");
    return BC_ERR_HIP - 1;
  }
  DevGroup G;
  memset(&G, 0, sizeof G);
  G.len = qlen;
  G.n_refs = (uint32_t)n;
  G.max_err = mismatches;
  void *d1 = nullptr, *d2 = nullptr, *dn = nullptr, *dl = nullptr, *dout = nullptr;
  bool ok = hipMalloc(&d1, n * 4 + 16) == hipSuccess && hipMalloc(&d2, n * 4 + 16) == hipSuccess &&
            hipMalloc(&dn, n * 4 + 16) == hipSuccess && hipMalloc(&dl, n + 16) == hipSuccess &&
            hipMalloc(&dout, 16) == hipSuccess;
  uint32_t out = kFail;
  if (ok && n) {
    ok = hipMemcpy(d1, r1.data(), n * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d2, r2.data(), n * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(dn, rn.data(), n * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(dl, rl.data(), n, hipMemcpyHostToDevice) == hipSuccess;
  }
  if (ok) {
    G.r1 = (const uint32_t*)d1;
    G.r2 = (const uint32_t*)d2;
    G.rn = (const uint32_t*)dn;
    G.rlen = (const uint8_t*)dl;
    hipLaunchKernelGGL(fix_error_kernel, dim3(1), dim3(64), 0, 0, G, q1, q2, qn, qx, (uint32_t*)dout);
    ok = hipGetLastError() == hipSuccess && hipMemcpy(&out, dout, 4, hipMemcpyDeviceToHost) == hipSuccess;
  }
  (void)hipFree(d1);
  (void)hipFree(d2);
  (void)hipFree(dn);
  (void)hipFree(dl);
  (void)hipFree(dout);
  if (!ok) {
    set_error(std::string("bc_fix_error: HIP failure: ") + hipGetErrorString(hipGetLastError()));
    return BC_ERR_HIP - 1;
  }
  return out == kFail ? -1 : (int64_t)out;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// synthetic workloads
// ------------------------------------------------------------------------------------------------
struct bc_synth {
  SynthDev S;                       // host view (refs -> host memory)
  std::vector<std::string> flat;    // per group: concatenated reference bases
  int dev_device = -1;
  SynthDev D;                       // device view (refs -> device memory)
  std::vector<void*> dev_allocs;
};

extern "C" {

bc_synth* bc_synth_create(const bc_plan* p, const bc_synth_params* prm) {
  if (!p || !prm) {
    set_error("bc_synth_create: null argument");
    return nullptr;
  }
  if (!p->unsupported.empty() || p->length > (uint32_t)kSynthMaxL || p->groups.size() > (size_t)kMaxGroups) {
    set_error("bc_synth_create: scheme not supported by the generator");
    return nullptr;
  }
  if (prm->read_len <= p->length || prm->phred_hi < prm->phred_lo || prm->lowq_hi < prm->lowq_lo) {
    set_error("bc_synth_create: read_len must exceed the format length; Phred ranges must be ordered");
    return nullptr;
  }
  bc_synth* s = new bc_synth();
  SynthDev& S = s->S;
  memset(&S, 0, sizeof S);
  S.seed = prm->seed;
  S.read_len = prm->read_len;
  S.L = p->length;
  S.p_sub = prm->p_sub;
  S.p_n = prm->p_n;
  S.p_lowq = prm->p_lowq;
  S.phred_lo = prm->phred_lo;
  S.phred_hi = prm->phred_hi;
  S.lowq_lo = prm->lowq_lo;
  S.lowq_hi = prm->lowq_hi;
  S.n_molecules = prm->n_molecules;
  S.n_groups = (uint32_t)p->groups.size();
  s->flat.resize(S.n_groups);
  for (uint32_t g = 0; g < S.n_groups; ++g) {
    const auto& fg = p->groups[g];
    SynthGroup& G = S.groups[g];
    G.type = fg.type;
    G.off = fg.off;
    G.len = fg.len;
    const KnownSet* set = nullptr;
    if (fg.type == kGroupSample && p->samples.size()) set = &p->samples;
    if (fg.type == kGroupBarcode && p->counted_loaded) set = &p->counted[fg.number - 1];
    if (fg.type != kGroupRandom) S.n_sb++;
    if (set) {
      for (const auto& q : set->seqs) {
        if (q.size() != fg.len) {
          set_error("bc_synth_create: every known barcode must have the length of its group");
          delete s;
          return nullptr;
        }
        s->flat[g] += q;
      }
      G.n_refs = (uint32_t)set->size();
    }
  }
  for (uint32_t g = 0; g < S.n_groups; ++g) S.groups[g].refs = s->flat[g].data();
  for (uint32_t i = 0; i < p->pos.size(); ++i) {
    const auto& fp = p->pos[i];
    S.fmt[i] = fp.kind == kPosConst ? (uint8_t)fp.letter : (fp.kind == kPosFmtN ? (uint8_t)'n' : (uint8_t)(0x80 | fp.group));
  }
  return s;
}

void bc_synth_destroy(bc_synth* s) {
  if (!s) return;
  if (s->dev_device >= 0) {
    (void)hipSetDevice(s->dev_device);
    for (void* p : s->dev_allocs) (void)hipFree(p);
  }
  delete s;
}

int bc_synth_generate_host(bc_synth* s, uint64_t first, uint64_t n, void* seq, void* qual, uint32_t stride) {
  if (stride < s->S.read_len) {
    set_error("bc_synth_generate: stride below read_len");
    return BC_ERR_INVALID;
  }
  uint8_t* so = (uint8_t*)seq;
  uint8_t* qo = (uint8_t*)qual;
  for (uint64_t t = 0; t < n; ++t) {
    SynthRead R;
    synth_begin(s->S, first + t, R);
    for (uint32_t p = 0; p < s->S.read_len; ++p) {
      uint8_t b, q;
      synth_byte(s->S, first + t, R, p, b, q);
      so[t * stride + p] = b;
      if (qo) qo[t * stride + p] = q;
    }
    for (uint32_t p = s->S.read_len; p < stride; ++p) {
      so[t * stride + p] = '\n';
      if (qo) qo[t * stride + p] = '\n';
    }
  }
  return BC_OK;
}

int bc_synth_generate_device(bc_synth* s, int device_id, void* hip_stream, uint64_t first, uint64_t n, void* d_seq,
                             void* d_qual, uint32_t stride) {
  if (stride < s->S.read_len) {
    set_error("bc_synth_generate: stride below read_len");
    return BC_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(device_id));
  if (s->dev_device != device_id) {
    for (void* p : s->dev_allocs) (void)hipFree(p);
    s->dev_allocs.clear();
    s->D = s->S;
    for (uint32_t g = 0; g < s->S.n_groups; ++g) {
      void* d = nullptr;
      HIP_TRY(hipMalloc(&d, s->flat[g].size() + 16));
      s->dev_allocs.push_back(d);
      if (!s->flat[g].empty()) HIP_TRY(hipMemcpy(d, s->flat[g].data(), s->flat[g].size(), hipMemcpyHostToDevice));
      s->D.groups[g].refs = (const char*)d;
    }
    s->dev_device = device_id;
  }
  if (n == 0) return BC_OK;
  const uint64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(synth_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)hip_stream, s->D, first, n,
                     (uint8_t*)d_seq, (uint8_t*)d_qual, stride);
  HIP_TRY(hipGetLastError());
  return BC_OK;
}

int bc_synth_make_set(uint64_t seed, uint32_t n, uint32_t k, uint32_t min_dist, char* out) {
  if (k == 0 || k > 32 || (k < 16 && (uint64_t)n > (1ull << (2 * k)))) {
    set_error("bc_synth_make_set: cannot draw that many distinct k-mers");
    return BC_ERR_INVALID;
  }
  std::vector<uint64_t> acc;  // 2 bits per base
  acc.reserve(n);
  const uint64_t mask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
  uint64_t ctr = 0;
  while (acc.size() < n) {
    if (ctr > 400ull * n + 100000) {
      set_error("bc_synth_make_set: could not place the requested k-mers at that minimum distance");
      return BC_ERR_INVALID;
    }
    const uint64_t c = synth_mix(seed ^ 0xB0C0DEull, ctr++, k) & mask;
    bool ok = true;
    for (uint64_t a : acc) {
      const uint64_t x = a ^ c;
      const uint64_t diff = (x | (x >> 1)) & 0x5555555555555555ull;
      if ((uint32_t)__builtin_popcountll(diff) < std::max(min_dist, 1u)) {
        ok = false;
        break;
      }
    }
    if (ok) acc.push_back(c);
  }
  const char* acgt = "ACGT";
  for (uint32_t i = 0; i < n; ++i) {
    for (uint32_t b = 0; b < k; ++b) out[(size_t)i * (k + 1) + b] = acgt[(acc[i] >> (2 * b)) & 3];
    out[(size_t)i * (k + 1) + k] = 0;
  }
  return BC_OK;
}

}  // extern "C"
