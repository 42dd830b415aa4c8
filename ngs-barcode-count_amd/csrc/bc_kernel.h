// bc_kernel.h -- device code of the hot kernel: SequenceParser::parse (parse.rs:53-76) for a batch.
// Included by bc_engine.hip (generic kernel: the plan is read from device memory) and by the
// scheme-specialised translation unit the engine compiles at run time with hiprtc (the plan is a
// compile-time constant, so every shift, offset and loop bound of the scheme becomes an immediate).
#pragma once
#include "bc_lane.h"

#ifdef BC_PROFILE
// wall-clock breakdown of a wave's tile loop: clock ticks between consecutive marks, summed over all waves
extern "C" __device__ unsigned long long bc_g_profile[16];
#endif

namespace bc {

#ifndef BC_TPB
#define BC_TPB 256
#endif
constexpr int kTPB = BC_TPB;       // reads (lanes) per workgroup
#ifndef BC_MIN_WAVES
#define BC_MIN_WAVES 3  // occupancy floor (waves per SIMD) the register allocator must honour
#endif

// ------------------------------------------------------------------------------------------------
// wave-level pieces
// ------------------------------------------------------------------------------------------------
// The lane's number, computed where it is asked for: the searches below run for a few reads in a hundred, and what
// they derive from a lane number that is known once and for all (segment, entry, block ...) is otherwise hoisted out of
// the loop over the tiles and kept in registers the hot path is short of.
__device__ __forceinline__ uint32_t lane_now() {
  uint32_t l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
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
  const uint32_t lane = lane_now();
  Nearest s;
  nearest_init(s);
  for (uint32_t j = lane; j < G.n_refs; j += 64) {
    bool ex;
    const uint32_t d = ref_distance(q1, q2, qn, qx, G.len, G.r1()[j], G.r2()[j], G.rn()[j], G.rlen()[j], ex);
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
// one pigeonhole index: nb blocks of blen bases from bit 0.  Returns true when the search is decided
// (the best distance found is <= the number of blocks finished, see below); s accumulates the candidates.
__device__ __forceinline__ bool wave_scan_blocks(uint32_t nb, uint32_t blen, const BC_GLOBAL uint32_t* off_base,
                                                 const BC_GLOBAL uint32_t* list_base, uint32_t n_idx, uint32_t q1, uint32_t q2,
                                                 Nearest& s) {
  const uint32_t bm = (1u << blen) - 1u;
  const uint32_t nbk = 1u << (2 * blen);
  const uint32_t lane = lane_now();
  const BC_GLOBAL uint4* entries = reinterpret_cast<const BC_GLOBAL uint4*>(list_base);
  for (uint32_t b = 0; b < nb; ++b) {
    const uint32_t val = ((q1 >> (b * blen)) & bm) | (((q2 >> (b * blen)) & bm) << blen);
    const BC_GLOBAL uint32_t* off = off_base + (size_t)b * (nbk + 1);
    const uint32_t beg = off[val], end = off[val + 1];
    const BC_GLOBAL uint4* list = entries + (size_t)b * n_idx;
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
        // a reference that equals the capture on an earlier block was scored there
        bool earlier = false;
        for (uint32_t p = 0; p < b; ++p) earlier = earlier || ((diff >> (p * blen)) & bm) == 0u;
        if (on[k] && !earlier) nearest_add(s, popc(diff), e[k].z, diff == 0u);
      }
    }
    // A reference with d mismatches has at most d spoiled blocks, so it has shown up by the time
    // blocks 0..d are done: once the best distance so far is <= b nothing nearer or equally near
    // can still be hiding in the later blocks.
    const uint32_t kmin = wave_min_u32(s.key);
    if (kmin != 0xFFFFFFFFu && kmin <= b + 1u) return true;
  }
  return false;
}

// The coarse index (at most four long blocks, short buckets) in two memory round trips instead of two per block:
// all bucket bounds first, then the first 64 entries of every bucket together; longer buckets finish in a loop.
// Complete for nb - 1 mismatches: decided iff the best distance found is below nb.
// kDepth: such round trips (64 entries of every bucket each) before the loops
template <int kMaxBlocks, int kDepth = 1>
__device__ __forceinline__ bool wave_scan_blocks_wide(uint32_t nb, uint32_t blen, const BC_GLOBAL uint32_t* off_base,
                                                      const BC_GLOBAL uint32_t* list_base, uint32_t n_idx, uint32_t q1,
                                                      uint32_t q2, Nearest& s) {
  const uint32_t bm = (1u << blen) - 1u;
  const uint32_t nbk = 1u << (2 * blen);
  const uint32_t lane = lane_now();
  const BC_GLOBAL uint4* entries = reinterpret_cast<const BC_GLOBAL uint4*>(list_base);
  uint32_t beg[kMaxBlocks], end[kMaxBlocks];
#pragma unroll
  for (int b = 0; b < kMaxBlocks; ++b) {
    beg[b] = end[b] = 0;
    if ((uint32_t)b < nb) {
      const uint32_t val = ((q1 >> (b * blen)) & bm) | (((q2 >> (b * blen)) & bm) << blen);
      const BC_GLOBAL uint32_t* off = off_base + (size_t)b * (nbk + 1);
      beg[b] = off[val];
      end[b] = off[val + 1];
    }
  }
  auto score = [&](const uint4& x, uint32_t b, bool on) {
    const uint32_t diff = (q1 ^ x.x) | (q2 ^ x.y);
    bool earlier = false;  // equal to the capture on an earlier block: scored there
    for (uint32_t p = 0; p < b; ++p) earlier = earlier || ((diff >> (p * blen)) & bm) == 0u;
    if (on && !earlier) nearest_add(s, popc(diff), x.z, diff == 0u);
  };
#pragma unroll
  for (int t = 0; t < kDepth; ++t) {  // (one after the other: kMaxBlocks loads in flight, not kMaxBlocks x kDepth registers)
    uint4 e[kMaxBlocks];
#pragma unroll
    for (int b = 0; b < kMaxBlocks; ++b) {
      const uint32_t i = beg[b] + 64u * t + lane;
      e[b] = make_uint4(0, 0, 0, 0);
      if ((uint32_t)b < nb) e[b] = entries[(size_t)b * n_idx + (i < end[b] ? i : (beg[b] < end[b] ? beg[b] : 0u))];
    }
#pragma unroll
    for (int b = 0; b < kMaxBlocks; ++b)
      if ((uint32_t)b < nb) score(e[b], (uint32_t)b, beg[b] + 64u * t + lane < end[b]);
  }
#pragma unroll
  for (int b = 0; b < kMaxBlocks; ++b) {
    if ((uint32_t)b >= nb) continue;
    for (uint32_t i = beg[b] + 64u * kDepth + lane; i - lane < end[b]; i += 64u) {  // wave-uniform trip count
      const bool on = i < end[b];
      const uint4 x = entries[(size_t)b * n_idx + (on ? i : beg[b])];
      score(x, (uint32_t)b, on);
    }
  }
  const uint32_t kmin = wave_min_u32(s.key);
  return kmin != 0xFFFFFFFFu && kmin <= nb;  // key = distance + 1
}

// coarse_done: the coarse index has been asked already (coarse_probe_x4) and could not decide
__device__ __forceinline__ void wave_seeded_min(const DevGroup& G, uint32_t q1, uint32_t q2, bool coarse_done, uint32_t& kmin_out,
                                                bool& unique_out, uint32_t& idx_out) {
  const uint32_t lane = lane_now();
  Nearest s;
  nearest_init(s);
  // the coarse index decides whenever some reference is within two mismatches (its three long blocks
  // make for short buckets); only otherwise is the full one, with its budget + 1 short blocks, walked
  bool decided = false;
  if (G.seed2_nb && !coarse_done) decided = wave_scan_blocks_wide<3>(G.seed2_nb, G.seed2_blen, G.seed2_off(), G.seed2_list(), G.n_idx, q1, q2, s);
#ifndef BC_NO_MID  // (A/B builds)
  if (!decided && G.seed3_nb && !(G.seed3_nb > 4u)) {
    // nothing within two mismatches: the middle index settles three (64 entries of each of its four buckets per round
    // trip; a bucket of the 100 k x 20-nt index holds a hundred)
    nearest_init(s);
    decided = wave_scan_blocks_wide<4, 2>(G.seed3_nb, G.seed3_blen, G.seed3_off(), G.seed3_list(), G.n_idx, q1, q2, s);
  }
#endif
  if (!decided) {
    nearest_init(s);  // whatever the earlier passes met is met again
    wave_scan_blocks(G.seed_nb, G.seed_blen, G.seed_off(), G.seed_list(), G.n_idx, q1, q2, s);
  }
  for (uint32_t i = lane; i < G.n_odd; i += 64) {
    const uint32_t j = G.odd_list()[i];
    bool ex;
    const uint32_t d = ref_distance(q1, q2, 0u, 0u, G.len, G.r1()[j], G.r2()[j], G.rn()[j], G.rlen()[j], ex);
    nearest_add(s, d, j, ex);
  }
  const uint32_t kmin = wave_min_u32(s.key);
  const unsigned long long holders = __ballot(s.key == kmin);
  const int first = __ffsll(holders) - 1;
  kmin_out = kmin;
  unique_out = __popcll(holders) == 1 && (uint32_t)__shfl((int)s.count, first) == 1u && kmin != 0xFFFFFFFFu;
  idx_out = (uint32_t)__shfl((int)s.idx, first);
}

// The coarse index for up to FOUR N-free captures at once, sixteen lanes each (lanes 16 s .. 16 s + 15 serve capture s):
// a wavefront has one to three captures waiting for this search far more often than none or many, and one after the
// other each of them costs two memory round trips with three quarters of the lanes idle.  (q1, q2) = the capture of the
// calling lane's segment; seg_on = that segment has one.  Per segment: the smallest key (distance + 1, 0 for the
// capture itself), whether exactly one reference has it, that reference -- valid when kmin <= nb (decided, as in
// wave_scan_blocks_wide); otherwise the full index has to be walked for that capture.
struct CoarseVote {
  uint32_t kmin;                      // the segment's smallest key
  unsigned long long holders, single; // lanes holding it / holding it with exactly one reference
  uint32_t idx_lane;                  // this lane's reference
};
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int lane_index) {  // lane_index: the same in every lane
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane_index);
}
// segment k of a vote: smallest key, whether exactly one reference has it, that reference (scalar work only)
__device__ __forceinline__ void coarse_segment(const CoarseVote& c, int k, uint32_t& kmin, bool& unique, uint32_t& idx) {
  kmin = rdlane(c.kmin, 16 * k);
  const uint32_t hm = (uint32_t)(c.holders >> (16 * k)) & 0xFFFFu, sm = (uint32_t)(c.single >> (16 * k)) & 0xFFFFu;
  const int first = 16 * k + (hm ? __ffs((int)hm) - 1 : 0);
  idx = rdlane(c.idx_lane, first);
  unique = __popc(hm) == 1 && sm == hm && kmin != 0xFFFFFFFFu;
}
__device__ __forceinline__ void coarse_probe_x4(const DevGroup& G, uint32_t q1, uint32_t q2, bool seg_on, CoarseVote& vote) {
  constexpr int kMaxBlocks = 3;
  const uint32_t nb = G.seed2_nb, blen = G.seed2_blen, n_idx = G.n_idx;
  const uint32_t bm = (1u << blen) - 1u;
  const uint32_t nbk = 1u << (2 * blen);
  const uint32_t lane = lane_now(), j = lane & 15u;
  const BC_GLOBAL uint4* entries = reinterpret_cast<const BC_GLOBAL uint4*>(G.seed2_list());
  Nearest s;
  nearest_init(s);
  uint32_t beg[kMaxBlocks], end[kMaxBlocks];
#pragma unroll
  for (int b = 0; b < kMaxBlocks; ++b) {
    beg[b] = end[b] = 0;
    if ((uint32_t)b < nb && seg_on) {
      const uint32_t val = ((q1 >> (b * blen)) & bm) | (((q2 >> (b * blen)) & bm) << blen);
      const BC_GLOBAL uint32_t* off = G.seed2_off() + (size_t)b * (nbk + 1);
      beg[b] = off[val];
      end[b] = off[val + 1];
    }
  }
  auto score = [&](const uint4& x, uint32_t b, bool on) {
    const uint32_t diff = (q1 ^ x.x) | (q2 ^ x.y);
    bool earlier = false;  // equal to the capture on an earlier block: scored there
    for (uint32_t p = 0; p < b; ++p) earlier = earlier || ((diff >> (p * blen)) & bm) == 0u;
    if (on && !earlier) nearest_add(s, popc(diff), x.z, diff == 0u);
  };
  // the first thirty-two entries of every bucket in one round trip (a bucket of the 100 k x 20-nt index holds two
  // dozen); longer buckets finish in a loop
  uint4 e[kMaxBlocks][2];
#pragma unroll
  for (int b = 0; b < kMaxBlocks; ++b) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const uint32_t i = beg[b] + j + 16u * t;
      e[b][t] = make_uint4(0, 0, 0, 0);
      if (i < end[b]) e[b][t] = entries[(size_t)b * n_idx + i];
    }
  }
#pragma unroll
  for (int b = 0; b < kMaxBlocks; ++b) {
    score(e[b][0], (uint32_t)b, beg[b] + j < end[b]);
    score(e[b][1], (uint32_t)b, beg[b] + j + 16u < end[b]);
    for (uint32_t i = beg[b] + 32u + j; __any(i - j < end[b]); i += 16u) {  // (wave-uniform trip count)
      const bool on = i < end[b];
      uint4 x = make_uint4(0, 0, 0, 0);
      if (on) x = entries[(size_t)b * n_idx + i];
      score(x, (uint32_t)b, on);
    }
  }
  // minimum over the segment's sixteen lanes: rotations within the row (DPP), no trip through the LDS crossbar
  uint32_t kmin = s.key;
  {
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)kmin, 0x128 /* row_ror:8 */, 0xF, 0xF, false);
    kmin = t < kmin ? t : kmin;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)kmin, 0x124 /* row_ror:4 */, 0xF, 0xF, false);
    kmin = t < kmin ? t : kmin;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)kmin, 0x122 /* row_ror:2 */, 0xF, 0xF, false);
    kmin = t < kmin ? t : kmin;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)kmin, 0x121 /* row_ror:1 */, 0xF, 0xF, false);
    kmin = t < kmin ? t : kmin;
  }
  vote.kmin = kmin;
  vote.holders = __ballot(s.key == kmin);
  vote.single = __ballot(s.key == kmin && s.count == 1u);
  vote.idx_lane = s.idx;
}

// A capture with one to four 'N's (bn; b1, b2, bn the same in every lane) through the coarse index: its 4 .. 256
// substitutions (wave_fix_error_seeded) are four plain captures per pass of coarse_probe_x4.  A substitution the probe
// decides (key <= blocks) has its true minimum and count; one it cannot decide has nothing that near, so any decided
// substitution settles the capture: true, and res = the verdict.
__device__ __forceinline__ bool coarse_with_n(const DevGroup& G, uint32_t b1, uint32_t b2, uint32_t bn, uint32_t& res_out) {
  const uint32_t lane = lane_now();
  const uint32_t n_n = (uint32_t)__popc(bn);
  const uint32_t combos = 1u << (2u * n_n);
  uint32_t best = 0xFFFFFFFFu, res = kFail;
  bool ok = false, any_decided = false;
  for (uint32_t c0 = 0; c0 < combos; c0 += 4u) {
    const uint32_t c = c0 + (lane >> 4);
    uint32_t s1 = b1 & ~bn, s2 = b2 & ~bn;
    for (uint32_t t = 0, rem = bn; rem; ++t, rem &= rem - 1u) {  // base (c >> 2t) & 3 at the t-th 'N'
      s1 |= ((c >> (2u * t)) & 1u) << ctz(rem);
      s2 |= ((c >> (2u * t + 1u)) & 1u) << ctz(rem);
    }
    CoarseVote vote;
    coarse_probe_x4(G, s1, s2, true, vote);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t kk, ii;
      bool uu;
      coarse_segment(vote, k, kk, uu, ii);
      if (kk <= G.seed2_nb) {
        any_decided = true;
        if (kk < best) {
          best = kk;
          ok = uu;
          res = ii;
        } else if (kk == best) {
          ok = false;
        }
      }
    }
  }
  res_out = (ok && (best == 0u || best - 1u <= G.max_err)) ? res : kFail;
  return any_decided;
}

// A capture with up to two 'N's against plain references: 'N' is free (parse.rs:569), so its
// distance to a reference is that of the capture with each N replaced by the reference's base
// there.  The nearest references of the capture are therefore those of its 4 (16) substitutions
// at the smallest of their minimum distances, and the match is unique iff exactly one substitution
// reaches that distance and does so uniquely (the argument of single_n_lookup, bc_lane.h).
__device__ __forceinline__ uint32_t wave_fix_error_seeded(const DevGroup& G, uint32_t q1, uint32_t q2, uint32_t qn,
                                                          bool coarse_done = false) {
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
    wave_seeded_min(G, s1, s2, coarse_done, k, uniq, idx);
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

// the counter-table increment (no return value).  BC_ATOMIC_CPOL: cache-policy experiment (1 = nt, 2 = sc1)
#ifndef BC_ATOMIC_CPOL
#define BC_ATOMIC_CPOL 0
#endif
__device__ __forceinline__ void table_add(uint32_t* p) {
#if BC_ATOMIC_CPOL == 1
  asm volatile("global_atomic_add %0, %1, off nt" ::"v"(p), "v"(1u) : "memory");
#elif BC_ATOMIC_CPOL == 2
  asm volatile("global_atomic_add %0, %1, off sc1" ::"v"(p), "v"(1u) : "memory");
#else
  atomicAdd(p, 1u);
#endif
}

// ---- hot-counter cache -------------------------------------------------------------------------------------
// A library with a few very abundant members (a CRISPR screen's top guides, an enriched compound) sends a large share
// of the matched reads to a handful of counters, and memory-side atomics on ONE address run one after the other
// (~60 M/s: a Zipf-like guide distribution ran 13 x slower than a uniform one).  Every workgroup therefore keeps
// kHotSlots counters in LDS, claimed first come first served by the table indices it meets (tag = index, never
// evicted): an abundant index is all but certain to claim its slot within the workgroup's first few hundred reads, and
// from then on costs an LDS add; everything else goes to the table as before.  An index has two slots to try (two
// hashes seeded by the workgroup), so two abundant indices that collide in one workgroup do not in the others.  The cached counts are
// added to the table once, when the workgroup ends.  Plans whose table index fits 32 bits.
#ifndef BC_HOT_BITS
#define BC_HOT_BITS 8
#endif
constexpr uint32_t kHotBits = BC_HOT_BITS, kHotSlots = 1u << kHotBits, kHotEmpty = 0xFFFFFFFFu;
constexpr uint32_t kHotBytes = kHotSlots * 8u;
// The search queue of a wavefront (plans with DevPlan::defer_search): captures that need the search beyond one mismatch
// wait here -- {plane 1, plane 2, 'N' mask, tile and lane of the read} -- until four plain ones share one pass of the
// coarse index; those with 'N's (filed from the top end) are judged one by one at the end of the tile's iteration.
// Small on purpose (1 KiB per workgroup: a fifth workgroup still fits a CU next to 100-byte tiles, LDS being handed
// out in 1,280-byte pieces); a tile with more such captures than there is room for empties the queue in between.
constexpr uint32_t kQueueEntries = 16u, kQueueBytes = kQueueEntries * 16u;

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

// a table add that also flags the entry's 256-byte block for the next reset (DevPlan::dirty_off; a plain byte store
// of 1: idempotent, so no atomic is needed)
__device__ __forceinline__ void table_add_marked(const DevPlan& pl, uint32_t* __restrict__ table, uint32_t* __restrict__ bits,
                                                 uint64_t idx) {
  table_add(&table[idx]);
  if (bits && pl.dirty_off) reinterpret_cast<uint8_t*>(bits + pl.dirty_off)[idx >> 6] = (uint8_t)1;
}

// cache policy of the streaming tile fetch: 2 = nt (non-temporal).  Every byte of a batch is read exactly once, so
// it should not push the correction tables out of L2 or claim Infinity Cache lines; measured 2.3 % on config 3.
#ifndef BC_DMA_CPOL
#define BC_DMA_CPOL 2
#endif
// Asynchronous tile fetch: global -> LDS without passing through registers
// (global_load_lds_dwordx4: every lane moves 16 bytes, the wave 1 KiB per instruction, destination
// = wave-uniform LDS base + lane * 16).  `bytes` must be a multiple of 16 (a full 64-read tile is).
__device__ __forceinline__ void dma_tile(uint8_t* lds_dst, const uint8_t* __restrict__ src, uint32_t bytes, uint32_t lane) {
  for (uint32_t off0 = 0; off0 < bytes; off0 += 1024u) {
    const uint32_t off = off0 + lane * 16u;
    if (off < bytes)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                       (__attribute__((address_space(3))) void*)(lds_dst + off0), 16, 0, BC_DMA_CPOL);
  }
}

// waits until at most `keep` of this wave's youngest vector-memory operations are outstanding
// (they complete in issue order, so everything older has landed)
__device__ __forceinline__ void wait_vm_keep(uint32_t keep) {
  switch (keep) {  // the count is an immediate of the instruction
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
    case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
    case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
    case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
    case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break;
    case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
    case 31: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
    case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    case 33: asm volatile("s_waitcnt vmcnt(33)" ::: "memory"); break;
    case 34: asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); break;
    case 35: asm volatile("s_waitcnt vmcnt(35)" ::: "memory"); break;
    case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
    case 37: asm volatile("s_waitcnt vmcnt(37)" ::: "memory"); break;
    case 38: asm volatile("s_waitcnt vmcnt(38)" ::: "memory"); break;
    case 39: asm volatile("s_waitcnt vmcnt(39)" ::: "memory"); break;
    case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  // 0, or more than the ladder knows: wait for all
  }
}

struct DeviceOps {
#ifdef BC_PROFILE
  unsigned long long t_last, acc[12];
  __device__ __forceinline__ void mark(int k) {
    __builtin_amdgcn_sched_barrier(0);  // keep each phase's instructions on its side of the mark
    const unsigned long long now = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
    acc[k] += now - t_last;
    t_last = now;
  }
#else
  __device__ __forceinline__ void mark(int) const {}
#endif
  const Quad* area;       // the workgroup's LDS exact-match tables
  bool with_tables;       // ... are loaded for this launch
  __device__ __forceinline__ const Quad* lhash() const { return area; }
#ifdef JIT_LHASH
  __device__ __forceinline__ bool tables() const { return JIT_LHASH != 0; }  // fixed when the kernel was specialised
#else
  __device__ __forceinline__ bool tables() const { return with_tables; }
#endif
  uint8_t* tile;          // this wave's LDS region for sequence lines
  uint8_t* qtile;         // ... for quality lines (pipelined fetch) -- equals `tile` when fetched on demand
  const uint8_t* qsrc;    // this tile's quality lines in global memory
  const uint8_t* next_seq;  // next full tile of this wave (nullptr: none, or a partial one fetched synchronously)
  uint32_t bytes, region, lane;
  uint32_t chunks;        // 1-KiB fetch instructions per full tile
  uint32_t pending_add;   // 1 while the previous tile's counter atomic may still be in flight
  bool qual_async;        // this tile's quality lines were requested ahead of time

  __device__ __forceinline__ bool any(bool c) const { return __any(c) != 0; }
  uint32_t abl_;  // experiment switches (0 outside BC_EXPERIMENT builds)
  bool defer_;            // captures that need the seed indexes are handed back (kDeferred), not searched at once
  __device__ __forceinline__ void issued() const { asm volatile("" ::: "memory"); }
  // the sequence bytes are dead once the planes are built: the next tile's sequence lines can land
  __device__ __forceinline__ void sequence_consumed() const {
    wave_lds_fence();
    if (next_seq) dma_tile(tile, next_seq, 64u * (bytes_per_read()), lane);
  }
  __device__ __forceinline__ uint32_t bytes_per_read() const { return stride_; }
  uint32_t stride_;
  // younger: loads the lane code itself has in flight (issued after everything below)
  __device__ __forceinline__ const uint32_t* stage_quality(uint32_t younger) const {
    if (qual_async) {
      // requested before the previous tile's counter atomic and the next tile's sequence lines: those
      // (and the caller's own loads) may stay in flight
      wait_vm_keep((next_seq ? chunks : 0u) + pending_add + younger);
      wave_lds_fence();
      return reinterpret_cast<const uint32_t*>(qtile);
    }
    wave_lds_fence();
    stage_tile(qtile, qsrc, bytes, region, (uint8_t)'I', lane);
    wave_lds_fence();
    return reinterpret_cast<const uint32_t*>(qtile);
  }
  // tier_lookup_single_n (bc_lane.h) for the lanes that `want` it, two captures per pass: the 4 substitutions x 2
  // blocks x 4 bucket-head entries of one capture are 32 lanes' worth of one load, so a capture costs one round trip
  // of half the wave instead of four of all of it.  Every lane calls this.
  __device__ __forceinline__ uint32_t tier_single_n(const DevGroup& G, uint32_t q1, uint32_t q2, uint32_t qn, bool want,
                                                    bool& settled) const {
    const uint32_t lane = lane_now();  // (not the member: see lane_now)
    uint32_t out = kFail;
    settled = false;
    unsigned long long todo = __ballot(want);
    const uint32_t blen = G.tier_blen, bm = (1u << blen) - 1u, nbk = 1u << (2u * blen);
    const BC_GLOBAL uint32_t* bkt = G.tier_bkt();
    const uint32_t half = lane >> 5, sub = (lane >> 3) & 3u, blk = (lane >> 2) & 1u, ent = lane & 3u;
    while (todo) {
      const int s0 = __ffsll(todo) - 1;
      todo &= todo - 1;
      int s1 = s0;
      const bool two = todo != 0;
      if (two) {
        s1 = __ffsll(todo) - 1;
        todo &= todo - 1;
      }
      const bool on = half == 0u || two;
      // (lane indices that are the same for the whole wave: v_readlane instead of a trip through the LDS crossbar)
      const uint32_t b1 = half ? rdlane(q1, s1) : rdlane(q1, s0), b2 = half ? rdlane(q2, s1) : rdlane(q2, s0);
      const uint32_t bn = half ? rdlane(qn, s1) : rdlane(qn, s0);
      const uint32_t k = bn ? ctz(bn) : 0u;
      const uint32_t c1 = (b1 & ~bn) | ((sub & 1u) << k), c2 = (b2 & ~bn) | ((sub >> 1) << k);
      const uint32_t sh = blk * G.tier_stride;
      const uint32_t val = ((c1 >> sh) & bm) | (((c2 >> sh) & bm) << blen);
      uint32_t r1, r2, j, n_here;
      if (G.tier_compact) {
        const BC_GLOBAL uint32_t* line = bkt + ((size_t)blk * nbk + val) * 8u;
        const uint2 wv = *reinterpret_cast<const BC_GLOBAL uint2*>(line + 2u * ent);  // one load, not one per word
        asm volatile("" ::: "memory");  // (or the compiler fetches the second word in a round trip of its own, where it is used)
        const uint32_t w0 = wv.x, w1 = wv.y;
        const uint32_t len = G.len, lm = lowmask(len);
        const uint64_t x = ((uint64_t)w1 << 32) | w0;
        r1 = (uint32_t)x & lm;
        r2 = (uint32_t)(x >> len) & lm;
        j = (uint32_t)((x & 0x1FFFFFFFFFFFFFFFull) >> (2u * len));
        n_here = w1 >> 29;
      } else {
        const uint4 e = *reinterpret_cast<const BC_GLOBAL uint4*>(bkt + ((size_t)blk * nbk + val) * 16u + 4u * ent);
        asm volatile("" ::: "memory");
        r1 = e.x;
        r2 = e.y;
        j = e.z;
        n_here = e.w;
      }
      const uint32_t n = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)n_here, 0 /* quad_perm:[0,0,0,0] */, 0xF, 0xF, false);
      const uint32_t diff = (c1 ^ r1) | (c2 ^ r2);
      // a reference that equals the substitution on block 0 too was met there
      const bool use = on && ent < n && !(blk == 1u && (diff & bm) == 0u);
      uint32_t d = use ? (uint32_t)__popc(diff) : 0xFFFFFFFFu;
      uint32_t held = 1;  // references this lane has met at distance d
      if (__any(on && n > 4u)) {
        // the rest of a long bucket: its four lanes walk the list, four entries per round trip
        uint32_t i = 0, end = 0;
        if (on && n > 4u) {
          const BC_GLOBAL uint32_t* off = G.tier_off() + (size_t)blk * (nbk + 1u);
          i = off[val] + 4u + ent;
          end = off[val + 1u];
        }
        const BC_GLOBAL uint32_t* list = G.tier_list() + (size_t)blk * G.n_idx * 4u;
        while (__any(i < end)) {
          if (i < end) {
            const uint4 le = *reinterpret_cast<const BC_GLOBAL uint4*>(list + (size_t)i * 4u);
            asm volatile("" ::: "memory");
            const uint32_t l1 = le.x, l2 = le.y, lj = le.z;
            const uint32_t df = (c1 ^ l1) | (c2 ^ l2);
            if (!(blk == 1u && (df & bm) == 0u)) {
              const uint32_t dd = (uint32_t)__popc(df);
              if (dd < d) {
                d = dd;
                j = lj;
                held = 1;
              } else if (dd == d) {
                ++held;
              }
            }
          }
          i += 4u;
        }
      }
      // only distances 0 and 1 settle a capture: two votes instead of a minimum over the lanes
      const unsigned long long z0 = __ballot(d == 0u), z1 = __ballot(d == 1u);
      uint32_t res_h[2];
      bool sett_h[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t zh0 = (uint32_t)(z0 >> (32 * h)), zh1 = (uint32_t)(z1 >> (32 * h));
        const uint32_t m = zh0 ? 0u : (zh1 ? 1u : 2u);
        const uint32_t hb = zh0 ? zh0 : zh1;
        const int first = hb ? (__ffs((int)hb) - 1 + 32 * h) : 0;
        const uint32_t idx = rdlane(j, first);
        const bool one = __popc(hb) == 1 && rdlane(held, first) == 1u;
        sett_h[h] = m <= 1u;
        res_h[h] = (sett_h[h] && one && m <= G.max_err) ? idx : kFail;
      }
      const uint32_t res0 = res_h[0], res1 = res_h[1];
      const bool sett0 = sett_h[0], sett1 = sett_h[1];
      if ((int)lane == s0) {
        out = res0;
        settled = sett0;
      }
      if (two && (int)lane == s1) {
        out = res1;
        settled = sett1;
      }
    }
    return out;
  }
  // every lane calls this; lanes with `need` get their capture resolved one after the other
  __device__ __forceinline__ uint32_t nearest(const DevGroup& G, uint32_t q1, uint32_t q2, uint32_t qn, uint32_t qx,
                                              bool need) {
    const uint32_t lane = lane_now();  // (not the member: see lane_now)
    uint32_t out = kFail;
    unsigned long long todo = __ballot(need);
    unsigned long long probed = 0;  // captures the coarse index has been asked about
    if (defer_) {
      // captures the seed indexes answer (no foreign byte, at most two 'N's) are not searched here: the read goes back
      // as kPending with its capture, and the kernel's tile loop queues and judges it (match_count_body).  (A read that
      // gets here is anchored and of good quality, and a plan with a queue has this one group: the read's verdict is
      // this capture's.)  What is left -- foreign bytes, three 'N's and more -- is settled below.
      const bool later = need && qx == 0u && __popc(qn) <= 2;
      if (later) out = kDeferred;
      todo &= ~__ballot(later);
    }
    if (!defer_ && G.seed_nb && G.seed2_nb && G.seed2_nb <= 3u && G.n_odd == 0u) {
      // plain captures (no 'N', no foreign byte) four at a time through the coarse index; what it cannot decide (the
      // nearest reference is three or more mismatches away, or there is none) stays in `todo` for the full search
      unsigned long long plain = __ballot(need && qn == 0u && qx == 0u);
      probed |= plain;
      while (plain) {
        int src[4];
        int n_seg = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          src[k] = plain ? __ffsll(plain) - 1 : src[0];
          if (plain) {
            plain &= plain - 1;
            ++n_seg;
          }
        }
        const uint32_t seg = lane >> 4;
        const bool seg_on = (int)seg < n_seg;
        // (lane indices that are the same for the whole wave: v_readlane, not the LDS crossbar)
        uint32_t b1 = rdlane(q1, src[0]), b2 = rdlane(q2, src[0]);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
          const uint32_t t1 = rdlane(q1, src[k]), t2 = rdlane(q2, src[k]);
          b1 = seg == (uint32_t)k ? t1 : b1;
          b2 = seg == (uint32_t)k ? t2 : b2;
        }
        CoarseVote vote;
        coarse_probe_x4(G, b1, b2, seg_on, vote);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (k >= n_seg) break;
          uint32_t kmin, idx;
          bool uniq;
          coarse_segment(vote, k, kmin, uniq, idx);
          if (kmin != 0xFFFFFFFFu && kmin <= G.seed2_nb) {  // decided
            if ((int)lane == src[k]) out = (uniq && (kmin == 0u || kmin - 1u <= G.max_err)) ? idx : kFail;
            todo &= ~(1ull << src[k]);
          }
        }
      }
    }
    if (G.seed_nb && G.seed2_nb && G.seed2_nb <= 3u && G.n_odd == 0u) {
      // captures with one to four 'N's: their 4 .. 256 substitutions (wave_fix_error_seeded) are four plain captures per
      // pass of the same coarse probe -- a capture with three 'N's, one in a million, would otherwise hold its
      // wavefront for milliseconds in the scan of the whole set.  A substitution the probe decides (key <= blocks) has
      // its true minimum and count; one it cannot decide has nothing that near, so any decided substitution settles
      // the capture.
      unsigned long long with_n = __ballot(need && qx == 0u && qn != 0u && __popc(qn) <= 4) & todo;  // (& todo: not the queued ones)
      probed |= with_n;
      while (with_n) {
        const int src = __ffsll(with_n) - 1;
        with_n &= with_n - 1;
        const uint32_t b1 = rdlane(q1, src), b2 = rdlane(q2, src), bn = rdlane(qn, src);
        uint32_t res_n;
        const bool any_decided = coarse_with_n(G, b1, b2, bn, res_n);
        if (any_decided) {
          if ((int)lane == src) out = res_n;
          todo &= ~(1ull << src);
        }
      }
    }
    if (abl_ & 0x100000u) todo &= ~probed;  // experiment: what the coarse index cannot decide simply fails
    while (todo) {
      const int src = __ffsll(todo) - 1;
      todo &= todo - 1;
      const uint32_t b1 = (uint32_t)__shfl((int)q1, src);
      const uint32_t b2 = (uint32_t)__shfl((int)q2, src);
      const uint32_t bn = (uint32_t)__shfl((int)qn, src);
      const uint32_t bx = (uint32_t)__shfl((int)qx, src);
      // the seed index answers captures with at most two 'N's when every reference is plain
      const bool seeded = !defer_ && G.seed_nb && bx == 0u && G.n_odd == 0u && __popc(bn) <= 2;  // (queued otherwise)
      const uint32_t r = seeded ? wave_fix_error_seeded(G, b1, b2, bn, ((probed >> src) & 1ull) != 0ull)
                                : wave_fix_error(G, b1, b2, bn, bx, true);
      if (lane == (uint32_t)src) out = r;
    }
    return out;
  }
};

// ------------------------------------------------------------------------------------------------
// the hot kernel: SequenceParser::parse, 64 reads per wavefront, 4 wavefronts per workgroup
// ------------------------------------------------------------------------------------------------
// kLens: per-read lengths are given.  A separate instantiation because the length load would
// otherwise put a full vector-memory wait into every iteration, fetches in flight or not.
// kAligned: the stride is a multiple of 4 (known only to a kernel specialised for its batch shape)
template <int NW, int NWW, bool kLens, bool kAligned = false>
__device__ __forceinline__ void match_count_body(const DevPlan& pl, const uint8_t* __restrict__ seq,
                                                 const uint8_t* __restrict__ qual, const uint16_t* __restrict__ lens,
                                                 const uint16_t* __restrict__ qlens,
                                                 uint32_t stride, uint32_t read_len, uint32_t nd, uint64_t n_reads,
                                                 uint32_t region, uint32_t* __restrict__ table, uint32_t* __restrict__ bits,
                                                 unsigned long long* __restrict__ slots, uint32_t* __restrict__ vals,
                                                 uint64_t smask, unsigned long long* __restrict__ counters,
                                                 uint8_t* __restrict__ trace_outcome, uint64_t* __restrict__ trace_idx,
                                                 uint32_t flags) {
  extern __shared__ uint4 smem[];
  __shared__ uint32_t s_cnt[kNCounters];
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  // Plans with a search queue (DevPlan::defer_search) get the wave's number as a scalar: the tile numbers and offsets
  // that derive from it then live in scalar registers, which is what lets their kernel fit five waves per SIMD (93
  // vector registers; 98 and a spill without).  Elsewhere the scalar registers are the scarcer ones: config 2 ran 5 %
  // slower with it.
  const bool queues = pl.defer_search();
  const uint32_t wave = queues ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)) : tid >> 6;
  if (tid < kNCounters) s_cnt[tid] = 0;

  const bool with_qual = pl.quality_on && !(pl.abl() & 0x8u);
  // flags bit 0: software-pipelined tile fetch (two LDS regions per wave with the quality filter on);
  // without it a wave has one region, filled on demand
  // flags bit 1: the plan's exact-match tables are copied into LDS and used
  const bool pipe = (flags & 1u) != 0u;
#ifdef JIT_LHASH
  const uint32_t lhash_vec = JIT_LHASH ? pl.lhash_vec : 0u;
#else
  const uint32_t lhash_vec = (flags & 2u) ? pl.lhash_vec : 0u;
#endif
  DeviceOps ops;
  // LDS: [exact-match tables of the workgroup][per wave: sequence region (+ quality region)]
  {
    Quad* area = reinterpret_cast<Quad*>(smem);
    const BC_GLOBAL Quad* image = reinterpret_cast<const BC_GLOBAL Quad*>(pl.lhash_image());
    for (uint32_t i = tid; i < lhash_vec; i += kTPB) area[i] = image[i];
    ops.area = area;
    ops.with_tables = lhash_vec != 0u;
    ops.abl_ = pl.abl();
  }
  // flags bit 2: the workgroup's hot-counter cache sits between the tables and the tiles
  const bool hot = (flags & 4u) != 0u;
  const uint32_t hot_seed = blockIdx.x * 0x85EBCA6Bu;
  uint32_t* hot_tag = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(smem) + lhash_vec * 16u);
  uint32_t* hot_cnt = hot_tag + kHotSlots;
  if (hot) {
    for (uint32_t i = tid; i < kHotSlots; i += kTPB) {
      hot_tag[i] = kHotEmpty;
      hot_cnt[i] = 0u;
    }
  }
  const uint32_t two = (with_qual && pipe) ? 2u : 1u;
  // the waves' search queues sit between the cache and the tiles (the host reserves them whenever the plan has them:
  // lds_tiles_for)
  ops.defer_ = queues && !(pl.abl() & 0x200000u);
  uint4* const dq = reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(smem) + lhash_vec * 16u + (hot ? kHotBytes : 0u) + wave * kQueueBytes);
  uint32_t dq_plain = 0, dq_n = 0;  // entries waiting: plain ones in dq[0 ..), those with 'N' in dq[kQueueEntries - 1 ..) downwards
  uint32_t tile_seq = 0;            // tiles this wave has been through
  ops.tile = reinterpret_cast<uint8_t*>(smem) + lhash_vec * 16u + (hot ? kHotBytes : 0u) + (queues ? (kTPB / 64) * kQueueBytes : 0u) +
             wave * region * two;
  ops.qtile = ops.tile + (two == 2u ? region : 0u);
  ops.region = region;
  ops.lane = lane;
  ops.stride_ = stride;
  const uint32_t full_bytes = 64u * stride;
  ops.chunks = (full_bytes + 1023u) / 1024u;
  ops.pending_add = 0;

  // Persistent wavefronts: wave-tile t = 64 consecutive reads; this wave takes tiles
  // gid, gid + G, gid + 2G, ...  No workgroup barrier inside the loop; the outcome counters stay in
  // registers until the very end, so the six global counters see one add per workgroup.
  // Software pipeline: while tile t is being matched, the sequence lines of the wave's next tile
  // are already on their way into the sequence region (free once t's bit planes exist), and t's
  // quality lines were requested at the end of the previous iteration.
  const uint64_t n_tiles = (n_reads + 63u) >> 6;
  const uint64_t n_full = n_reads >> 6;  // tiles with all 64 reads
  const uint64_t n_waves = (uint64_t)gridDim.x * (kTPB / 64);
  uint32_t acc_cnt[kNCounters];
#pragma unroll
  for (int k = 0; k < kNCounters; ++k) acc_cnt[k] = 0;
  // slack bytes behind the reads that the lane code may touch: plain values, written once
  for (uint32_t i = full_bytes + lane; i < region; i += 64) {
    ops.tile[i] = 'A';
    if (with_qual) ops.qtile[i] = 'I';
  }
  // Two-level counting (`bits` given: large dense tables): the count of a tuple is bit + table entry.  A matched read
  // first tries to set its tuple's bit in a one-bit-per-tuple map -- 32 times denser than the table, so a good part of
  // it stays in the memory-side cache and a first occurrence costs no random 64-byte read and write-back of a table
  // line in DRAM; only a read that finds its bit set already adds to the table.  The atomic OR has to return the old
  // word: it is issued at the end of a tile and looked at one tile later (first_old / first_idx / first_on).
  // Whichever way a read is counted the total stays exact (bit 0 -> 1, or table + 1), so the choice is free: a wave whose
  // probes mostly find their bit set already (a batch counted twice, a library seen many times over) goes straight to the
  // table for the next fifteen tiles and probes again on the sixteenth.
  uint32_t first_old = 0, first_on = 0, direct_tiles = 0;
  uint64_t first_idx = 0;
  uint64_t t = (uint64_t)blockIdx.x * (kTPB / 64) + wave;
  bool seq_ready = false, qual_ready = false;  // requested ahead of time
  if (t < n_full && pipe) {
    dma_tile(ops.tile, seq + (t << 6) * stride, full_bytes, lane);
    seq_ready = true;
    if (with_qual) {
      dma_tile(ops.qtile, qual + (t << 6) * stride, full_bytes, lane);
      qual_ready = true;
    }
  }
  __syncthreads();  // exact-match tables in place, s_cnt zeroed
#ifdef BC_PROFILE
  for (int k = 0; k < 12; ++k) ops.acc[k] = 0;
  ops.t_last = __builtin_readcyclecounter();
#endif
  // the workgroup's LDS counters: an index that owns (or can still claim) its slot is counted there; what is left
  // for the table comes back
  auto hot_count = [&](bool add, uint32_t key) -> bool {
    uint32_t h = ((key ^ hot_seed) * 0x9E3779B1u) >> (32u - kHotBits);
    uint32_t tag = add ? hot_tag[h] : key ^ 1u;
#ifndef BC_HOT_ONE
    // taken by another index: a second slot to try.  With one slot per index the guides of rank ~20-250 of a
    // Zipf-like library owned theirs in some workgroups only (whoever comes first keeps a slot), and what was
    // left of their adds -- one address, ~60 M/s -- decided the kernel's time: 6.84 -> 5.75 ms per 125 M reads
    if (add && tag != key && tag != kHotEmpty) {
      h = ((key ^ hot_seed) * 0xC2B2AE35u) >> (32u - kHotBits);
      tag = hot_tag[h];
    }
#endif
    if (add && tag == kHotEmpty) {
      const uint32_t old = atomicCAS(&hot_tag[h], kHotEmpty, key);
      tag = old == kHotEmpty ? key : old;
    }
    if (add && tag == key) {
      atomicAdd(&hot_cnt[h], 1u);
      add = false;
    }
    return add;
  };
  for (; t < n_tiles; t += n_waves) {
    const uint64_t wfirst = t << 6;
    const uint32_t n_w = n_reads - wfirst < 64u ? (uint32_t)(n_reads - wfirst) : 64u;
    const uint64_t goff = wfirst * stride;  // multiple of 64 bytes: 16-byte aligned
    const uint64_t tn = t + n_waves;
    const bool next_full = tn < n_full && pipe;
    ops.qsrc = qual + goff;
    ops.bytes = n_w * stride;
    ops.next_seq = next_full ? seq + (tn << 6) * stride : nullptr;
    ops.qual_async = qual_ready;
    if (seq_ready) {
      // outstanding, oldest first: [this tile's sequence lines] [its quality lines] [the previous tile's
      // counter atomic] -- only the first must have landed
      wait_vm_keep((qual_ready ? ops.chunks : 0u) + ops.pending_add);
      wave_lds_fence();
    } else {
      wave_lds_fence();  // the previous tile's last LDS reads are done before the tile is overwritten
      stage_tile(ops.tile, seq + goff, ops.bytes, full_bytes, (uint8_t)'A', lane);
      wave_lds_fence();
    }

    ops.mark(1);
    const bool active = lane < n_w;
    uint32_t len = 0;
    if (active) len = kLens ? (uint32_t)lens[wfirst + lane] : read_len;
    // the quality line's own length, when it differs from the sequence line's (zip truncation, parse.rs:340-345)
    uint32_t qlen = len;
    if (kLens && qlens && active) qlen = (uint32_t)qlens[wfirst + lane];
    const uint32_t base = (active ? lane : 0u) * stride;
    const ReadResult r =
        process_read<DeviceOps, NW, NWW, kAligned>(pl, ops, reinterpret_cast<const uint32_t*>(ops.tile), base, len, qlen, nd, active);

    ops.mark(8);
    uint32_t outcome = r.outcome;
    // (outcome kPending: the read's one barcode needs the seed indexes -- ops.nearest.  It is queued at the end of this
    // iteration and judged and counted there, or at the end of a later one when four are together, at the latest with
    // the wave's last tile.  Until then the read is in no outcome counter; it is in TotalReads below.)
    if (pl.has_random) {
      // Results::add_count with a random barcode (info.rs:770-802): insert (tuple, random) into the
      // set; an element already present makes the read a duplicate (parse.rs:65-69)
      if (active && outcome == kMatched && !set_insert(slots, smask, r.dense_idx * pl.rspace + r.rcode))
        outcome = kDuplicate;
    }
    // outcome counters (SequenceErrors, info.rs:16-139)
#pragma unroll
    for (uint32_t k = 0; k < kNCounters; ++k) {
      if (k == kTotalReads) continue;
      acc_cnt[k] += (uint32_t)__popcll(__ballot(active && outcome == k));
    }
    acc_cnt[kTotalReads] += n_w;
    if (trace_outcome && active && !(queues && outcome == kPending)) {
      trace_outcome[wfirst + lane] = (uint8_t)outcome;
      trace_idx[wfirst + lane] = pl.has_random ? r.dense_idx * pl.rspace + r.rcode : r.dense_idx;
    }
    // the previous tile's first-occurrence probes, before anything new is requested (what is still in flight now was
    // requested long ago): a bit that was set already makes the read a repeat, which goes to the table
    bool repeat = false;
    if (bits) repeat = first_on != 0u && ((first_old >> ((uint32_t)first_idx & 31u)) & 1u) != 0u;
    const uint64_t repeat_idx = first_idx;
    // the quality region is free: request the next tile's quality lines (they have until that tile's
    // quality stage to arrive)
    seq_ready = next_full;
    qual_ready = false;
    if (next_full && with_qual) {
      wave_lds_fence();
      dma_tile(ops.qtile, qual + (tn << 6) * stride, full_bytes, lane);
      qual_ready = true;
    }
    // Results::add_count (info.rs:761-767): one no-return atomic into the dense counter table, or,
    // when captures are kept raw, into the hash map slot of the tuple key.  Issued last so that the
    // fetches above are older: nothing in the next tile has to wait for the atomic to retire.
    ops.pending_add = 0;
    if (bits) {
      const uint32_t n_rep = (uint32_t)__popcll(__ballot(repeat));
      if (n_rep) {
        if (repeat) table_add_marked(pl, table, bits, repeat_idx);
        ops.pending_add += 1u;
        if (2u * n_rep > (uint32_t)__popcll(__ballot(first_on != 0u))) direct_tiles = 16u;  // mostly repeats
      }
      if (direct_tiles) --direct_tiles;
    }
    first_on = 0u;
    if (!pl.has_random) {
      bool add = active && outcome == kMatched && !pl.discard_counts && !(pl.abl() & 0x4u);
      if (hot && !pl.sparse && __any(add)) add = hot_count(add, (uint32_t)r.dense_idx);
      if (__any(add)) {
        if (add) {
          if (pl.sparse) {
            atomicAdd(&vals[map_slot(slots, smask, r.dense_idx)], 1u);
          } else if (bits && direct_tiles == 0u) {
            first_old = atomicOr(&bits[r.dense_idx >> 5], 1u << ((uint32_t)r.dense_idx & 31u));
            first_idx = r.dense_idx;
            first_on = 1u;
          } else {
            table_add_marked(pl, table, bits, r.dense_idx);
          }
        }
        ops.pending_add += pl.sparse ? 0u : 1u;  // map_slot's compare-and-swap is waited for; the add is not
      }
    }
    ops.mark(9);
    if (ops.defer_) {
      // (here, at the end: what the count above needed is dead by now, and the searches' registers are not added to it)
      const DevGroup& G = pl.groups[0];
      const uint32_t fail_k = G.type == kGroupSample ? kSampleBarcode : kBarcode;  // parse.rs:132-140
      const uint64_t t0 = (uint64_t)blockIdx.x * (kTPB / 64) + wave;
      // verdicts of up to four queued captures, one per lane 0 / 16 / 32 / 48 (`mine`): outcome counters, trace, count
      auto settle = [&](bool mine, uint32_t res, uint32_t where) {
        const bool ok = mine && res != kFail;
        acc_cnt[kMatched] += (uint32_t)__popcll(__ballot(ok));
        const uint32_t n_fail = (uint32_t)__popcll(__ballot(mine && !ok));
        acc_cnt[kBarcode] += fail_k == kBarcode ? n_fail : 0u;
        acc_cnt[kSampleBarcode] += fail_k == kSampleBarcode ? n_fail : 0u;
        const uint64_t didx = ok ? (uint64_t)res * G.table_stride : 0ull;
        if (trace_outcome && mine) {
          const uint64_t rd = ((t0 + (uint64_t)(where >> 6) * n_waves) << 6) + (where & 63u);
          trace_outcome[rd] = (uint8_t)(ok ? (uint32_t)kMatched : fail_k);
          trace_idx[rd] = didx;
        }
        bool add = ok && !pl.discard_counts && !(pl.abl() & 0x4u);
        if (hot) add = hot_count(add, (uint32_t)didx);
        if (add) table_add_marked(pl, table, bits, didx);
      };
      unsigned long long pendm = __ballot(active && outcome == kPending);
      for (;;) {
        if (pendm) {
          // as many of the tile's pending reads as there is room for; r carries the capture (process_read)
          const uint32_t room = kQueueEntries - dq_plain - dq_n;
          const bool mine = ((pendm >> lane) & 1ull) != 0ull;
          const bool take = mine && (uint32_t)__popcll(pendm & ((1ull << lane) - 1ull)) < room;
          const unsigned long long tm = __ballot(take), tn_m = __ballot(take && r.rcode != 0u);
          const unsigned long long below = (1ull << lane) - 1ull;
          if (take) {
            const bool with_n = r.rcode != 0u;
            const uint32_t at = with_n ? kQueueEntries - 1u - dq_n - (uint32_t)__popcll(tn_m & below)
                                       : dq_plain + (uint32_t)__popcll(tm & ~tn_m & below);
            dq[at] = make_uint4((uint32_t)r.dense_idx, (uint32_t)(r.dense_idx >> 32), (uint32_t)r.rcode, (tile_seq << 6) | lane);
          }
          dq_n += (uint32_t)__popcll(tn_m);
          dq_plain += (uint32_t)__popcll(tm & ~tn_m);
          pendm &= ~tm;
          wave_lds_fence();
        }
        // captures with 'N': one at a time -- the coarse index over their substitutions, then the other indexes
        while (dq_n) {
          --dq_n;
          const uint4 q = dq[kQueueEntries - 1u - dq_n];
          const uint32_t b1 = rdlane(q.x, 0), b2 = rdlane(q.y, 0), bn = rdlane(q.z, 0), where = rdlane(q.w, 0);
          uint32_t res;
          if (!coarse_with_n(G, b1, b2, bn, res)) res = (pl.abl() & 0x100000u) ? kFail : wave_fix_error_seeded(G, b1, b2, bn, true);
          settle(lane == 0u, res, where);
        }
        // plain captures: four share a pass of the coarse index; with no four together they wait for the next tile,
        // unless this is the wave's last or the queue has to take more of this one
        while (dq_plain >= ((tn < n_tiles && !pendm) ? 4u : 1u)) {
          const uint32_t n_seg = dq_plain < 4u ? dq_plain : 4u;
          dq_plain -= n_seg;
          const uint32_t seg = lane >> 4;
          const bool seg_on = seg < n_seg;
          const uint4 q = dq[dq_plain + (seg_on ? seg : 0u)];
          CoarseVote vote;
          coarse_probe_x4(G, q.x, q.y, seg_on, vote);
          uint32_t res = kFail;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if ((uint32_t)k >= n_seg) break;
            uint32_t kmin, idx, rk;
            bool uniq;
            coarse_segment(vote, k, kmin, uniq, idx);
            if (kmin != 0xFFFFFFFFu && kmin <= G.seed2_nb)  // decided
              rk = (uniq && (kmin == 0u || kmin - 1u <= G.max_err)) ? idx : kFail;
            else if (pl.abl() & 0x100000u)
              rk = kFail;  // experiment: what the coarse index cannot decide simply fails
            else
              rk = wave_fix_error_seeded(G, rdlane(q.x, 16 * k), rdlane(q.y, 16 * k), 0u, true);
            res = (int)lane == 16 * k ? rk : res;
          }
          settle(seg_on && (lane & 15u) == 0u, res, q.w);
        }
        if (!pendm) break;
      }
      ++tile_seq;
    }
  }
  if (bits && first_on != 0u && ((first_old >> ((uint32_t)first_idx & 31u)) & 1u) != 0u) table_add_marked(pl, table, bits, first_idx);
#ifdef BC_PROFILE
  if (lane == 0)
    for (int k = 0; k < 12; ++k) atomicAdd(&bc_g_profile[k], ops.acc[k]);
#endif

  if (lane == 0) {
#pragma unroll
    for (uint32_t k = 0; k < kNCounters; ++k)
      if (acc_cnt[k]) atomicAdd(&s_cnt[k], acc_cnt[k]);
  }
  __syncthreads();
  if (tid < kNCounters) {
    const uint32_t v = s_cnt[tid];
    if (v) atomicAdd(&counters[tid], (unsigned long long)v);
  }
  if (hot) {  // the cached counts, one table add per used slot
    for (uint32_t i = tid; i < kHotSlots; i += kTPB) {
      const uint32_t c = hot_cnt[i];
      if (c) {
        atomicAdd(&table[hot_tag[i]], c);
        if (bits && pl.dirty_off) reinterpret_cast<uint8_t*>(bits + pl.dirty_off)[hot_tag[i] >> 6] = (uint8_t)1;
      }
    }
  }
}

}  // namespace bc
