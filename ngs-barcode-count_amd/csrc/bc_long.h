// bc_long.h -- the wave-per-read form of SequenceParser::match_seq (parse.rs:89-148) for inputs the lane-per-read
// kernel (bc_lane.h / bc_kernel.h) is not built for: reads of more than 320 bases, barcode groups or known barcodes
// of more than 32 bases, more than 31 tolerated constant-region errors.  Nothing is packed into bit planes here:
// one wavefront walks one read's bytes the way the reference walks its chars --
//   * anchor (Regex::is_match / captures, parse.rs:92-95, 153-156): lanes = candidate offsets,
//   * repair (fix_constant_region, parse.rs:287-313): lanes = windows, wave min-reduce with a tie count,
//   * quality runs (low_quality, parse.rs:331-375): lanes = bytes of the run,
//   * barcodes (SequenceMatchResult::new, parse.rs:439-524; fix_error, parse.rs:553-593): lanes = references,
// -- and one lane counts the result.  It is the slow path: two to three orders of magnitude below the lane-per-read
// kernel, in return for no length limits.
// Wide keys: captures kept raw (no conversion file: README.md "Barcode-seq", info.rs:742-757) and random barcodes
// whose base-5 codes no longer fit the 64-bit tuple key of the lane-per-read kernel -- more than 27 bases, or several
// captures that together overflow it -- are counted here under a key of several 64-bit words: word 0 a fingerprint of
// the rest (never the empty-slot value), then the captures bit by bit (three bit planes per raw capture: ASCII bit 1,
// ASCII bit 2, 'N'; 32 bits per index into a known set), in a device hash table whose slots are that wide
// (wide_find_or_insert).  Up to kMaxKeyWords words: about 145 raw bases per read.
#pragma once
#include "bc_device_plan.h"

namespace bc {

constexpr int kMaxKeyWords = 8;
constexpr unsigned long long kWideEmpty = ~0ull;

struct LongGroup {
  uint32_t type;     // GroupType
  uint32_t off, len; // capture inside a match
  uint32_t n_refs;   // 0: no known set, the capture's base-5 code is the key digit
  uint32_t max_err;
  uint32_t key_bit;  // wide keys: first payload bit of this group's field (32 bits of index, or 3 * len plane bits)
  uint64_t table_stride;
  uint64_t ref_text_a;  // the references' bytes, one after the other
  uint64_t ref_off_a;   // u32[n_refs + 1]: where each one starts
};

struct LongPlan {
  uint32_t L, RL, max_const, quality_on, n_runs, n_groups, n_const, n_fmtn;
  uint32_t has_random, rnd_off, rnd_len, sparse, discard_counts, no_repair;
  uint32_t wide;       // keys are key_words 64-bit words (see the head of this file); 0: one mixed-radix 64-bit key
  uint32_t key_words;  // fingerprint + payload words
  uint32_t rnd_bit;    // first payload bit of the random barcode's planes
  uint32_t pad_;
  uint64_t rspace;
  uint64_t const_pos_a;  // u32[n_const]: format positions holding a constant base ...
  uint64_t const_chr_a;  // u8[n_const]:  ... and the base (upper case)
  uint64_t fmtn_pos_a;   // u32[n_fmtn]: scheme-N positions ([AGCT], info.rs:291-294)
  uint32_t run_off[kMaxRuns], run_len[kMaxRuns], run_thr[kMaxRuns];
  LongGroup groups[kMaxGroups];
};

#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t long_wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
  return v;
}
__device__ __forceinline__ uint32_t long_wave_min(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t t = (uint32_t)__shfl_xor((int)v, o);
    v = t < v ? t : v;
  }
  return v;
}

// ---- wide keys ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t wide_mix(uint64_t x) {
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}
// word 0 of a key: a fingerprint of its payload words, top bit clear (so it never equals kWideEmpty)
__device__ __forceinline__ uint64_t wide_fingerprint(const uint64_t* key, uint32_t W) {
  uint64_t h = 0x9E3779B97F4A7C15ull;
  for (uint32_t w = 1; w < W; ++w) h = wide_mix(h ^ key[w]) + w;
  return h >> 1;
}
// `n` (<= 32) bits of `v` into the payload at bit `at` (payload bit 0 = bit 0 of word 1)
__device__ __forceinline__ void wide_put(uint64_t* key, uint32_t at, uint32_t v, uint32_t n) {
  const uint32_t w = 1u + (at >> 6), sh = at & 63u;
  const uint64_t x = (uint64_t)v & ((n >= 32u) ? 0xFFFFFFFFull : ((1ull << n) - 1ull));
  key[w] |= x << sh;
  if (sh + n > 64u) key[w + 1u] |= x >> (64u - sh);
}
// the three bit planes of a raw capture (A=00 C=10 T=01 G=11 in (bit 1, bit 2) of the ASCII code; 'N': its own plane,
// the other two clear) at payload bit `at`; false when a byte has no code (none of A,C,G,T,N)
__device__ __forceinline__ bool wide_put_capture(uint64_t* key, uint32_t at, const uint8_t* cap, uint32_t len) {
  bool ok = true;
  for (uint32_t i = 0; i < len; ++i) {
    const uint8_t c = cap[i];
    const bool n = c == 'N';
    ok = ok && (n || c == 'A' || c == 'C' || c == 'G' || c == 'T');
    if (!n && ((c >> 1) & 1u)) wide_put(key, at + i, 1u, 1u);
    if (!n && ((c >> 2) & 1u)) wide_put(key, at + len + i, 1u, 1u);
    if (n) wide_put(key, at + 2u * len + i, 1u, 1u);
  }
  return ok;
}
// Slot of a wide key in a table of W-word slots, inserting it when absent (is_new).  ONE lane of a wavefront at a
// time may be inside: the owner of a freshly claimed slot publishes the payload and then the slot's ready flag; a
// lane that meets its own fingerprint in a slot waits for that flag before it compares -- waiting on a lane of its
// own wavefront would never end.
__device__ __forceinline__ uint64_t wide_find_or_insert(unsigned long long* __restrict__ slots, uint32_t* __restrict__ ready,
                                                        uint64_t mask, uint32_t W, const uint64_t* key, bool& is_new) {
  uint64_t h = wide_mix(key[0]) & mask;
  for (;;) {
    unsigned long long* s = slots + h * W;
    const unsigned long long old = atomicCAS(s, kWideEmpty, (unsigned long long)key[0]);
    if (old == kWideEmpty) {
      for (uint32_t w = 1; w < W; ++w) __hip_atomic_store(s + w, (unsigned long long)key[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ready + h, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      is_new = true;
      return h;
    }
    if (old == (unsigned long long)key[0]) {
      while (__hip_atomic_load(ready + h, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(2);
      bool same = true;
      for (uint32_t w = 1; w < W; ++w)
        same = same && __hip_atomic_load(s + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)key[w];
      if (same) {
        is_new = false;
        return h;
      }
    }
    h = (h + 1) & mask;
  }
}

// one read by one wavefront; returns the Outcome and, for a read that passed every test, its table / key index and
// the base-5 code of its random barcode -- or, for a plan with wide keys, the key in wide_key[0 .. key_words)
// (wave-private memory, written by lane 0)
__device__ __forceinline__ uint32_t long_match_read(const LongPlan& pl, const uint8_t* __restrict__ seq, uint32_t len,
                                                    const uint8_t* __restrict__ qual, uint32_t qlen, uint64_t& dense_idx,
                                                    uint64_t& rcode, uint64_t* wide_key) {
  const uint32_t lane = __lane_id();
  const uint32_t L = pl.L;
  const BC_GLOBAL uint32_t* cpos = reinterpret_cast<const BC_GLOBAL uint32_t*>(pl.const_pos_a);
  const BC_GLOBAL uint8_t* cchr = reinterpret_cast<const BC_GLOBAL uint8_t*>(pl.const_chr_a);
  const BC_GLOBAL uint32_t* npos = reinterpret_cast<const BC_GLOBAL uint32_t*>(pl.fmtn_pos_a);
  dense_idx = 0;
  rcode = 0;
  // bytes >= 0x80: the reference's char positions no longer equal byte positions -- not judged here
  {
    bool hi = false;
    for (uint32_t i = lane; i < len; i += 64) hi = hi || seq[i] >= 0x80u;
    if (__any(hi)) return kUnsupported;
  }
  auto valid_base = [](uint8_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; };

  // ---- anchor: the smallest offset at which every constant base matches and every scheme-N base is one of ACGT ----
  bool found = false, repaired = false;
  uint32_t start = 0;
  if (len >= L) {
    const uint32_t n_off = len - L + 1u;
    for (uint32_t o0 = 0; o0 < n_off && !found; o0 += 64u) {
      const uint32_t o = o0 + lane;
      bool ok = o < n_off;
      for (uint32_t k = 0; k < pl.n_const; ++k) {
        if (!__any(ok)) break;
        if (ok) ok = seq[o + cpos[k]] == cchr[k];
      }
      for (uint32_t k = 0; k < pl.n_fmtn && __any(ok); ++k)
        if (ok) ok = valid_base(seq[o + npos[k]]);
      const unsigned long long m = __ballot(ok);
      if (m) {
        found = true;
        start = o0 + (uint32_t)__ffsll(m) - 1u;
      }
    }
  }
  // ---- repair: the unique window among 0 .. len-L-1 with the fewest constant mismatches, within the budget -------
  if (!found && len > L && !pl.no_repair) {
    const uint32_t n_win = len - L;  // the last window is never tested (parse.rs:291-295)
    uint32_t best = 0xFFFFFFFFu, best_at = 0, ties = 0;
    for (uint32_t i = lane; i < n_win; i += 64u) {
      uint32_t mm = 0;
      for (uint32_t k = 0; k < pl.n_const; ++k) {
        const uint8_t c = seq[i + cpos[k]];
        mm += (c != cchr[k] && c != 'N') ? 1u : 0u;  // 'N' is free (parse.rs:569); constants are never 'N'
      }
      if (mm < best) {
        best = mm;
        best_at = i;
        ties = 1;
      } else if (mm == best) {
        ties = 2;
      }
    }
    const uint32_t gmin = long_wave_min(best);
    const unsigned long long holders = __ballot(best == gmin && best != 0xFFFFFFFFu);
    if (holders && gmin <= pl.max_const && __popcll(holders) == 1) {
      const int src = __ffsll(holders) - 1;
      if ((uint32_t)__shfl((int)ties, src) == 1u) {
        const uint32_t at = (uint32_t)__shfl((int)best_at, src);
        // the window replaces the read, constants overwritten by the format (parse.rs:270-283); the regex then has
        // to match it at offset 0, which only the scheme-N positions can still prevent
        bool ok = true;
        for (uint32_t k = lane; k < pl.n_fmtn; k += 64u) ok = ok && valid_base(seq[at + npos[k]]);
        if (__all(ok)) {
          found = true;
          repaired = true;
          start = at;
        }
      }
    }
  }
  if (!found) return kConstantRegion;  // parse.rs:145

  // ---- quality (parse.rs:98-119, 331-375); after a repair the quality line is read from offset 0 ------------------
  if (pl.quality_on) {
    const uint32_t qstart = repaired ? 0u : start;
    const uint32_t avail = qlen > qstart ? qlen - qstart : 0u;
    const uint32_t zip = avail < pl.RL ? avail : pl.RL;
    bool low = false;
    for (uint32_t r = 0; r < pl.n_runs; ++r) {
      const uint32_t ro = pl.run_off[r], rl = pl.run_len[r];
      if (ro + rl < zip) {  // a run is only evaluated when the zip continues past it (parse.rs:348-356)
        uint32_t sum = 0;
        for (uint32_t i = lane; i < rl; i += 64u) sum += (uint32_t)((uint8_t)(qual[qstart + ro + i] - 33u));  // u8 wrap
        sum = long_wave_sum(sum);
        low = low || sum < pl.run_thr[r];
      }
    }
    if (low) return kLowQuality;  // parse.rs:111
  }

  // ---- barcodes (parse.rs:439-524) ----------------------------------------------------------------------------------
  bool unsupported = false;
  uint64_t didx = 0;
  if (pl.wide && lane == 0)
    for (uint32_t w = 0; w < pl.key_words; ++w) wide_key[w] = 0;
  for (uint32_t g = 0; g < pl.n_groups; ++g) {
    const LongGroup& G = pl.groups[g];
    const uint8_t* cap = seq + start + G.off;
    if (G.n_refs == 0 && pl.wide) {
      // no known set: the capture is taken as it is (parse.rs:453-454, 487), plane by plane into the wide key
      bool ok = true;
      if (lane == 0) ok = wide_put_capture(wide_key, G.key_bit, cap, G.len);
      if (!__shfl((int)ok, 0)) unsupported = true;  // a byte outside ACGTN has no code
      continue;
    }
    if (G.n_refs == 0) {
      // no known set: the capture is taken as it is; its base-5 code is the key digit (A,C,T,G,N = 0..4)
      uint64_t code = 0;
      for (uint32_t i = G.len; i-- > 0;) {
        const uint8_t c = cap[i];
        const uint32_t d = c == 'A' ? 0u : c == 'C' ? 1u : c == 'T' ? 2u : c == 'G' ? 3u : c == 'N' ? 4u : 5u;
        if (d == 5u) unsupported = true;  // a byte outside ACGTN has no code
        code = code * 5u + (d == 5u ? 0u : d);
      }
      didx += code * G.table_stride;
      continue;
    }
    const BC_GLOBAL uint8_t* text = reinterpret_cast<const BC_GLOBAL uint8_t*>(G.ref_text_a);
    const BC_GLOBAL uint32_t* roff = reinterpret_cast<const BC_GLOBAL uint32_t*>(G.ref_off_a);
    // key 0: the reference IS the capture (AHashSet::contains, parse.rs:457/489); else distance + 1 (fix_error)
    uint32_t key = 0xFFFFFFFFu, idx = 0, cnt = 0;
    for (uint32_t j = lane; j < G.n_refs; j += 64u) {
      const uint32_t a = roff[j], rl = roff[j + 1] - a;
      const uint32_t n = rl < G.len ? rl : G.len;  // zip stops at the shorter string (parse.rs:568)
      uint32_t d = 0;
      bool same = rl == G.len;
      for (uint32_t i = 0; i < n; ++i) {
        const uint8_t c = cap[i], r = text[a + i];
        same = same && c == r;
        d += (c != r && c != 'N' && r != 'N') ? 1u : 0u;
      }
      const uint32_t k = same ? 0u : d + 1u;
      if (k < key) {
        key = k;
        idx = j;
        cnt = 1;
      } else if (k == key) {
        cnt = 2;
      }
    }
    const uint32_t kmin = long_wave_min(key);
    const unsigned long long holders = __ballot(key == kmin && key != 0xFFFFFFFFu);
    uint32_t verdict = kFail;
    if (holders && __popcll(holders) == 1) {
      const int src = __ffsll(holders) - 1;
      const uint32_t c1 = (uint32_t)__shfl((int)cnt, src);
      if (c1 == 1u && (kmin == 0u || kmin - 1u <= G.max_err)) verdict = (uint32_t)__shfl((int)idx, src);
    }
    if (verdict == kFail) return G.type == kGroupSample ? kSampleBarcode : kBarcode;  // parse.rs:132-140
    if (pl.wide) {
      if (lane == 0) wide_put(wide_key, G.key_bit, verdict, 32u);
    } else {
      didx += (uint64_t)verdict * G.table_stride;
    }
  }
  if (pl.has_random && pl.wide) {  // kept as captured, never corrected (parse.rs:510-516)
    bool ok = true;
    if (lane == 0) ok = wide_put_capture(wide_key, pl.rnd_bit, seq + start + pl.rnd_off, pl.rnd_len);
    if (!__shfl((int)ok, 0)) unsupported = true;
  } else if (pl.has_random) {
    const uint8_t* cap = seq + start + pl.rnd_off;
    uint64_t code = 0;
    for (uint32_t i = pl.rnd_len; i-- > 0;) {
      const uint8_t c = cap[i];
      const uint32_t d = c == 'A' ? 0u : c == 'C' ? 1u : c == 'T' ? 2u : c == 'G' ? 3u : c == 'N' ? 4u : 5u;
      if (d == 5u) unsupported = true;
      code = code * 5u + (d == 5u ? 0u : d);
    }
    rcode = code;
  }
  if (unsupported) return kUnsupported;
  if (pl.wide && lane == 0) wide_key[0] = wide_fingerprint(wide_key, pl.key_words);
  dense_idx = didx;
  return kMatched;
}
#endif

}  // namespace bc
