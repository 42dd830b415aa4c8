// bc_lane.h -- everything ONE lane does for ONE read (SequenceParser::match_seq, parse.rs:89-148).
//
// Layout idea: a read is turned into bit planes (bit i of plane word w = base 32*w+i), so that
//  * the leftmost exact anchor of the format regex (parse.rs:92-95, 151-157) is an AND over the
//    constant positions of shifted "base == letter" vectors -- every candidate offset at once;
//  * constant-region repair (fix_constant_region, parse.rs:287-313 -> fix_error, parse.rs:553-593)
//    is a bit-sliced add of shifted "base != letter and base != N" vectors -- the mismatch count
//    of every window at once -- followed by a bit-sliced unique-minimum search;
//  * a captured barcode is a handful of bits that index a precomputed correction table.
// Wave-level steps (vote, the cooperative Hamming search) go through `Ops`, so the same code runs
// lane-per-read on the GPU and, for tests only, one read at a time on the host (tests/emu).
#pragma once
#include "bc_device_plan.h"
#include "bc_intrin.h"

namespace bc {

template <int NW>
struct Planes {
  uint32_t p1[NW];  // ASCII bit 1 of each base
  uint32_t p2[NW];  // ASCII bit 2 of each base
  uint32_t pn[NW];  // base is 'N'
  uint32_t px[NW];  // base is none of A,C,G,T,N (always a mismatch, never a wildcard)
};

BC_HD uint64_t hash64(uint64_t x) {
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

// ---- ASCII -> bit planes -------------------------------------------------------------------
// Fast conversion: 4 bases per dword, three v_dot4_u32_u8 gathers per dword.  `bad` ends up
// non-zero when any converted byte is not one of A,C,G,T,N; the planes are then only valid
// after pack_exact() has rebuilt pn/px.
template <int NW>
BC_HD void pack_read(const uint32_t* t32, uint32_t base, uint32_t nd, Planes<NW>& P, uint32_t& bad) {
  const uint32_t a = base & 3u;
  const uint32_t k0 = base >> 2;
  uint32_t prev = t32[k0];
  bad = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    uint32_t o1 = 0, o2 = 0, on = 0;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      uint32_t a1 = 0, a2 = 0, an = 0;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t idx = (uint32_t)(w * 8 + h * 2 + j);
        if (idx < nd) {  // wave-uniform
          const uint32_t nxt = t32[k0 + idx + 1];
          const uint32_t d = alignbyte(nxt, prev, a);
          prev = nxt;
          const uint32_t wt = j ? 0x80402010u : 0x08040201u;
          a1 = udot4(d & 0x02020202u, wt, a1);
          a2 = udot4(d & 0x04040404u, wt, a2);
          an = udot4(d & 0x08080808u, wt, an);
          const uint32_t ex = perm(0x4EFFFFFFu, 0x47544341u, (d >> 1) & 0x07070707u);
          bad = sad_u8(d, ex, bad);
        }
      }
      o1 |= (a1 >> 1) << (8 * h);
      o2 |= (a2 >> 2) << (8 * h);
      on |= (an >> 3) << (8 * h);
    }
    P.p1[w] = o1;
    P.p2[w] = o2;
    P.pn[w] = on;
    P.px[w] = 0;
  }
}

// Exact classification of every byte (slow path, taken by a whole wave when any lane saw a
// byte outside ACGTN): rebuilds pn and px; `hi` gets the positions of bytes >= 0x80.
template <int NW>
BC_HD void pack_exact(const uint32_t* t32, uint32_t base, uint32_t nd, Planes<NW>& P, uint32_t (&hi)[NW]) {
  const uint32_t a = base & 3u;
  const uint32_t k0 = base >> 2;
  uint32_t prev = t32[k0];
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    uint32_t ov = 0, on = 0, oh = 0;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      uint32_t av = 0, an = 0, ah = 0;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t idx = (uint32_t)(w * 8 + h * 2 + j);
        if (idx < nd) {
          const uint32_t nxt = t32[k0 + idx + 1];
          const uint32_t d = alignbyte(nxt, prev, a);
          prev = nxt;
          const uint32_t wt = j ? 0x80402010u : 0x08040201u;
          const uint32_t ex = perm(0u, 0x47544341u, (d >> 1) & 0x03030303u);
          av = udot4(zero_bytes(d ^ ex) >> 7, wt, av);
          an = udot4(zero_bytes(d ^ 0x4E4E4E4Eu) >> 7, wt, an);
          ah = udot4((d >> 7) & 0x01010101u, wt, ah);
        }
      }
      ov |= av << (8 * h);
      on |= an << (8 * h);
      oh |= ah << (8 * h);
    }
    P.pn[w] = on;
    P.px[w] = ~(ov | on);
    hi[w] = oh;
  }
}

// ---- multi-word vector helpers ----------------------------------------------------------------
template <int NW>
BC_HD void shr_uniform(uint32_t (&v)[NW], uint32_t sh) {  // sh in [0,31], wave-uniform
#pragma unroll
  for (int i = 0; i < NW; ++i) v[i] = alignbit(i + 1 < NW ? v[i + 1] : 0u, v[i], sh);
}

template <int NW>
BC_HD void shr_lane(uint32_t (&v)[NW], uint32_t s) {  // per-lane shift, s < 32*NW
#pragma unroll
  for (int bit = 0; (1 << bit) < NW; ++bit) {
    const bool t = (s >> (5 + bit)) & 1u;
    const int d = 1 << bit;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const uint32_t up = (i + d < NW) ? v[i + d] : 0u;
      v[i] = t ? up : v[i];
    }
  }
  const uint32_t sh = s & 31u;
#pragma unroll
  for (int i = 0; i < NW; ++i) v[i] = alignbit(i + 1 < NW ? v[i + 1] : 0u, v[i], sh);
}

// bits [off, off+len) of a vector, off/len wave-uniform, len <= 32
template <int NW>
BC_HD uint32_t extract_uniform(const uint32_t (&v)[NW], uint32_t off, uint32_t len) {
  const uint32_t k = off >> 5;
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    if (k == (uint32_t)i) {
      lo = v[i];
      hi = (i + 1 < NW) ? v[i + 1] : 0u;
    }
  }
  return alignbit(hi, lo, off & 31u) & lowmask(len);
}

// bit `pos` of a vector, pos per lane
template <int NW>
BC_HD uint32_t test_bit(const uint32_t (&v)[NW], uint32_t pos) {
  uint32_t wsel = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) wsel = ((pos >> 5) == (uint32_t)i) ? v[i] : wsel;
  return (wsel >> (pos & 31u)) & 1u;
}

template <int NW>
BC_HD void low_bits(uint32_t (&m)[NW], uint32_t n) {  // bits [0, n) set
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const uint32_t lo = 32u * (uint32_t)w;
    m[w] = n > lo ? lowmask(n - lo) : 0u;
  }
}

// "base at position i equals letter c" for every i (c in kCodeA..kCodeG)
template <int NW>
BC_HD void eq_vector(const Planes<NW>& P, const uint32_t (&inr)[NW], int c, uint32_t (&v)[NW]) {
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const uint32_t b1 = (c & 1) ? P.p1[w] : ~P.p1[w];
    const uint32_t b2 = (c & 2) ? P.p2[w] : ~P.p2[w];
    v[w] = b1 & b2 & inr[w] & ~(P.pn[w] | P.px[w]);
  }
}

// ---- quality (RawSequenceRead::low_quality, parse.rs:331-375) ----------------------------------
// sum of (byte - 33) mod 256 over `len` bytes starting at byte address `addr` of the quality tile
BC_HD uint32_t score_sum(const uint32_t* q32, uint32_t addr, uint32_t len) {
  const uint32_t a = addr & 3u;
  const uint32_t k0 = addr >> 2;
  const uint32_t nd = (len + 3u) >> 2;  // wave-uniform
  uint32_t prev = q32[k0];
  uint32_t sum = 0;
  for (uint32_t j = 0; j < nd; ++j) {
    const uint32_t nxt = q32[k0 + j + 1];
    uint32_t d = alignbyte(nxt, prev, a);
    prev = nxt;
    // per-byte d - 0x21 with u8 wrap (`ch as u8 - 33` in a release build, parse.rs:326)
    d = ((d | 0x80808080u) - 0x21212121u) ^ ((d & 0x80808080u) ^ 0x80808080u);
    if (j == nd - 1 && (len & 3u)) d &= lowmask(8u * (len & 3u));
    sum = sad_u8(d, 0u, sum);
  }
  return sum;
}

// ---- distance of a capture to one reference (the inner loop of fix_error, parse.rs:562-575) ----
// q*: capture planes over qlen positions; r*: reference planes, rl = its length.
// Positions compared = the common prefix (zip, parse.rs:568); 'N' on either side is free.
BC_HD uint32_t ref_distance(uint32_t q1, uint32_t q2, uint32_t qn, uint32_t qx, uint32_t qlen, uint32_t r1, uint32_t r2,
                            uint32_t rn, uint32_t rl, bool& exact) {
  const uint32_t cmp = lowmask(rl < qlen ? rl : qlen);
  const uint32_t diff = ((q1 ^ r1) | (q2 ^ r2));
  // string equality (AHashSet::contains, parse.rs:457, 489): same length, same N positions, same bases
  exact = (rl == qlen) && (qx == 0u) && (qn == rn) && ((diff & ~qn) == 0u);
  return popc((diff | qx) & ~qn & ~rn & cmp);
}

// running state of a unique-minimum search.  key = 0 for the reference that IS the capture
// (AHashSet::contains is tried first, parse.rs:457, 489), distance + 1 otherwise.
struct Nearest {
  uint32_t key;    // smallest key seen
  uint32_t idx;    // a reference with that key
  uint32_t count;  // how many references have it (saturating at 2)
};
BC_HD void nearest_init(Nearest& s) {
  s.key = 0xFFFFFFFFu;
  s.idx = kFail;
  s.count = 0;
}
BC_HD void nearest_add(Nearest& s, uint32_t d, uint32_t idx, bool exact) {
  const uint32_t k = exact ? 0u : d + 1u;
  if (k < s.key) {
    s.key = k;
    s.idx = idx;
    s.count = 1;
  } else if (k == s.key) {
    s.count = 2;
  }
}
// fix_error's verdict (parse.rs:577-592), order independent:
// the exact member if there is one, else the unique reference at the minimum distance when that
// distance is within the budget
BC_HD uint32_t nearest_result(uint32_t key, uint32_t idx, uint32_t count, uint32_t max_err) {
  return (count == 1u && (key == 0u || key - 1u <= max_err)) ? idx : kFail;
}

// ---- the per-read decision tree ---------------------------------------------------------------
struct ReadResult {
  uint32_t outcome;    // Outcome; kMatched means "passed every test" (duplicate detection is later)
  uint64_t dense_idx;  // index into the dense (sample, tuple) counter table
};

// Ops must provide:
//   bool any(bool)                               -- wave vote
//   uint32_t nearest(const DevGroup&, q1,q2,qn,qx, bool need) -- cooperative Hamming search, every lane calls it
template <class Ops, int NW>
BC_HD ReadResult process_read(const DevPlan& pl, Ops& ops, const uint32_t* seq32, const uint32_t* qual32, uint32_t base,
                              uint32_t len, uint32_t nd, bool active) {
  ReadResult res;
  res.outcome = kMatched;
  res.dense_idx = 0;

  Planes<NW> P;
  uint32_t bad;
  if (pl.ablate & 0x40u) {
    bad = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { P.p1[w] = base * 2654435761u + w; P.p2[w] = base * 40503u + w; P.pn[w] = 0; P.px[w] = 0; }
  } else {
    pack_read<NW>(seq32, base, nd, P, bad);
  }
  uint32_t inr[NW];
  low_bits<NW>(inr, len);
  bool unsupported = false;
  if (ops.any(active && bad != 0u)) {
    uint32_t hi[NW];
    pack_exact<NW>(seq32, base, nd, P, hi);
    uint32_t h = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) h |= hi[w] & inr[w];
    unsupported = h != 0u;  // non-ASCII: the reference's char positions no longer equal byte positions
  }
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    P.pn[w] &= inr[w];
    P.px[w] &= inr[w];
  }

  const uint32_t L = pl.L;
  // ---- leftmost exact anchor: Regex::is_match / captures (parse.rs:92-95, 153-156) -------------
  uint32_t acc[NW];
  low_bits<NW>(acc, len >= L ? len - L + 1u : 0u);  // offsets o with o + L <= len
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint32_t ns = (pl.ablate & 0x10u) ? 0u : pl.n_steps[c];
    if (ns) {
      uint32_t v[NW];
      eq_vector<NW>(P, inr, c, v);
      for (uint32_t s = 0; s < ns; ++s) {
        const uint32_t st = pl.steps[c][s];
        shr_uniform<NW>(v, st & 31u);
        if (st & 0x80u) {
#pragma unroll
          for (int w = 0; w < NW; ++w) acc[w] &= v[w];
        }
      }
    }
  }
  // scheme 'N' positions must be one of A,G,C,T ([AGCT]{n}, info.rs:291-294)
  uint32_t fnok[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) fnok[w] = 0xFFFFFFFFu;
  if (pl.has_fmtn) {
    uint32_t v[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) v[w] = inr[w] & ~(P.pn[w] | P.px[w]);
    const uint32_t ns = pl.n_steps[kClassFmtN];
    for (uint32_t s = 0; s < ns; ++s) {
      const uint32_t st = pl.steps[kClassFmtN][s];
      shr_uniform<NW>(v, st & 31u);
      if (st & 0x80u) {
#pragma unroll
        for (int w = 0; w < NW; ++w) fnok[w] &= v[w];
      }
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) acc[w] &= fnok[w];
  }
  uint32_t start = 0;
  bool found = false;
#pragma unroll
  for (int w = NW - 1; w >= 0; --w) {
    if (acc[w]) {
      start = 32u * (uint32_t)w + ctz(acc[w]);
      found = true;
    }
  }

  // ---- constant-region repair: fix_constant_region (parse.rs:287-313) --------------------------
  bool repaired = false;
  if (!(pl.ablate & 0x1u) && ops.any(active && !found && !unsupported)) {
    // windows 0 .. len-L-1 only: the last window is never tested (parse.rs:291-295)
    uint32_t cand[NW];
    low_bits<NW>(cand, len > L ? len - L : 0u);
    uint32_t cnt[5][NW];
    uint32_t ovf[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      ovf[w] = 0;
#pragma unroll
      for (int b = 0; b < 5; ++b) cnt[b][w] = 0;
    }
    const uint32_t nb = pl.nb;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint32_t ns = pl.n_steps[c];
      if (ns) {
        uint32_t v[NW];
        eq_vector<NW>(P, inr, c, v);
        // mismatch = differs and neither side is 'N' (parse.rs:569); format 'N's are not in the program
#pragma unroll
        for (int w = 0; w < NW; ++w) v[w] = inr[w] & ~P.pn[w] & ~v[w];
        for (uint32_t s = 0; s < ns; ++s) {
          const uint32_t st = pl.steps[c][s];
          shr_uniform<NW>(v, st & 31u);
          if (st & 0x80u) {
#pragma unroll
            for (int w = 0; w < NW; ++w) {
              uint32_t carry = v[w];
#pragma unroll
              for (int b = 0; b < 5; ++b) {
                if ((uint32_t)b < nb) {
                  const uint32_t t = cnt[b][w] & carry;
                  cnt[b][w] ^= carry;
                  carry = t;
                }
              }
              ovf[w] |= carry;
            }
          }
        }
      }
    }
    // unique minimum over the candidate windows (fix_error, parse.rs:577-592)
#pragma unroll
    for (int w = 0; w < NW; ++w) cand[w] &= ~ovf[w];
#pragma unroll
    for (int b = 4; b >= 0; --b) {
      if ((uint32_t)b < nb) {
        uint32_t t[NW];
        uint32_t nz = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          t[w] = cand[w] & ~cnt[b][w];
          nz |= t[w];
        }
#pragma unroll
        for (int w = 0; w < NW; ++w) cand[w] = nz ? t[w] : cand[w];
      }
    }
    uint32_t n_min = 0, val = 0, pos = 0;
#pragma unroll
    for (int w = NW - 1; w >= 0; --w) {
      n_min += popc(cand[w]);
      if (cand[w]) pos = 32u * (uint32_t)w + ctz(cand[w]);
    }
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      uint32_t o = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) o |= cand[w] & cnt[b][w];
      val |= (o ? 1u : 0u) << b;
    }
    if (!found && n_min == 1u && val <= pl.max_const) {
      // the window replaces the read, constants overwritten by the format
      // (insert_barcodes_constant_region, parse.rs:270-283); the regex then has to match it
      // at offset 0, which only the scheme-N positions can still prevent
      if (test_bit<NW>(fnok, pos)) {
        found = true;
        repaired = true;
        start = pos;
      }
    }
  }

  uint32_t outcome = kMatched;
  if (!found) outcome = kConstantRegion;  // parse.rs:145

  // ---- quality filter (parse.rs:98-119, 331-375) ------------------------------------------------
  if (pl.quality_on && !(pl.ablate & 0x8u)) {
    // after a repair the quality line is read from offset 0
    const uint32_t qstart = repaired ? 0u : start;
    const uint32_t avail = len - qstart;  // quality line assumed as long as the sequence line
    const uint32_t zip = avail < pl.RL ? avail : pl.RL;
    bool low = false;
    for (uint32_t r = 0; r < pl.n_runs; ++r) {
      const uint32_t ro = pl.run_off[r], rl = pl.run_len[r];
      // a run is only evaluated when the zip continues past it (parse.rs:348-356)
      const bool evaluated = (ro + rl) < zip;
      const uint32_t sum = score_sum(qual32, base + (found ? qstart : 0u) + ro, rl);
      low = low || (evaluated && sum < pl.run_thr[r]);
    }
    if (outcome == kMatched && low) outcome = kLowQuality;  // parse.rs:111
  }

  // ---- barcodes: SequenceMatchResult::new (parse.rs:439-524) -----------------------------------
  // bring the construct to bit 0 so that every capture sits at a wave-uniform position
  if (!found) start = 0;
  shr_lane<NW>(P.p1, start);
  shr_lane<NW>(P.p2, start);
  shr_lane<NW>(P.pn, start);
  shr_lane<NW>(P.px, start);
  uint64_t didx = 0;
  for (uint32_t g = 0; g < ((pl.ablate & 0x20u) ? 0u : pl.n_groups); ++g) {
    const DevGroup& G = pl.groups[g];
    if (G.mode == kSetNone) continue;
    const uint32_t q1 = extract_uniform<NW>(P.p1, G.off, G.len);
    const uint32_t q2 = extract_uniform<NW>(P.p2, G.off, G.len);
    const uint32_t qn = extract_uniform<NW>(P.pn, G.off, G.len);
    const uint32_t qx = extract_uniform<NW>(P.px, G.off, G.len);
    const bool live = active && outcome == kMatched;
    const bool clean = (qn | qx) == 0u;
    uint32_t r = kFail;
    bool need = false;
    if (G.mode == kSetDirect) {
      if (live) {
        if (clean) {
          const uint32_t t = G.dtable[q1 | (q2 << G.len)];
          r = t == kFail16 ? kFail : t;
        } else {
          need = true;
        }
      }
    } else if (G.mode == kSetHash) {
      if (live) {
        need = true;
        if (clean) {
          const uint64_t key = (uint64_t)q1 | ((uint64_t)q2 << 32);
          uint32_t h = (uint32_t)hash64(key) & G.hmask;
          for (;;) {
            const uint32_t v = G.hvals[h];
            if (v == kFail) break;
            if (G.hkeys[h] == key) {
              r = v;
              need = false;
              break;
            }
            h = (h + 1u) & G.hmask;
          }
        }
      }
    } else {
      need = live;
    }
    if (pl.ablate & 0x2u) need = false;
    const uint32_t rr = ops.nearest(G, q1, q2, qn, qx, need);
    if (need) r = rr;
    if (live) {
      if (r == kFail)
        outcome = (G.type == kGroupSample) ? kSampleBarcode : kBarcode;  // parse.rs:132-140
      else
        didx += (uint64_t)r * G.table_stride;
    }
  }
  if (unsupported) outcome = kUnsupported;
  res.outcome = outcome;
  res.dense_idx = didx;
  return res;
}

}  // namespace bc
