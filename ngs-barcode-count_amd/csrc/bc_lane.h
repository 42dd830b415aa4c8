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

// candidate windows a specialised kernel has to get right: its batch's read length is a compile-time constant
#if defined(BC_JIT_TU) && JIT_READ_LEN
#define BC_STATIC_CAND(L, NWW) ((uint32_t)JIT_READ_LEN >= (L) ? (uint32_t)JIT_READ_LEN - (L) + 1u : 0u)
#else
#define BC_STATIC_CAND(L, NWW) (32u * (uint32_t)(NWW))
#endif

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
// Fast conversion, eight bases (two dwords d0, d1) at a time.  The plane bits of a base are ASCII bits 1, 2 and 3
// (A=000 C=001 T=010 G=011 N=111).  M = (d0 >> 1 on bits 0-2 of every byte) | (d1 << 3 on bits 4-6): one word that
// holds the three bits of all eight bases, so that ONE v_dot4_u32_u8 per plane (weights 1, 2, 4, 8 on the masked
// nibbles) gathers eight plane bits in base order.  Planes 2 and N are taken from M at their own bit (mask
// 0x22.., 0x44..), which scales the gathered byte by 2 / 4: the shift that puts the byte into its plane word
// absorbs that.  `bad` ends up non-zero when any converted byte is not one of A,C,G,T,N (the byte rebuilt from its
// three bits by a v_perm_b32 table differs from the byte itself); the planes are then only valid after
// pack_exact() has rebuilt pn/px.
// kAligned: every read starts on a dword of the tile (the stride is a multiple of 4): no byte realignment
template <int NW, bool kAligned>
BC_HD void pack_read(const uint32_t* t32, uint32_t base, uint32_t nd, Planes<NW>& P, uint32_t& bad) {
  const uint32_t a = base & 3u;
  const uint32_t* src = t32 + (base >> 2);
  // every LDS read is issued before the first use: the conversion never waits on one load at a time
  uint32_t raw[NW * 8 + 1];
#pragma unroll
  for (int i = 0; i <= NW * 8; ++i) raw[i] = src[i];
  bad = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    uint32_t o1 = 0, o2 = 0, on = 0;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      if ((uint32_t)(w * 8 + h * 2) < nd) {  // wave-uniform, per 8 bases
        const int idx = w * 8 + h * 2;
        const uint32_t d0 = kAligned ? raw[idx] : alignbyte(raw[idx + 1], raw[idx], a);
        // the pair's second dword may lie past the read (nd odd): its bits land past the read too
        const uint32_t d1 = kAligned ? raw[idx + 1] : alignbyte(raw[idx + 2 <= NW * 8 ? idx + 2 : idx + 1], raw[idx + 1], a);
        const uint32_t m = bitop3<0xE4>(d0 >> 1, d1 << 3, 0x07070707u);  // (a & c) | (b & ~c)
        const uint32_t g1 = udot4(m & 0x11111111u, 0x08040201u, 0u);      // plane-1 bits of the 8 bases
        const uint32_t g2 = udot4(m & 0x22222222u, 0x08040201u, 0u);      // plane-2 bits, times 2
        const uint32_t gn = udot4(m & 0x44444444u, 0x08040201u, 0u);      // N bits, times 4
        if (h == 0) {
          o1 = g1;
          o2 = g2 >> 1;
          on = gn >> 2;
        } else {
          o1 |= g1 << (8 * h);
          o2 |= g2 << (8 * h - 1);
          on |= gn << (8 * h - 2);
        }
        // validity: ACGTN rebuilt from the three bits must give the byte back
        const uint32_t e0 = perm(0x4EFFFFFFu, 0x47544341u, m & 0x07070707u);
        const uint32_t e1 = perm(0x4EFFFFFFu, 0x47544341u, (m >> 4) & 0x07070707u);
        bad = bitop3<0xF6>(bad, d0, e0);  // bad | (d0 ^ e0)
        if ((uint32_t)(idx + 1) < nd) bad = bitop3<0xF6>(bad, d1, e1);
      }
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
  const uint32_t* src = t32 + (base >> 2);
  uint32_t prev = src[0];
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    uint32_t ov = 0, on = 0, oh = 0;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      uint32_t av = 0, an = 0, ah = 0;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int idx = w * 8 + h * 2 + j;
        const uint32_t nxt = src[idx + 1];
        const uint32_t d = alignbyte(nxt, prev, a);
        prev = nxt;
        if ((uint32_t)(w * 8 + h * 2) < nd) {
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

// bits [off, off+len) of the three base planes (off/len wave-uniform, len <= 32): the word is
// picked by a scalar branch, so each plane costs one funnel shift and one mask
template <int NW>
BC_HD void extract_planes(const Planes<NW>& P, uint32_t off, uint32_t len, uint32_t& q1, uint32_t& q2, uint32_t& qn) {
  const uint32_t k = off >> 5, sh = off & 31u, m = lowmask(len);
  q1 = q2 = qn = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    if (k == (uint32_t)i) {
      q1 = alignbit(i + 1 < NW ? P.p1[i + 1] : 0u, P.p1[i], sh) & m;
      q2 = alignbit(i + 1 < NW ? P.p2[i + 1] : 0u, P.p2[i], sh) & m;
      qn = alignbit(i + 1 < NW ? P.pn[i + 1] : 0u, P.pn[i], sh) & m;
      BC_KEEP_BRANCH3(q1, q2, qn);
    }
  }
}

// base-5 code of a capture kept raw: A,C,T,G,N -> 0,1,2,3,4, base 0 least significant
BC_HD uint64_t base5_code(uint32_t q1, uint32_t q2, uint32_t qn, uint32_t len) {
  uint64_t code = 0;
  for (uint32_t i = len; i-- > 0;) {
    const uint32_t d = ((qn >> i) & 1u) ? 4u : (((q1 >> i) & 1u) | (((q2 >> i) & 1u) << 1));
    code = code * 5u + d;
  }
  return code;
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

template <int NW>
BC_HD void shr_to(uint32_t (&d)[NW], const uint32_t (&src)[NW], uint32_t sh) {  // sh in [0,31], wave-uniform
#pragma unroll
  for (int i = 0; i < NW; ++i) d[i] = alignbit(i + 1 < NW ? src[i + 1] : 0u, src[i], sh);
}

// Bit-sliced counters: cnt[b][w] holds bit b of the mismatch count of every window of word w;
// counts above 2^NB-1 are remembered in ovf.  add_bits adds, to every window at once, one vector
// of weight 1 (s) and one of weight 2 (c) -- the sum and carry of a 3:2 compression of three
// mismatch vectors, so three format positions cost little more than one.
template <int NWW, int NB>
struct Counters {
  uint32_t cnt[NB ? NB : 1][NWW];
  uint32_t ovf[NWW];
};

template <int NWW, int NB>
BC_HD void add_bits(Counters<NWW, NB>& C, int w, uint32_t s, uint32_t c) {
  if (NB == 0) {
    C.ovf[w] = bitop3<0xFE>(C.ovf[w], s, c);  // a | b | c
    return;
  }
  const uint32_t t = C.cnt[0][w] & s;
  C.cnt[0][w] ^= s;
  if (NB == 1) {
    C.ovf[w] = bitop3<0xFE>(C.ovf[w], c, t);
    return;
  }
  // weight 2: two incoming bits (c, t)
  const uint32_t c1 = C.cnt[1][w];
  C.cnt[1][w] = bitop3<kTT_Xor3>(c1, c, t);
  uint32_t carry = bitop3<kTT_Maj>(c1, c, t);
#pragma unroll
  for (int b = 2; b < NB - 1; ++b) {
    const uint32_t u = C.cnt[b][w] & carry;
    C.cnt[b][w] ^= carry;
    carry = u;
  }
  if (NB > 2) {
    C.ovf[w] = bitop3<kTT_AorBandC>(C.ovf[w], C.cnt[NB - 1][w], carry);
    C.cnt[NB - 1][w] ^= carry;
  } else {
    C.ovf[w] |= carry;
  }
}

// Mismatch count of every window against the constant bases of the format (the inner loops of
// fix_error, parse.rs:562-575, for all windows of fix_constant_region, parse.rs:291-304, at once).
// A read base equal to the format base, or 'N', is no mismatch (parse.rs:569).
template <int NW, int NWW, int NB, bool kAnyX>
BC_HD void count_mismatches(const DevPlan& pl, const Planes<NW>& P, Counters<NWW, NB>& C) {
#pragma unroll
  for (int w = 0; w < NWW; ++w) {
    C.ovf[w] = 0;
#pragma unroll
    for (int b = 0; b < (NB ? NB : 1); ++b) C.cnt[b][w] = 0;
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (pl.n_pos[c] == 0) continue;
    // v = "base equals letter c, or is 'N'" (N is free, parse.rs:569).  Bits past the read are
    // not masked: only windows that lie inside the read are ever candidates.
    uint32_t v[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      // f(p1, p2, pn) = pn | (p1 == c1 & p2 == c2): one instruction per word
      v[w] = c == 0 ? bitop3<0xAB>(P.p1[w], P.p2[w], P.pn[w])
           : c == 1 ? bitop3<0xBA>(P.p1[w], P.p2[w], P.pn[w])
           : c == 2 ? bitop3<0xAE>(P.p1[w], P.p2[w], P.pn[w])
                    : bitop3<0xEA>(P.p1[w], P.p2[w], P.pn[w]);
      if (kAnyX) v[w] &= ~P.px[w];  // a foreign byte never matches
    }
    const uint32_t* pp = pl.prog[c];
    const uint32_t n = pl.n3[c];
    uint32_t cur = pp[0];
    if (pl.prog_mode[c] == 0u) {
      // three positions per iteration, no branch inside: x, y, z are the match vectors at the
      // three positions, their complements the mismatch bits
      for (uint32_t i = 0; i < n; ++i) {
        const uint32_t nxt = pp[i + 1];
        uint32_t a[NW], b[NW];
        shr_to<NW>(a, v, cur & 31u);
        shr_to<NW>(b, a, (cur >> 8) & 31u);
        shr_to<NW>(v, b, (cur >> 16) & 31u);
#pragma unroll
        for (int w = 0; w < NWW; ++w) {
          add_bits<NWW, NB>(C, w, bitop3<kTT_Xnor3>(a[w], b[w], v[w]), bitop3<kTT_NotMaj>(a[w], b[w], v[w]));
        }
        cur = nxt;
      }
      const uint32_t k = cur >> 24;  // the positions left over: 0, 1 or 2
      if (k == 2u) {
        uint32_t a[NW], b[NW];
        shr_to<NW>(a, v, cur & 31u);
        shr_to<NW>(b, a, (cur >> 8) & 31u);
#pragma unroll
        for (int w = 0; w < NWW; ++w) add_bits<NWW, NB>(C, w, a[w] ^ b[w], bitop3<kTT_Nor2ab>(a[w], b[w], 0u));
      } else if (k == 1u) {
        uint32_t a[NW];
        shr_to<NW>(a, v, cur & 31u);
#pragma unroll
        for (int w = 0; w < NWW; ++w) add_bits<NWW, NB>(C, w, ~a[w], 0u);
      }
    } else {
      for (uint32_t i = 0; i < n; ++i) {
        const uint32_t e = pp[i];
        shr_uniform<NW>(v, e & 31u);
        if (e >> 24) {
#pragma unroll
          for (int w = 0; w < NWW; ++w) add_bits<NWW, NB>(C, w, ~v[w], 0u);
        }
      }
    }
  }
}

// ---- the same counts for a kernel specialised to its scheme (every format position a compile-time constant) --------
// calls f(p) for every format position p of class c, in ascending order (the decoding of DevPlan::prog)
template <class F>
BC_HD void for_each_position(const DevPlan& pl, int c, F&& f) {
  const uint32_t* pp = pl.prog[c];
  const uint32_t n = pl.n3[c];
  uint32_t pos = 0;
  if (pl.prog_mode[c] == 0u) {
#pragma unroll
    for (uint32_t i = 0; i < n; ++i) {
      const uint32_t e = pp[i];
      pos += e & 31u;
      f(pos);
      pos += (e >> 8) & 31u;
      f(pos);
      pos += (e >> 16) & 31u;
      f(pos);
    }
    const uint32_t e = pp[n], k = e >> 24;
    if (k >= 1u) {
      pos += e & 31u;
      f(pos);
    }
    if (k == 2u) {
      pos += (e >> 8) & 31u;
      f(pos);
    }
  } else {
#pragma unroll
    for (uint32_t i = 0; i < n; ++i) {
      const uint32_t e = pp[i];
      pos += e & 31u;
      if (e >> 24) f(pos);
    }
  }
}

// Carry-save accumulation of one-bit vectors: level l holds at most two pending vectors of weight 2^l; a third
// one turns the three into a sum (stays) and a carry (moves up) with two v_bitop3_b32 per word.  In a fully unrolled
// specialised kernel the fill counts are compile-time constants, so this is straight-line code: n vectors cost about
// n - log2(n) full adders instead of the n/3 * (2 + NB + 2) operations of the ripple counters above.
template <int NWW, int kLevels>
struct CsaCounter {
  uint32_t pend[kLevels][2][NWW];
  int fill[kLevels];
  uint32_t over[NWW];  // carries out of the top level
  BC_HD void init() {
#pragma unroll
    for (int l = 0; l < kLevels; ++l) fill[l] = 0;
#pragma unroll
    for (int w = 0; w < NWW; ++w) over[w] = 0;
  }
  BC_HD void add(const uint32_t (&x)[NWW]) {
    // (no array is indexed by a fill count and nothing returns early: every branch below folds away once the
    // caller's loops are unrolled, and until then there is nothing that would need scratch memory)
    uint32_t carry[NWW];
#pragma unroll
    for (int w = 0; w < NWW; ++w) carry[w] = x[w];
    bool live = true;
#pragma unroll
    for (int l = 0; l < kLevels; ++l) {
      if (!live) continue;
      if (fill[l] == 0) {
#pragma unroll
        for (int w = 0; w < NWW; ++w) pend[l][0][w] = carry[w];
        fill[l] = 1;
        live = false;
      } else if (fill[l] == 1) {
#pragma unroll
        for (int w = 0; w < NWW; ++w) pend[l][1][w] = carry[w];
        fill[l] = 2;
        live = false;
      } else {
#pragma unroll
        for (int w = 0; w < NWW; ++w) {
          const uint32_t a = pend[l][0][w], b = pend[l][1][w], c = carry[w];
          pend[l][0][w] = bitop3<kTT_Xor3>(a, b, c);
          carry[w] = bitop3<kTT_Maj>(a, b, c);
        }
        fill[l] = 1;
      }
    }
    if (live) {
#pragma unroll
      for (int w = 0; w < NWW; ++w) over[w] |= carry[w];
    }
  }
  // final bits: bit[l][w] = bit l of the count of every window; overflow past the top level in over[]
  BC_HD void finish(uint32_t (&bit)[kLevels][NWW]) {
    uint32_t carry[NWW];
#pragma unroll
    for (int w = 0; w < NWW; ++w) carry[w] = 0;
    bool have_carry = false;
#pragma unroll
    for (int l = 0; l < kLevels; ++l) {
      // the level's pending vectors plus the carry from below: up to three one-bit vectors
      const int n = fill[l] + (have_carry ? 1 : 0);
#pragma unroll
      for (int w = 0; w < NWW; ++w) {
        const uint32_t a = fill[l] >= 1 ? pend[l][0][w] : 0u;
        const uint32_t b = fill[l] >= 2 ? pend[l][1][w] : 0u;
        const uint32_t c = have_carry ? carry[w] : 0u;
        if (n <= 1) {
          bit[l][w] = a | c;
          carry[w] = 0;
        } else if (n == 2) {
          const uint32_t y = fill[l] >= 2 ? b : c;
          bit[l][w] = a ^ y;
          carry[w] = a & y;
        } else {
          bit[l][w] = bitop3<kTT_Xor3>(a, b, c);
          carry[w] = bitop3<kTT_Maj>(a, b, c);
        }
      }
      have_carry = n >= 2;
    }
    if (have_carry) {
#pragma unroll
      for (int w = 0; w < NWW; ++w) over[w] |= carry[w];
    }
  }
};

// n_cand: only windows 0 .. n_cand-1 are ever looked at (the read length is part of the specialisation), so a window
// word may hold junk above them -- which lets the last window word of most positions be a plain shift.
template <int NW, int NWW, int NB, bool kAnyX>
BC_HD void count_mismatches_static(const DevPlan& pl, const Planes<NW>& P, uint32_t n_cand, Counters<NWW, NB>& C) {
  constexpr int kLevels = 9;  // counts below 512: more constant positions than a 320-base read can hold
  CsaCounter<NWW, kLevels> acc;
  acc.init();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (pl.n_pos[c] == 0) continue;
    // u = "base is neither letter c nor 'N'": the mismatch indicator (N is free, parse.rs:569); zero past the read's
    // words, and whatever the padding bytes give inside the last word -- only windows inside the read are candidates
    uint32_t u[NW + NWW + 1];
#pragma unroll
    for (int w = 0; w < NW + NWW + 1; ++w) {
      if (w >= NW) {
        u[w] = 0;
        continue;
      }
      // the complement of f(p1, p2, pn) = pn | (p1 == c1 & p2 == c2)
      u[w] = c == 0 ? bitop3<0xFF ^ 0xAB>(P.p1[w], P.p2[w], P.pn[w])
           : c == 1 ? bitop3<0xFF ^ 0xBA>(P.p1[w], P.p2[w], P.pn[w])
           : c == 2 ? bitop3<0xFF ^ 0xAE>(P.p1[w], P.p2[w], P.pn[w])
                    : bitop3<0xFF ^ 0xEA>(P.p1[w], P.p2[w], P.pn[w]);
      if (kAnyX) u[w] |= P.px[w];  // a foreign byte never matches
    }
    for_each_position(pl, c, [&](uint32_t pos) {
      const uint32_t k = pos >> 5, sh = pos & 31u;
      uint32_t x[NWW];
      if (sh == 0u) {
#pragma unroll
        for (int w = 0; w < NWW; ++w) x[w] = u[k + w];
      } else if (NWW == 1 && sh + n_cand <= 32u) {
        x[0] = u[k] >> sh;
      } else if (NWW == 2 && sh + n_cand <= 64u) {
        x[0] = alignbit(u[k + 1], u[k], sh);
        x[1] = u[k + 1] >> sh;  // its top sh bits are junk no candidate window sees: a plain shift will do
      } else {
#pragma unroll
        for (int w = 0; w < NWW; ++w) x[w] = alignbit(u[k + w + 1], u[k + w], sh);
      }
      acc.add(x);
    });
  }
  uint32_t bit[kLevels][NWW];
  acc.finish(bit);
#pragma unroll
  for (int w = 0; w < NWW; ++w) {
    C.ovf[w] = acc.over[w];
#pragma unroll
    for (int l = 0; l < kLevels; ++l) {
      if (l < NB)
        C.cnt[l][w] = bit[l][w];
      else
        C.ovf[w] |= bit[l][w];
    }
    if (NB == 0) C.cnt[0][w] = 0;
  }
}

// AND of the class vector over the class's positions, for every window (scheme-N positions)
template <int NW, int NWW>
BC_HD void and_program(const DevPlan& pl, int c, uint32_t (&v)[NW], uint32_t (&acc)[NWW]) {
  const uint32_t* pp = pl.prog[c];
  const uint32_t n = pl.n3[c];
  const bool singles = pl.prog_mode[c] != 0u;
  for (uint32_t i = 0; i <= n; ++i) {
    if (singles && i == n) break;
    const uint32_t e = pp[i];
    // triples mode: three positions per entry, then the k left over; singles mode: k in the entry
    const uint32_t steps = singles ? 1u : (i < n ? 3u : (e >> 24));
    const bool use = singles ? (e >> 24) != 0u : true;
    for (uint32_t t = 0; t < steps; ++t) {
      shr_uniform<NW>(v, (e >> (8 * t)) & 31u);
      if (use) {
#pragma unroll
        for (int w = 0; w < NWW; ++w) acc[w] &= v[w];
      }
    }
  }
}

// Anchor + repair for one read.  Returns true when a construct was located: `start` is its
// offset and `repaired` says whether the constant region had to be repaired.
//  * exact anchor (Regex::is_match / captures, parse.rs:92-95, 153-156): the leftmost window with
//    no mismatch, no 'N' on a constant position, and valid bases on the scheme-N positions;
//  * otherwise fix_constant_region (parse.rs:287-313): among the windows 0 .. len-L-1 (the last
//    window is never tested, parse.rs:291-295) the unique minimum-mismatch window within the
//    budget (fix_error, parse.rs:577-592), provided its scheme-N positions are valid bases.
// kAnyX: some lane of the wave holds a byte outside ACGTN (a separate, rarely run instantiation, so that the common
// one carries no trace of the px plane)
template <class Ops, int NW, int NWW, int NB, bool kAnyX>
BC_HD bool locate(const DevPlan& pl, Ops& ops, const Planes<NW>& P, const uint32_t (&inr)[NW], uint32_t len, bool live,
                  uint32_t& start, bool& repaired) {
  const uint32_t L = pl.L;
  Counters<NWW, NB> C;
  // BC_NO_STATIC_COUNT: the engine's second try for a scheme whose straight-line form the compiler did not manage
  // to keep in registers (bc_engine.hip jit_function)
#if (defined(BC_JIT_TU) || defined(BC_EMU_STATIC_COUNT)) && !defined(BC_NO_STATIC_COUNT)
  // candidate windows of this kernel's batch shape: 0 .. read_len - L for fixed-length reads, any window otherwise
  count_mismatches_static<NW, NWW, NB, kAnyX>(pl, P, BC_STATIC_CAND(L, NWW), C);
#else
  count_mismatches<NW, NWW, NB, kAnyX>(pl, P, C);
#endif

  uint32_t fnok[NWW];
#pragma unroll
  for (int w = 0; w < NWW; ++w) fnok[w] = 0xFFFFFFFFu;
  if (pl.has_fmtn) {  // [AGCT]{n}, info.rs:291-294
    uint32_t v[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) v[w] = kAnyX ? (inr[w] & ~(P.pn[w] | P.px[w])) : (inr[w] & ~P.pn[w]);
    and_program<NW, NWW>(pl, kClassFmtN, v, fnok);
  }

  // windows without any mismatch ('N' still free)
  uint32_t z[NWW];
  low_bits<NWW>(z, len >= L ? len - L + 1u : 0u);
  uint32_t anyn = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) anyn |= P.pn[w];
#pragma unroll
  for (int w = 0; w < NWW; ++w) {
    z[w] &= ~C.ovf[w] & fnok[w];
#pragma unroll
    for (int b = 0; b < NB; ++b) z[w] &= ~C.cnt[b][w];
  }
  bool found = false;
  start = 0;
  repaired = false;
  for (;;) {
    uint32_t o = 0, any = 0;
#pragma unroll
    for (int w = NWW - 1; w >= 0; --w) {
      if (z[w]) {
        o = 32u * (uint32_t)w + ctz(z[w]);
        any = 1;
      }
    }
    if (!any) break;
    bool nfree = true;
    if (anyn) {  // the regex needs the literal base: an 'N' on a constant position is no match
      uint32_t t[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) t[w] = P.pn[w];
      shr_lane<NW>(t, o);
      uint32_t hit = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) hit |= t[w] & pl.cmask[w];
      nfree = hit == 0u;
    }
    if (nfree) {
      found = true;
      start = o;
      break;
    }
#pragma unroll
    for (int w = 0; w < NWW; ++w)
      if ((o >> 5) == (uint32_t)w) z[w] &= ~(1u << (o & 31u));
  }

  if (!pl.no_repair && !(pl.abl() & 0x1u) && ops.any(live && !found)) {
    uint32_t cand[NWW];
    low_bits<NWW>(cand, len > L ? len - L : 0u);
#pragma unroll
    for (int w = 0; w < NWW; ++w) cand[w] &= ~C.ovf[w];
#pragma unroll
    for (int b = NB - 1; b >= 0; --b) {
      uint32_t t[NWW];
      uint32_t nz = 0;
#pragma unroll
      for (int w = 0; w < NWW; ++w) {
        t[w] = cand[w] & ~C.cnt[b][w];
        nz |= t[w];
      }
#pragma unroll
      for (int w = 0; w < NWW; ++w) cand[w] = nz ? t[w] : cand[w];
    }
    uint32_t n_min = 0, val = 0, pos = 0;
#pragma unroll
    for (int w = NWW - 1; w >= 0; --w) {
      n_min += popc(cand[w]);
      if (cand[w]) pos = 32u * (uint32_t)w + ctz(cand[w]);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      uint32_t o = 0;
#pragma unroll
      for (int w = 0; w < NWW; ++w) o |= cand[w] & C.cnt[b][w];
      val |= (o ? 1u : 0u) << b;
    }
    // the window replaces the read, constants overwritten by the format
    // (insert_barcodes_constant_region, parse.rs:270-283); the regex then has to match it at
    // offset 0, which only the scheme-N positions can still prevent
    if (!found && n_min == 1u && val <= pl.max_const && test_bit<NWW>(fnok, pos)) {
      found = true;
      repaired = true;
      start = pos;
    }
  }
  return found;
}

// counter width is a compile-time constant; any width whose range covers max_const (or none at all when no mismatch
// is allowed) gives the same verdicts
template <class Ops, int NW, int NWW, bool kAnyX>
BC_HD bool locate_nb(const DevPlan& pl, Ops& ops, const Planes<NW>& P, const uint32_t (&inr)[NW], uint32_t len, bool live,
                     uint32_t& start, bool& repaired) {
  if (pl.max_const == 0u) return locate<Ops, NW, NWW, 0, kAnyX>(pl, ops, P, inr, len, live, start, repaired);
  if (pl.nb <= 2u) return locate<Ops, NW, NWW, 2, kAnyX>(pl, ops, P, inr, len, live, start, repaired);
  if (pl.nb == 3u) return locate<Ops, NW, NWW, 3, kAnyX>(pl, ops, P, inr, len, live, start, repaired);
  return locate<Ops, NW, NWW, 5, kAnyX>(pl, ops, P, inr, len, live, start, repaired);
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

// The same sum for the common case that no byte is below '!' (33), in two v_sad_u8 per dword: sum |b - 33| equals
// sum (b - 33) exactly when every b >= 33, and is larger otherwise -- so ok = false tells the caller to take score_sum
// (bytes below 33 wrap in the reference's release build, parse.rs:326; a well-formed FASTQ has none).
BC_HD uint32_t score_sum_fast(const uint32_t* q32, uint32_t addr, uint32_t len, bool& ok) {
  const uint32_t a = addr & 3u;
  const uint32_t k0 = addr >> 2;
  const uint32_t nd = (len + 3u) >> 2;  // wave-uniform
  uint32_t prev = q32[k0];
  uint32_t s_abs = 0, s_raw = 0;
  for (uint32_t j = 0; j < nd; ++j) {
    const uint32_t nxt = q32[k0 + j + 1];
    uint32_t d = alignbyte(nxt, prev, a);
    prev = nxt;
    if (j == nd - 1 && (len & 3u)) {  // bytes past the run count as '!' = score 0
      const uint32_t m = lowmask(8u * (len & 3u));
      d = (d & m) | (0x21212121u & ~m);
    }
    s_abs = sad_u8(d, 0x21212121u, s_abs);
    s_raw = sad_u8(d, 0u, s_raw);
  }
  ok = s_raw == s_abs + 33u * 4u * nd;
  return s_abs;
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

// one entry of a kSetDirect correction table (layout: bc_device_plan.h)
BC_HD uint32_t dtable_entry(const DevGroup& G, uint32_t q) {
  const uint32_t q1 = q & lowmask(G.len), q2 = q >> G.len;
  Nearest s;
  nearest_init(s);
  for (uint32_t j = 0; j < G.n_refs; ++j) {
    bool ex;
    const uint32_t d = ref_distance(q1, q2, 0u, 0u, G.len, G.r1()[j], G.r2()[j], G.rn()[j], G.rlen()[j], ex);
    nearest_add(s, d, j, ex);
  }
  const uint32_t r = nearest_result(s.key, s.idx, s.count, G.max_err);
  uint32_t dmin = s.key == 0u ? 0u : s.key - 1u;
  if (dmin > 255u) dmin = 255u;
  return (r == kFail ? (uint32_t)kFail16 : r) | (dmin << 16) | ((s.count == 1u ? 1u : 0u) << 24);
}

// A capture with exactly one 'N' (and nothing else unusual) against a set of plain, equal-length
// references: 'N' is free (parse.rs:569), so its distance to a reference r is the distance of
// the capture with N replaced by r's base there.  Hence the nearest references of the capture are
// the nearest references of its four substitutions at the smallest of their four minimum
// distances D; the match is unique iff exactly one substitution reaches D and does so uniquely.
// (table, len, max_err may differ from lane to lane)
BC_HD uint32_t single_n_lookup(const BC_GLOBAL uint32_t* table, uint32_t len, uint32_t max_err, uint32_t q1, uint32_t q2,
                               uint32_t qn) {
  const uint32_t k = ctz(qn);
  const uint32_t b1 = q1 & ~qn, b2 = q2 & ~qn;
  uint32_t best = 256u, res = kFail;
  bool ok = false;
  uint32_t tt[4];
#pragma unroll
  for (uint32_t b = 0; b < 4; ++b) tt[b] = table[(b1 | ((b & 1u) << k)) | ((b2 | ((b >> 1) << k)) << len)];
#pragma unroll
  for (uint32_t b = 0; b < 4; ++b) {
    const uint32_t t = tt[b];
    const uint32_t d = (t >> 16) & 0xFFu;
    if (d < best) {
      best = d;
      ok = ((t >> 24) & 1u) != 0u;
      res = t & 0xFFFFu;
    } else if (d == best) {
      ok = false;
    }
  }
  return (ok && best <= max_err && res != (uint32_t)kFail16) ? res : kFail;
}

// LDS exact-match table in front of dtable (bc_device_plan.h): the reference index of a capture
// that is a stored reference, kFail for everything else
struct alignas(16) Quad {
  uint32_t x, y, z, w;
};
// `off`, `shift`, `len` may differ from lane to lane
BC_HD uint32_t lhash_probe(const Quad* __restrict__ area, uint32_t off, uint32_t nb, uint32_t len, uint32_t key) {
  const uint32_t ibits = 32u - 2u * len;
  const Quad e = area[off + lhash_bucket(key, kLhashMul1, nb)];
  const Quad f = area[off + lhash_bucket(key, kLhashMul2, nb)];
  const uint32_t kk = key << ibits;
  // an entry of this key leaves only its index after the xor; every other entry keeps a high bit
  const uint32_t m = min3u(min3u(e.x ^ kk, e.y ^ kk, e.z ^ kk), min3u(e.w ^ kk, f.x ^ kk, f.y ^ kk),
                           min3u(f.z ^ kk, f.w ^ kk, 0xFFFFFFFFu));
  return (m >> ibits) == 0u ? m : kFail;
}
BC_HD uint32_t lhash_lookup(const Quad* __restrict__ area, const DevGroup& G, uint32_t key) {
  return lhash_probe(area, G.lhash_off, G.lhash_nb, G.len, key);
}

// A capture with one 'N' against a complete LDS table: the reference it stands for when exactly one
// of its four substitutions is a reference (distance 0, unique); kFail with settled = true when two
// or more are (a tie at distance 0) or when no mismatch is allowed; otherwise settled = false -- the
// nearest references are one or more mismatches away and dtable has to be asked (single_n_lookup).
BC_HD uint32_t single_n_lhash(const Quad* __restrict__ area, uint32_t off, uint32_t nb, uint32_t len, uint32_t max_err,
                              uint32_t q1, uint32_t q2, uint32_t qn, bool& settled) {
  const uint32_t k = ctz(qn);
  const uint32_t b1 = q1 & ~qn, b2 = q2 & ~qn;
  const uint32_t ibits = 32u - 2u * len;
  // all eight bucket reads are issued before the first is looked at: one LDS round trip
  uint32_t key[4];
  Quad e[4], f[4];
#pragma unroll
  for (uint32_t b = 0; b < 4; ++b) {
    key[b] = (b1 | ((b & 1u) << k)) | ((b2 | ((b >> 1) << k)) << len);
    e[b] = area[off + lhash_bucket(key[b], kLhashMul1, nb)];
    f[b] = area[off + lhash_bucket(key[b], kLhashMul2, nb)];
  }
  uint32_t hits = 0, res = kFail;
#pragma unroll
  for (uint32_t b = 0; b < 4; ++b) {
    const uint32_t kk = key[b] << ibits;
    const uint32_t m = min3u(min3u(e[b].x ^ kk, e[b].y ^ kk, e[b].z ^ kk), min3u(e[b].w ^ kk, f[b].x ^ kk, f[b].y ^ kk),
                             min3u(f[b].z ^ kk, f[b].w ^ kk, 0xFFFFFFFFu));
    const bool hit = (m >> ibits) == 0u;
    hits += hit ? 1u : 0u;
    res = hit ? m : res;
  }
  settled = hits != 0u || max_err == 0u;
  return hits == 1u ? res : kFail;
}

// First tier of the search in a large set (bc_device_plan.h): every reference within one mismatch of
// the N-free capture (q1, q2) sits in one of its two buckets, so a lane that finds one there knows the
// minimum distance (0 or 1) and how many references have it.  settled = false: nothing that near --
// the pigeonhole search over all budget+1 blocks has to decide.
struct TierLines {
  uint32_t val[2];
  Quad v[2][4];  // the two buckets' heads as loaded (the compact form fills two quads of each)
};
// both buckets' heads (one line each) are requested before either is looked at: one round trip.  Whole quads with
// fixed positions: a per-dword loop over a length only known at run time ends up in scratch memory.
template <uint32_t kFirst, uint32_t kLast>
BC_HD void tier_fetch_blocks(const DevGroup& G, uint32_t q1, uint32_t q2, TierLines& t) {
  const uint32_t blen = G.tier_blen, bm = (1u << blen) - 1u, nbk = 1u << (2u * blen);
  const BC_GLOBAL Quad* bkt = reinterpret_cast<const BC_GLOBAL Quad*>(G.tier_bkt());
#pragma unroll
  for (uint32_t b = kFirst; b <= kLast; ++b) {
    const uint32_t sh = b * G.tier_stride;
    t.val[b] = ((q1 >> sh) & bm) | (((q2 >> sh) & bm) << blen);
    if (G.tier_compact) {
      const BC_GLOBAL Quad* line = bkt + ((size_t)b * nbk + t.val[b]) * 2u;
      t.v[b][0] = line[0];
      t.v[b][1] = line[1];
    } else {
      const BC_GLOBAL Quad* line = bkt + ((size_t)b * nbk + t.val[b]) * 4u;
      t.v[b][0] = line[0];
      t.v[b][1] = line[1];
      t.v[b][2] = line[2];
      t.v[b][3] = line[3];
    }
  }
}
BC_HD void tier_fetch(const DevGroup& G, uint32_t q1, uint32_t q2, TierLines& t) { tier_fetch_blocks<0, 1>(G, q1, q2, t); }
// one entry of a bucket head: {r1, r2, index}; n = references in the bucket (valid for entry 0)
template <uint32_t k>
BC_HD void tier_entry(const DevGroup& G, const Quad* v, uint32_t& r1, uint32_t& r2, uint32_t& j, uint32_t& n) {
  if (G.tier_compact) {
    // 32 bytes: four entries of r1 | r2 << len | index << 2 len, the count in the first one's top three bits
    const uint32_t len = G.len, lm = lowmask(len);
    const Quad& h = v[k >> 1];
    const uint32_t lo = (k & 1u) ? h.z : h.x, hi = (k & 1u) ? h.w : h.y;
    const uint64_t x = ((uint64_t)hi << 32) | lo;
    r1 = (uint32_t)x & lm;
    r2 = (uint32_t)(x >> len) & lm;
    j = (uint32_t)((x & 0x1FFFFFFFFFFFFFFFull) >> (2u * len));
    n = hi >> 29;  // (capped at 7, which is all the callers' tests need)
  } else {
    r1 = v[k].x;
    r2 = v[k].y;
    j = v[k].z;
    n = v[k].w;
  }
}
BC_HD void tier_score(const DevGroup& G, uint32_t q1, uint32_t q2, const TierLines& t, uint32_t& best, uint32_t& cnt,
                      uint32_t& idx) {
  const uint32_t blen = G.tier_blen, bm = (1u << blen) - 1u, nbk = 1u << (2u * blen);
  best = 0xFFFFFFFFu;
  cnt = 0;
  idx = kFail;
  auto score = [&](uint32_t r1, uint32_t r2, uint32_t j, bool second_block, bool on) {
    const uint32_t diff = (q1 ^ r1) | (q2 ^ r2);
    // a reference that equals the capture on block 0 too was met there
    const bool use = on && !(second_block && (diff & bm) == 0u);
    const uint32_t d = popc(diff);
    if (use && d < best) {
      best = d;
      cnt = 1;
      idx = j;
    } else if (use && d == best) {
      ++cnt;
    }
  };
#pragma unroll
  for (uint32_t b = 0; b < 2; ++b) {
    uint32_t n, r1, r2, j, nk;
    tier_entry<0>(G, t.v[b], r1, r2, j, n);
    score(r1, r2, j, b == 1u, 0u < n);
    tier_entry<1>(G, t.v[b], r1, r2, j, nk);
    score(r1, r2, j, b == 1u, 1u < n);
    tier_entry<2>(G, t.v[b], r1, r2, j, nk);
    score(r1, r2, j, b == 1u, 2u < n);
    tier_entry<3>(G, t.v[b], r1, r2, j, nk);
    score(r1, r2, j, b == 1u, 3u < n);
    if (n > 4u) {  // the rest of a long bucket
      const BC_GLOBAL uint32_t* off = G.tier_off() + (size_t)b * (nbk + 1u);
      const BC_GLOBAL uint32_t* list = G.tier_list() + (size_t)b * G.n_idx * 4u;
      for (uint32_t i = off[t.val[b]] + 4u, end = off[t.val[b] + 1u]; i < end; ++i)
        score(list[i * 4u], list[i * 4u + 1u], list[i * 4u + 2u], b == 1u, true);
    }
  }
}
// The capture itself among the (at most four) references of its block-0 bucket: a reference at distance 0 is the
// unique nearest one, whatever the other block's bucket holds.  False for a longer bucket or a reference listed twice:
// tier_score decides those.
BC_HD bool tier_exact_in_block0(const DevGroup& G, uint32_t q1, uint32_t q2, const TierLines& t, uint32_t& idx) {
  uint32_t n, r1, r2, j, nk, hits = 0;
  idx = kFail;
  tier_entry<0>(G, t.v[0], r1, r2, j, n);
  if (0u < n && ((q1 ^ r1) | (q2 ^ r2)) == 0u) {
    ++hits;
    idx = j;
  }
  tier_entry<1>(G, t.v[0], r1, r2, j, nk);
  if (1u < n && ((q1 ^ r1) | (q2 ^ r2)) == 0u) {
    ++hits;
    idx = j;
  }
  tier_entry<2>(G, t.v[0], r1, r2, j, nk);
  if (2u < n && ((q1 ^ r1) | (q2 ^ r2)) == 0u) {
    ++hits;
    idx = j;
  }
  tier_entry<3>(G, t.v[0], r1, r2, j, nk);
  if (3u < n && ((q1 ^ r1) | (q2 ^ r2)) == 0u) {
    ++hits;
    idx = j;
  }
  return hits == 1u && n <= 4u;
}
BC_HD void tier_probe(const DevGroup& G, uint32_t q1, uint32_t q2, uint32_t& best, uint32_t& cnt, uint32_t& idx) {
  TierLines t;
  tier_fetch(G, q1, q2, t);
  tier_score(G, q1, q2, t, best, cnt, idx);
}

BC_HD uint32_t tier_lookup(const DevGroup& G, uint32_t q1, uint32_t q2, bool& settled) {
  uint32_t best, cnt, idx;
  tier_probe(G, q1, q2, best, cnt, idx);
  settled = best <= 1u;
  return (settled && cnt == 1u && best <= G.max_err) ? idx : kFail;
}

// The same for a capture with one 'N': 'N' is free (parse.rs:569), so a reference's distance to the
// capture is its distance to the substitution that carries the reference's own base there -- every
// reference shows up at its true distance under exactly one of the four substitutions.  Hence the
// minimum over the four probes is the minimum distance, and the references at it are the probes' counts
// added up.
BC_HD uint32_t tier_lookup_single_n(const DevGroup& G, uint32_t q1, uint32_t q2, uint32_t qn, bool& settled) {
  const uint32_t k = ctz(qn);
  const uint32_t b1 = q1 & ~qn, b2 = q2 & ~qn;
  uint32_t best = 0xFFFFFFFFu, total = 0, res = kFail;
  for (uint32_t b = 0; b < 4; ++b) {
    uint32_t sb, sc, si;
    tier_probe(G, b1 | ((b & 1u) << k), b2 | ((b >> 1) << k), sb, sc, si);
    if (sb < best) {
      best = sb;
      total = sc;
      res = si;
    } else if (sb == best) {
      total += sc;
    }
  }
  settled = best <= 1u;
  return (settled && total == 1u && best <= G.max_err) ? res : kFail;
}

// ---- the per-read decision tree ---------------------------------------------------------------
struct ReadResult {
  uint32_t outcome;    // Outcome; kMatched means "passed every test" (duplicate detection is later)
  uint64_t dense_idx;  // index into the dense (sample, tuple) counter table
  uint64_t rcode;      // base-5 code of the random barcode (0 without one)
};

// Ops must provide:
//   bool any(bool)                               -- wave vote
//   uint32_t nearest(const DevGroup&, q1,q2,qn,qx, bool need) -- cooperative Hamming search, every lane calls it
//   uint32_t tier_single_n(const DevGroup&, q1,q2,qn, bool want, bool& settled) -- tier_lookup_single_n of the
//                                                   lanes that want it, every lane calls it
//   void sequence_consumed()                     -- called once, by every lane, after the last read of the
//                                                   sequence bytes (the GPU starts fetching the next tile)
//   void issued()                                -- loads written above this call are issued before anything below
//   void mark(int)                               -- profiling hook (no-op outside BC_PROFILE builds)
//   const Quad* lhash(), bool tables()           -- the LDS exact-match area (plan.lhash_vec uint4s), and whether
//                                                   it is loaded
//   const uint32_t* stage_quality(uint32_t n)    -- called once, by every lane, when the quality filter is
//                                                   on: returns where the quality lines are; n = loads of its
//                                                   own the lane code still has in flight
// NW = 32-base words per read; NWW = words of candidate offsets / repair windows (len - L + 1 <= 32*NWW)
// kAligned: base is a multiple of 4 for every lane (the stride is)
template <class Ops, int NW, int NWW, bool kAligned = false>
// len / qlen: length of the sequence line / of the quality line (they differ in trimmed or damaged files)
BC_HD ReadResult process_read(const DevPlan& pl, Ops& ops, const uint32_t* seq32, uint32_t base, uint32_t len,
                              uint32_t qlen, uint32_t nd, bool active) {
  ReadResult res;
  res.outcome = kMatched;
  res.dense_idx = 0;
  res.rcode = 0;

  Planes<NW> P;
  uint32_t bad;
  if (pl.abl() & 0x40u) {
    bad = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { P.p1[w] = base * 2654435761u + w; P.p2[w] = base * 40503u + w; P.pn[w] = 0; P.px[w] = 0; }
  } else {
    pack_read<NW, kAligned>(seq32, base, nd, P, bad);
  }
  uint32_t inr[NW];
  low_bits<NW>(inr, len);
  bool unsupported = false;
  uint32_t pending_n = 0;  // 'N' mask of a capture handed back for a later search (kPending)
  const bool anyx = ops.any(active && bad != 0u);
  if (anyx) {
    uint32_t hi[NW];
    pack_exact<NW>(seq32, base, nd, P, hi);
    uint32_t h = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) h |= hi[w] & inr[w];
    unsupported = h != 0u;  // non-ASCII: the reference's char positions no longer equal byte positions
  }
  ops.mark(2);
  ops.sequence_consumed();
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    P.pn[w] &= inr[w];
    P.px[w] &= inr[w];
  }

  // ---- locate the construct: exact anchor, else constant-region repair -----------------------------
  uint32_t start = 0;
  bool repaired = false, found = false;
  {
    const bool live = active && !unsupported;
    // counter width is a compile-time constant; any width whose range covers max_const (or none
    // at all when no mismatch is allowed) gives the same verdicts
    if (pl.abl() & 0x10u) {
      found = true;
    } else if (anyx) {
      found = locate_nb<Ops, NW, NWW, true>(pl, ops, P, inr, len, live, start, repaired);
    } else {
      found = locate_nb<Ops, NW, NWW, false>(pl, ops, P, inr, len, live, start, repaired);
    }
  }

  uint32_t outcome = kMatched;
  if (!found) outcome = kConstantRegion;  // parse.rs:145
  ops.mark(3);

  // ---- quality filter (parse.rs:98-119, 331-375) ------------------------------------------------
  // In the reference it comes before the barcodes are looked at; here it runs while the first group
  // chunk's table gathers are in flight.  The outcome only depends on the priority of the tests
  // (constant region, quality, sample barcode, counted barcodes), which the order below keeps.
  const uint32_t start_found = start;
  bool quality_done = false;
  // in_flight: vector-memory loads this lane code has issued and not yet consumed (they are younger than
  // the quality fetch and need not land for it)
  auto quality_filter = [&](uint32_t in_flight) {
    quality_done = true;
    if (pl.quality_on && !(pl.abl() & 0x8u)) {
      const uint32_t* qual32 = ops.stage_quality(in_flight);
      ops.mark(4);
      // after a repair the quality line is read from offset 0
      const uint32_t qstart = repaired ? 0u : start_found;
      // scores.iter().skip(start).zip(regions) (parse.rs:340-345): what is left of the QUALITY line after the skip
      const uint32_t avail = qlen > qstart ? qlen - qstart : 0u;
      const uint32_t zip = avail < pl.RL ? avail : pl.RL;
      bool low = false, plain = true;
      for (uint32_t r = 0; r < pl.n_runs; ++r) {
        const uint32_t ro = pl.run_off[r], rl = pl.run_len[r];
        // a run is only evaluated when the zip continues past it (parse.rs:348-356)
        const bool evaluated = (ro + rl) < zip;
        bool ok;
        const uint32_t sum = score_sum_fast(qual32, base + (found ? qstart : 0u) + ro, rl, ok);
        plain = plain && ok;
        low = low || (evaluated && sum < pl.run_thr[r]);
      }
      if (ops.any(!plain)) {  // some quality byte below '!': the wrapping form, every run again
        low = false;
        for (uint32_t r = 0; r < pl.n_runs; ++r) {
          const uint32_t ro = pl.run_off[r], rl = pl.run_len[r];
          const bool evaluated = (ro + rl) < zip;
          const uint32_t sum = score_sum(qual32, base + (found ? qstart : 0u) + ro, rl);
          low = low || (evaluated && sum < pl.run_thr[r]);
        }
      }
      if (outcome == kMatched && low) outcome = kLowQuality;  // parse.rs:111
    }
    ops.mark(5);
  };

  // ---- barcodes: SequenceMatchResult::new (parse.rs:439-524) -----------------------------------
  // bring the construct to bit 0 so that every capture sits at a wave-uniform position
  if (!found) start = 0;
  shr_lane<NW>(P.p1, start);
  shr_lane<NW>(P.p2, start);
  shr_lane<NW>(P.pn, start);
  if (anyx) shr_lane<NW>(P.px, start);
  uint64_t didx = 0;
  bool raw_foreign = false;
  const bool located = active && outcome == kMatched;  // anchored; the quality verdict is still to come
  const uint32_t ng = (pl.abl() & 0x20u) ? 0u : pl.n_groups;
  // four groups at a time: first every capture is cut out and looked up (LDS, then the table gathers
  // of what LDS could not answer), then the verdicts are consumed in order
  for (uint32_t g0 = 0; g0 < ng; g0 += 4) {
    uint32_t q1[4], q2[4], qn[4], qx[4], r[4], tl[4];
    // per-lane flags of the four groups, one bit each.  Packed words (vector registers) rather than
    // bools: a bool that lives across the phase is a wave-wide mask in two scalar registers, and the
    // kernel is short of those.
    uint32_t need_m = 0, gather_m = 0, gather_n_m = 0;
    uint32_t pend_n = 0;  // groups (bit u) whose capture holds exactly one 'N' and has a complete LDS table
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      r[u] = kFail;
      tl[u] = kFail;
      q1[u] = q2[u] = qn[u] = qx[u] = 0;
      if (g0 + u < ng) {  // wave-uniform
        const DevGroup& G = pl.groups[g0 + u];
        extract_planes<NW>(P, G.off, G.len, q1[u], q2[u], qn[u]);
        if (anyx) qx[u] = extract_uniform<NW>(P.px, G.off, G.len);
      }
    }
    // LDS exact-match lookups of all four captures, every lane, no branches: the eight bucket reads
    // share one round trip (a capture with 'N's simply finds nothing useful)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (g0 + u < ng) {
        const DevGroup& G = pl.groups[g0 + u];
        if (G.mode == kSetDirect && G.lhash_nb && ops.tables())
          tl[u] = lhash_lookup(ops.lhash(), G, q1[u] | (q2[u] << G.len));
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (g0 + u < ng) {  // wave-uniform
        const DevGroup& G = pl.groups[g0 + u];
        if (G.mode != kSetNone) {
          const bool clean = (qn[u] | qx[u]) == 0u;
          if (located) {
            if (G.mode == kSetDirect) {
              const bool use_lds = G.lhash_nb && ops.tables();
              if (clean) {
                // a capture that is a reference is answered from LDS; the others go to the table, unless
                // no mismatch is allowed: then "not a reference" is already the verdict
                r[u] = tl[u];
                gather_m |= (r[u] == kFail && !(use_lds && G.lhash_complete && G.max_err == 0u)) ? (1u << u) : 0u;
              } else if (qx[u] == 0u && !G.has_odd && (qn[u] & (qn[u] - 1u)) == 0u && !(pl.abl() & 0x100u)) {
                if (use_lds && G.lhash_complete)
                  pend_n |= 1u << u;  // settled in LDS below, all groups in one pass
                else
                  gather_n_m |= 1u << u;
              } else {
                need_m |= 1u << u;
              }
            } else if (G.mode != kSetHash) {
              need_m |= 1u << u;
            }
          }
        }
        if (G.mode == kSetHash) {  // wave-uniform: the lanes of the wave work together below
          const bool clean = (qn[u] | qx[u]) == 0u;
          const bool tier = G.tier_blen && !(pl.abl() & 0x800u);
          const bool probe = located && clean;
          const bool one_n = located && !clean && tier && qx[u] == 0u && (qn[u] & (qn[u] - 1u)) == 0u && !(pl.abl() & 0x4000u);
          need_m |= located ? (1u << u) : 0u;
          if (!clean && (pl.abl() & 0x4000u)) need_m &= ~(1u << u);  // experiment: captures with 'N' simply fail
          if (tier) {
            // the one-mismatch tier also finds the capture itself (distance 0): no separate exact lookup.  The
            // plain captures' lines are requested first; the rare captures with one 'N' are then settled by the
            // whole wave, their loads sharing the round trip (moving the pass behind the second fetch: no gain)
            TierLines t;
#ifdef BC_TIER_BOTH
            if (probe) tier_fetch(G, q1[u], q2[u], t);
#else
            // block 0's line first: four reads in five carry a reference unchanged, which that line alone proves --
            // the other half of the tier is then touched by one read in five only and leaves the L2 to the first
            if (probe) tier_fetch_blocks<0, 0>(G, q1[u], q2[u], t);
#endif
            ops.issued();  // (the compiler would otherwise move the loads down to their use, behind the pass)
            if (ops.any(one_n)) {
              bool settled = false;
              const uint32_t rn = ops.tier_single_n(G, q1[u], q2[u], qn[u], one_n, settled);
              if (one_n && settled) {
                r[u] = rn;
                need_m &= ~(1u << u);
              }
            }
#ifdef BC_TIER_BOTH
            const bool exact = false;
#else
            uint32_t exact_idx = kFail;
            const bool exact = probe && tier_exact_in_block0(G, q1[u], q2[u], t, exact_idx);
            if (exact) {
              r[u] = exact_idx;
              need_m &= ~(1u << u);
            }
            if (probe && !exact) tier_fetch_blocks<1, 1>(G, q1[u], q2[u], t);
#endif
            if (probe && !exact) {
              uint32_t best, cnt, idx;
              tier_score(G, q1[u], q2[u], t, best, cnt, idx);
              if (best <= 1u) {
                r[u] = (cnt == 1u && best <= G.max_err) ? idx : kFail;
                need_m &= ~(1u << u);
              }
            }
          } else if (probe) {
            const uint64_t key = (uint64_t)q1[u] | ((uint64_t)q2[u] << 32);
            uint32_t h = (uint32_t)hash64(key) & G.hmask;
            for (;;) {
              const uint32_t v = G.hvals()[h];
              if (v == kFail) break;
              if (G.hkeys()[h] == key) {
                r[u] = v;
                need_m &= ~(1u << u);
                break;
              }
              h = (h + 1u) & G.hmask;
            }
          }
        }
      }
    }
    // Captures with one 'N' are rare per lane but not per wavefront: every lane settles its lowest
    // pending group, so one pass usually serves the whole wave whichever groups the 'N's fell into.
    while (ops.any(pend_n != 0u)) {
      const uint32_t u_sel = pend_n ? ctz(pend_n) : 0u;
      uint32_t s_q1 = 0, s_q2 = 0, s_qn = 1u, s_off = 0, s_nb = 1u, s_len = 1u, s_max = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (g0 + u < ng) {
          const DevGroup& G = pl.groups[g0 + u];
          const bool me = u_sel == (uint32_t)u;
          s_q1 = me ? q1[u] : s_q1;
          s_q2 = me ? q2[u] : s_q2;
          s_qn = me ? qn[u] : s_qn;
          s_off = me ? G.lhash_off : s_off;
          s_nb = me ? G.lhash_nb : s_nb;
          s_len = me ? G.len : s_len;
          s_max = me ? G.max_err : s_max;
        }
      }
      bool settled = true;
      uint32_t rs = kFail;
      if (pend_n) rs = single_n_lhash(ops.lhash(), s_off, s_nb, s_len, s_max, s_q1, s_q2, s_qn, settled);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool me = pend_n != 0u && u_sel == (uint32_t)u;
        r[u] = me ? rs : r[u];
        gather_n_m |= (me && !settled) ? (1u << u) : 0u;
      }
      pend_n &= pend_n - 1u;
    }
    if (pl.abl() & 0x1000u) gather_m = 0;  // perf experiment: what the table round trip costs
    // correction-table gathers of the captures LDS did not answer: all lanes load (the idle ones entry
    // 0, one shared cache line), so the four loads are in flight together
    uint32_t tv[4];
    uint32_t n_gathers = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      tv[u] = 0;
      if (g0 + u < ng && pl.groups[g0 + u].mode == kSetDirect) {  // wave-uniform: no branch between the loads
        const DevGroup& G = pl.groups[g0 + u];
        ++n_gathers;
        tv[u] = G.dtable()[((gather_m >> u) & 1u) ? (q1[u] | (q2[u] << G.len)) : 0u];
      }
    }
    ops.mark(6);
    // ... and while they are, the quality lines are judged
    if (!quality_done) quality_filter(n_gathers);
    // one 'N' and no substitution is a reference (or no LDS table): four table entries decide.  As
    // above, every lane takes its lowest pending group, so the wave usually needs one round trip.
    {
      uint32_t pend_g = gather_n_m;
      while (ops.any(pend_g != 0u)) {
        const uint32_t u_sel = pend_g ? ctz(pend_g) : 0u;
        uint32_t s_q1 = 0, s_q2 = 0, s_qn = 1u, s_len = 1u, s_max = 0;
        uint64_t s_tab = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (g0 + u < ng && pl.groups[g0 + u].mode == kSetDirect) {
            const DevGroup& G = pl.groups[g0 + u];
            const bool me = u_sel == (uint32_t)u;
            s_q1 = me ? q1[u] : s_q1;
            s_q2 = me ? q2[u] : s_q2;
            s_qn = me ? qn[u] : s_qn;
            s_len = me ? G.len : s_len;
            s_max = me ? G.max_err : s_max;
            s_tab = me ? (uint64_t)G.dtable() : s_tab;
          }
        }
        uint32_t rs = kFail;
        if (pend_g) rs = single_n_lookup(reinterpret_cast<const BC_GLOBAL uint32_t*>(s_tab), s_len, s_max, s_q1, s_q2, s_qn);
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = (pend_g != 0u && u_sel == (uint32_t)u) ? rs : r[u];
        pend_g &= pend_g - 1u;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if ((gather_m >> u) & 1u) r[u] = (tv[u] & 0xFFFFu) == (uint32_t)kFail16 ? kFail : (tv[u] & 0xFFFFu);
    ops.mark(7);
    const bool pre_ok = active && outcome == kMatched;  // anchored and of good quality
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (g0 + u < ng) {
        const DevGroup& G = pl.groups[g0 + u];
        if (G.mode != kSetNone) {
          // a read that already failed an earlier group is not searched again (parse.rs:481, 500)
          bool nd_ = ((need_m >> u) & 1u) && outcome == kMatched;
          if (pl.abl() & 0x2u) nd_ = false;
          if ((pl.abl() & 0x8000u) && qn[u]) nd_ = false;  // experiment: captures with 'N' never reach the search
          const uint32_t rr = ops.nearest(G, q1[u], q2[u], qn[u], qx[u], nd_);
          if (nd_) r[u] = rr;
          if (pre_ok && outcome == kMatched) {
            if (r[u] == kFail)
              outcome = (G.type == kGroupSample) ? kSampleBarcode : kBarcode;  // parse.rs:132-140
            else if (pl.defer_search() && r[u] == kDeferred) {
              // single-group plans only (DevPlan::defer_search): the caller searches later; the capture goes back in
              // the result's index and random-barcode fields, which such a read has no use for
              outcome = kPending;
              didx = (uint64_t)q1[u] | ((uint64_t)q2[u] << 32);
              pending_n = qn[u];
            } else
              didx += (uint64_t)r[u] * G.table_stride;
          }
        } else if (pre_ok && outcome == kMatched) {
          // no known set: the capture is taken as it is (parse.rs:453-454, 487)
          if (qx[u]) raw_foreign = true;  // a byte outside ACGTN has no code -- matters only if the read ends up counted
          didx += base5_code(q1[u], q2[u], qn[u], G.len) * G.table_stride;
        }
      }
    }
  }
  if (!quality_done) quality_filter(0u);  // a scheme without barcode groups
  // a later group may still have failed the read: then that failure is the outcome, as in the reference
  if (raw_foreign && outcome == kMatched) unsupported = true;
  // ---- random barcode: kept as captured, never corrected (parse.rs:510-516) ------------------
  if (pl.has_random) {
    uint32_t r1, r2, rn;
    extract_planes<NW>(P, pl.rnd_off, pl.rnd_len, r1, r2, rn);
    const uint32_t rx = anyx ? extract_uniform<NW>(P.px, pl.rnd_off, pl.rnd_len) : 0u;
    if (rx && active && outcome == kMatched) unsupported = true;  // a byte outside ACGTN has no code
    res.rcode = base5_code(r1, r2, rn, pl.rnd_len);
  }
  if (unsupported) outcome = kUnsupported;
  if (pl.defer_search() && outcome == kPending) res.rcode = pending_n;
  res.outcome = outcome;
  res.dense_idx = didx;
  return res;
}

}  // namespace bc
