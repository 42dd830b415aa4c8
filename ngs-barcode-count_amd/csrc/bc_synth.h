// bc_synth.h -- counter-based synthetic read generator.
// Every byte of read i is a pure function of (seed, i, position), so the SAME code produces the
// same reads on the host (parity tests at small sizes) and on the device (100 M-read workloads
// that never exist on disk).  Plain integer arithmetic only.
#pragma once
#include <stdint.h>

#include "bc_device_plan.h"
#include "bc_intrin.h"

namespace bc {

constexpr int kSynthMaxL = kMaxNW * 32;

struct SynthGroup {
  uint32_t type;  // GroupType
  uint32_t off, len;
  uint32_t n_refs;     // 0: random bases
  uint32_t zmul;       // Zipf variant: rank k is reference (k - 1) * zmul mod n_refs (coprime: the abundant references
                       // are scattered over the set, as they are in a real library, not its first few entries)
  const char* refs;    // n_refs * len ASCII bases
};

struct SynthDev {
  uint64_t seed;
  uint32_t read_len;
  uint32_t L;
  uint32_t p_sub, p_n, p_lowq;  // probabilities * 2^32
  uint32_t phred_lo, phred_hi, lowq_lo, lowq_hi;
  uint64_t n_molecules;
  uint32_t zipf;                 // counted-barcode indices drawn octave-uniformly (P(k) ~ 1/k) instead of uniformly
  uint64_t geo_total;            // > 0: PCR copies per molecule geometric with mean 2 (P(c) = 2^-c), the copies scattered
                                 // over the geo_total reads of the job by a fixed permutation
  uint32_t n_groups;
  uint32_t n_sb;                 // groups that can receive low qualities (sample + counted)
  SynthGroup groups[kMaxGroups];
  // per construct position: 'A','C','G','T' constant, 'n' scheme-N (random base), or 0x80|group index
  uint8_t fmt[kSynthMaxL];
};

BC_HD uint64_t synth_mix(uint64_t seed, uint64_t i, uint64_t k) {
  uint64_t x = seed + i * 0x9E3779B97F4A7C15ull + k * 0xD1B54A32D192ED03ull;
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32;
  return x;
}

// one base + one quality byte of read i at position pos
struct SynthRead {
  uint32_t off;        // construct start
  uint64_t mol;        // molecule the construct content comes from
  int lowq_group;      // group with low qualities or -1
  uint32_t ref_idx[kMaxGroups];
  uint64_t rnd_bits[kMaxGroups];
};

// a fixed bijection of [0, n): multiply / xor-shift / multiply on the next power of two, walked until it lands below n
BC_HD uint64_t synth_permute(uint64_t x, uint64_t n, uint64_t seed) {
  uint32_t k = 1;
  while ((1ull << k) < n) ++k;
  const uint64_t mask = k >= 64 ? ~0ull : ((1ull << k) - 1ull);
  const uint64_t a = (seed * 2ull + 0x9E3779B97F4A7C15ull) | 1ull, b = (seed * 6ull + 0xD1B54A32D192ED03ull) | 1ull;
  do {
    x = (x * a) & mask;
    x ^= x >> ((k + 1) / 2);
    x = (x * b + seed) & mask;
    x ^= x >> ((k + 1) / 2);
  } while (x >= n);
  return x;
}

// Molecule of slot j when every molecule has 1 + Geometric(1/2) copies (mean 2): slots come in blocks of 128 shared by
// 64 molecules; molecule m of a block has as many copies as its hash has trailing zeros, plus one; the block's last
// molecule takes what is left (or nothing).  No prefix sum over the job: 63 hashes at most.
BC_HD uint64_t synth_geo_molecule(uint64_t seed, uint64_t j) {
  const uint64_t block = j >> 7;
  const uint32_t r = (uint32_t)(j & 127u);
  uint32_t upto = 0, m = 0;
  for (; m < 63u; ++m) {
    const uint64_t h = synth_mix(seed ^ 0x6E0C0FFEEull, block, m) | (1ull << 40);
    uint32_t c = 1;
    for (uint64_t t = h; !(t & 1ull); t >>= 1) ++c;
    upto += c;
    if (r < upto) break;
  }
  return block * 64ull + m;
}

BC_HD void synth_begin(const SynthDev& S, uint64_t i, SynthRead& R) {
  const uint32_t slack = S.read_len > S.L ? S.read_len - S.L : 0u;
  R.off = slack ? (uint32_t)(synth_mix(S.seed, i, 0) % slack) : 0u;  // uniform in [0, R-L-1]
  R.mol = S.n_molecules ? synth_mix(S.seed, i, 1) % S.n_molecules : i;
  if (S.geo_total) R.mol = synth_geo_molecule(S.seed, synth_permute(i % S.geo_total, S.geo_total, S.seed));
  const uint64_t hl = synth_mix(S.seed, i, 2);
  R.lowq_group = -1;
  if ((uint32_t)hl < S.p_lowq && S.n_sb) {
    uint32_t which = (uint32_t)((hl >> 32) % S.n_sb);
    for (uint32_t g = 0; g < S.n_groups; ++g) {
      if (S.groups[g].type != kGroupRandom) {
        if (which == 0) {
          R.lowq_group = (int)g;
          break;
        }
        --which;
      }
    }
  }
  for (uint32_t g = 0; g < S.n_groups; ++g) {
    const uint64_t hg = synth_mix(S.seed ^ 0xB0C0DEull, R.mol, 16 + g);
    const uint32_t nr = S.groups[g].n_refs;
    R.ref_idx[g] = nr ? (uint32_t)(hg % nr) : 0u;
    if (S.zipf && nr > 1u && S.groups[g].type == kGroupBarcode) {
      // Zipf-like with exponent 1, in integer arithmetic (the same on host and device): the octave [2^b, 2^(b+1)) of
      // the rank k is uniform, k is uniform inside it, so P(k) = 1 / (B 2^b) ~ 1/k.  Rank 1 is reference 0.
      uint32_t octaves = 0;
      while ((1ull << octaves) <= nr) ++octaves;  // ranks 1 .. nr span `octaves` octaves
      const uint32_t b = (uint32_t)((hg >> 40) % octaves);
      uint64_t k = (1ull << b) + ((hg & 0xFFFFFFFFull) % (1ull << b));
      if (k > nr) k = (k % nr) + 1ull;
      R.ref_idx[g] = (uint32_t)(((k - 1ull) * S.groups[g].zmul) % nr);
    }
    R.rnd_bits[g] = hg;
  }
}

BC_HD void synth_byte(const SynthDev& S, uint64_t i, const SynthRead& R, uint32_t pos, uint8_t& base, uint8_t& qual) {
  const uint64_t hp = synth_mix(S.seed, i, 64 + pos);
  const char* acgt = "ACGT";
  uint8_t b = (uint8_t)acgt[(hp >> 34) & 3];
  int grp = -1;
  if (pos >= R.off && pos < R.off + S.L) {
    const uint8_t f = S.fmt[pos - R.off];
    if (f & 0x80u) {
      grp = f & 0x7F;
      const SynthGroup& G = S.groups[grp];
      const uint32_t k = pos - R.off - G.off;
      if (G.n_refs)
        b = (uint8_t)G.refs[(uint64_t)R.ref_idx[grp] * G.len + k];
      else
        b = (uint8_t)acgt[(R.rnd_bits[grp] >> (2 * (k & 31))) & 3];
    } else if (f != 'n') {
      b = f;
    }
  }
  const uint32_t m = (uint32_t)hp;
  const uint64_t sub_n = (uint64_t)S.p_sub + (uint64_t)S.p_n;
  if (m < S.p_sub) {
    // one of the other three bases
    uint32_t cur = b == 'A' ? 0u : (b == 'C' ? 1u : (b == 'G' ? 2u : 3u));
    b = (uint8_t)acgt[(cur + 1u + (uint32_t)((hp >> 36) % 3)) & 3u];
  } else if ((uint64_t)m < sub_n) {
    b = 'N';
  }
  uint32_t lo = S.phred_lo, hi = S.phred_hi;
  if (grp >= 0 && grp == R.lowq_group) {
    lo = S.lowq_lo;
    hi = S.lowq_hi;
  }
  const uint32_t q = lo + (uint32_t)((hp >> 40) & 0xFFFFu) % (hi - lo + 1u);
  base = b;
  qual = (uint8_t)(33u + q);
}

}  // namespace bc
