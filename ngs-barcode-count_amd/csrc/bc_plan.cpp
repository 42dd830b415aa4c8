// bc_plan.cpp -- scheme compiler, known-barcode loaders and error budgets (host side of the
// C ABI).  Mirrors, for the hot path's static inputs, the reference's
//   SequenceFormat::parse_format_file   info.rs:215-310
//   BarcodeConversions::*               info.rs:364-456
//   MaxSeqErrors::new                   info.rs:490-543
// and lowers them to the bit-plane programs the gfx950 kernels run (bc_device_plan.h).
#include "bc_plan.hpp"

#include <string.h>

#include <algorithm>

#include "../../include/barcode_count_hip.h"

namespace bc {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }
const char* get_error() { return g_error.c_str(); }

namespace {

// str::lines(): split on '\n', a trailing "\r" of each line is dropped, no empty last line
std::vector<std::string> rust_lines(const char* text, size_t len) {
  std::vector<std::string> out;
  size_t i = 0;
  while (i < len) {
    size_t j = i;
    while (j < len && text[j] != '\n') ++j;
    size_t e = j;
    if (e > i && text[e - 1] == '\r') --e;
    out.emplace_back(text + i, e - i);
    i = j + 1;
  }
  return out;
}

inline bool is_base_ci(char c) {
  switch (c) {
    case 'A': case 'T': case 'G': case 'C': case 'a': case 't': case 'g': case 'c': return true;
    default: return false;
  }
}

enum TokKind { kTokNone, kTokSample, kTokBarcode, kTokRandom, kTokNs, kTokBases };

// one token of (?i)(\{\d+\})|(\[\d+\])|(\(\d+\))|N+|[ATGC]+  (info.rs:232); anything else is skipped
TokKind lex(const std::string& s, size_t i, size_t& tok_len) {
  const char c = s[i];
  const char* open = "{[(";
  const char* close = "}])";
  for (int k = 0; k < 3; ++k) {
    if (c == open[k]) {
      size_t j = i + 1;
      while (j < s.size() && s[j] >= '0' && s[j] <= '9') ++j;
      if (j == i + 1 || j >= s.size() || s[j] != close[k]) return kTokNone;
      tok_len = j + 1 - i;
      return k == 0 ? kTokBarcode : (k == 1 ? kTokSample : kTokRandom);
    }
  }
  if (c == 'N' || c == 'n') {
    size_t j = i;
    while (j < s.size() && (s[j] == 'N' || s[j] == 'n')) ++j;
    tok_len = j - i;
    return kTokNs;
  }
  if (is_base_ci(c)) {
    size_t j = i;
    while (j < s.size() && is_base_ci(s[j])) ++j;
    tok_len = j - i;
    return kTokBases;
  }
  return kTokNone;
}

std::vector<std::string> split_take(const std::string& line, size_t want) {
  std::vector<std::string> f;
  size_t s = 0;
  for (size_t i = 0; i <= line.size() && f.size() < want; ++i) {
    if (i == line.size() || line[i] == ',') {
      f.emplace_back(line, s, i - s);
      s = i + 1;
    }
  }
  return f;
}

}  // namespace
}  // namespace bc

using namespace bc;

void bc_plan::recompute_budgets() {
  // main.rs:55-63 feeds SequenceFormat's fields to MaxSeqErrors::new
  std::vector<uint16_t> sizes(barcode_lengths.begin(), barcode_lengths.end());
  std::vector<uint16_t> out(2 + sizes.size() + 1);
  bc_max_seq_errors(opt_sample, sample_length, opt_barcode, sizes.data(), (uint32_t)sizes.size(), opt_constant,
                    (uint16_t)constant_region_length, out.data());
  max_constant = out[0];
  max_sample = out[1];
  max_barcode.assign(out.begin() + 2, out.begin() + 2 + sizes.size());
}

uint32_t bc_plan::quality_threshold(uint32_t n) const {
  // smallest integer score sum s with !(fl32(s / n) < min_quality): the f32 mean test of
  // parse.rs:352-355 is then exactly "sum < T_n" (sums of u8 scores are exact in f32)
  if (n == 0) return 0;
  uint32_t lo = 0, hi = 255u * n + 1u;  // hi: the test can never pass -> everything below is low
  const float fn = (float)n;
  auto low = [&](uint32_t s) { return ((float)s / fn) < min_quality; };
  if (!low(0)) return 0;
  if (low(255u * n)) return hi;
  lo = 0;
  hi = 255u * n;  // low(lo) true, low(hi) false
  while (hi - lo > 1) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (low(mid))
      lo = mid;
    else
      hi = mid;
  }
  return hi;
}

static bool pack_ref(const std::string& s, uint32_t& r1, uint32_t& r2, uint32_t& rn) {
  r1 = r2 = rn = 0;
  for (size_t i = 0; i < s.size(); ++i) {
    const char c = s[i];
    if (c == 'N') {
      rn |= 1u << i;
    } else if (c == 'A' || c == 'C' || c == 'G' || c == 'T') {
      r1 |= (uint32_t)((c >> 1) & 1) << i;
      r2 |= (uint32_t)((c >> 2) & 1) << i;
    } else {
      return false;
    }
  }
  return true;
}

bool bc_plan::lower(HostDevPlan& out) const {
  if (!unsupported.empty()) {
    set_error("unsupported scheme: " + unsupported);
    return false;
  }
  if (regex_length == 0 || length > (uint32_t)kMaxNW * 32u) {
    set_error("unsupported scheme: format length must be 1.." + std::to_string(kMaxNW * 32));
    return false;
  }
  if (literal_n_constant) {
    set_error("unsupported scheme: lower-case 'n' constants (literal 'N' bases) need the wave-per-read kernel");
    return false;
  }
  if (max_constant > 31) {
    set_error("unsupported plan: --max-errors-constant above 31");
    return false;
  }
  DevPlan& P = out.plan;
  memset(&P, 0, sizeof(P));
  P.L = regex_length;  // bytes a match spans (shorter than format_string only for a token mixing 'N' and 'n')
  P.RL = (uint32_t)regions_string.size();
  P.max_const = max_constant;
  P.no_repair = lowercase_constants ? 1u : 0u;
  uint32_t nb = 0;
  while ((1u << nb) <= (uint32_t)max_constant) ++nb;  // counters hold 0 .. 2^nb-1 >= max_constant
  P.nb = nb;
  P.quality_on = min_quality > 0.0f ? 1u : 0u;  // parse.rs:98

  // shift/apply programs, one per position class
  std::vector<uint32_t> where[kClasses];
  for (uint32_t p = 0; p < pos.size(); ++p) {
    if (pos[p].kind == kPosConst)
      where[(pos[p].letter >> 1) & 3].push_back(p);
    else if (pos[p].kind == kPosFmtN)
      where[kClassFmtN].push_back(p);
  }
  for (int c = 0; c < kClasses; ++c) {
    std::vector<uint32_t> deltas;
    uint32_t cur = 0;
    bool gap = false;
    for (uint32_t p : where[c]) {
      deltas.push_back(p - cur);
      gap = gap || (p - cur) > 31;
      cur = p;
    }
    P.n_pos[c] = (uint32_t)deltas.size();
    uint32_t n = 0;
    if (!gap) {
      P.prog_mode[c] = 0;
      size_t i = 0;
      for (; i + 3 <= deltas.size() && n < (uint32_t)kMaxEntries; i += 3)
        P.prog[c][n++] = deltas[i] | (deltas[i + 1] << 8) | (deltas[i + 2] << 16);
      if (i + 3 <= deltas.size()) {
        set_error("unsupported scheme: too many constant positions");
        return false;
      }
      const uint32_t k = (uint32_t)(deltas.size() - i);
      uint32_t e = k << 24;
      for (uint32_t t = 0; t < k; ++t) e |= deltas[i + t] << (8 * t);
      P.n3[c] = n;
      P.prog[c][n] = e;
    } else {
      P.prog_mode[c] = 1;
      for (uint32_t d : deltas) {
        while (d > 31 && n < (uint32_t)kMaxEntries) {
          P.prog[c][n++] = 31u;  // k = 0: shift only
          d -= 31;
        }
        if (n >= (uint32_t)kMaxEntries) {
          set_error("unsupported scheme: too many constant positions");
          return false;
        }
        P.prog[c][n++] = (1u << 24) | d;
      }
      P.n3[c] = n;
    }
  }
  for (uint32_t p = 0; p < pos.size(); ++p)
    if (pos[p].kind == kPosConst) P.cmask[p >> 5] |= 1u << (p & 31);
  P.has_fmtn = where[kClassFmtN].empty() ? 0u : 1u;

  // quality runs: maximal stretches of one non-'C' letter of regions_string (parse.rs:340-372)
  {
    uint32_t n = 0;
    size_t i = 0;
    const std::string& r = regions_string;
    while (i < r.size()) {
      size_t j = i;
      while (j < r.size() && r[j] == r[i]) ++j;
      if (r[i] != 'C' && j < r.size()) {  // a run ending the string is never evaluated (Appendix A Q9)
        if (n >= (uint32_t)kMaxRuns) {
          set_error("unsupported scheme: more than " + std::to_string(kMaxRuns) + " barcode runs");
          return false;
        }
        P.run_off[n] = (uint32_t)i;
        P.run_len[n] = (uint32_t)(j - i);
        P.run_thr[n] = quality_threshold((uint32_t)(j - i));
        ++n;
      }
      i = j;
    }
    P.n_runs = n;
  }

  // groups: sample first, then counted barcodes in order (the dense index is row-major in that order)
  struct Pending {
    const FormatGroup* g;
    const KnownSet* set;
    uint32_t max_err;
  };
  std::vector<Pending> order;
  for (const auto& g : groups)
    if (g.type == kGroupSample) order.push_back({&g, &samples, max_sample});
  for (uint32_t b = 1; b <= barcode_num; ++b)
    for (const auto& g : groups)
      if (g.type == kGroupBarcode && g.number == b) order.push_back({&g, &counted[b - 1], max_barcode[b - 1]});
  if (order.size() > (size_t)kMaxGroups) {
    set_error("unsupported scheme: too many barcode groups");
    return false;
  }
  P.n_groups = (uint32_t)order.size();
  out.sets.assign(order.size(), HostSet());
  out.n_samples = sample_barcode ? (uint32_t)samples.size() : 1u;
  // Results::add_count with a sample file but no sample group: the key "barcode" is absent, the
  // increment lands in a temporary, yet the read counts as matched (info.rs:762-766)
  // (with a random barcode the same call creates the "barcode" entry instead, info.rs:792-801)
  P.discard_counts = (!sample_barcode && samples.size() > 0 && !random_barcode) ? 1u : 0u;
  unsigned __int128 entries = 1;
  for (int i = (int)order.size() - 1; i >= 0; --i) {
    const Pending& pd = order[i];
    DevGroup& G = P.groups[i];
    G.type = pd.g->type;
    G.off = pd.g->off;
    G.len = pd.g->len;
    if (G.len == 0 || G.len > 32) {
      set_error("unsupported scheme: barcode groups must be 1..32 bases");
      return false;
    }
    G.n_refs = (uint32_t)pd.set->size();
    G.max_err = pd.max_err;
    G.table_stride = (uint64_t)entries;
    if (G.n_refs == 0) {
      // no known set (sample_seqs.is_empty() parse.rs:453 / counted_barcode_seqs.is_empty() parse.rs:487):
      // the capture itself is the key; its base-5 code takes the place of a reference index
      if (G.len > 27) {
        set_error("unsupported plan: barcodes without a conversion file must be at most 27 bases");
        return false;
      }
      G.mode = kSetNone;
      P.sparse = 1;
      for (uint32_t k = 0; k < G.len; ++k) entries *= 5;
      if (entries >= ((unsigned __int128)1 << 63)) {
        set_error("unsupported plan: the barcodes without conversion files do not fit a 64-bit key");
        return false;
      }
      continue;
    }
    entries *= G.n_refs;
    if (entries >= ((unsigned __int128)1 << 63)) {
      set_error("unsupported plan: the (sample, barcode tuple) space does not fit a 64-bit key");
      return false;
    }
    HostSet& H = out.sets[i];
    H.r1.resize(G.n_refs);
    H.r2.resize(G.n_refs);
    H.rn.resize(G.n_refs);
    H.rlen.resize(G.n_refs);
    G.has_odd = 0;
    for (uint32_t j = 0; j < G.n_refs; ++j) {
      const std::string& s = pd.set->seqs[j];
      if (s.size() > 32 || !pack_ref(s, H.r1[j], H.r2[j], H.rn[j])) {
        set_error("unsupported plan: known barcode '" + s + "' (only A,C,G,T,N and at most 32 bases)");
        return false;
      }
      H.rlen[j] = (uint8_t)s.size();
      if (H.rn[j] || s.size() != G.len) G.has_odd = 1;
    }
    if (G.len <= 10 && G.n_refs < 65535u) {
      G.mode = kSetDirect;
    } else {
      G.mode = kSetHash;
      uint32_t slots = 16;
      while (slots < 2 * G.n_refs) slots <<= 1;
      G.hmask = slots - 1;
      H.hkeys.assign(slots, 0);
      H.hvals.assign(slots, kFail);
      for (uint32_t j = 0; j < G.n_refs; ++j) {
        if (H.rn[j] || H.rlen[j] != G.len) continue;  // only an identical string is an exact member
        const uint64_t key = (uint64_t)H.r1[j] | ((uint64_t)H.r2[j] << 32);
        // same mixer as bc::hash64 in bc_lane.h
        uint64_t x = key;
        x ^= x >> 30;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 27;
        x *= 0x94D049BB133111EBull;
        x ^= x >> 31;
        uint32_t h = (uint32_t)x & G.hmask;
        while (H.hvals[h] != kFail) h = (h + 1) & G.hmask;
        H.hkeys[h] = key;
        H.hvals[h] = j;
      }
      // pigeonhole seed index (see bc_device_plan.h)
      const uint32_t nb = G.max_err + 1;
      uint32_t blen = G.len / nb;
      if (blen > 8) blen = 8;
      if (G.n_refs >= 128 && blen >= 3) {
        std::vector<uint32_t> plain;
        for (uint32_t j = 0; j < G.n_refs; ++j) {
          if (H.rn[j] || H.rlen[j] != G.len)
            H.odd_list.push_back(j);
          else
            plain.push_back(j);
        }
        // [blocks][4^bl + 1] bucket starts + [blocks][plain] entries {r1, r2, index, 0} ordered by block value
        auto build_index = [&](uint32_t blocks, uint32_t bl, std::vector<uint32_t>& offs, std::vector<uint32_t>& list) {
          const uint32_t nbk = 1u << (2 * bl), bm = (1u << bl) - 1;
          offs.assign((size_t)blocks * (nbk + 1), 0);
          list.assign((size_t)blocks * plain.size() * 4, 0);
          for (uint32_t b = 0; b < blocks; ++b) {
            auto value = [&](uint32_t j) { return ((H.r1[j] >> (b * bl)) & bm) | (((H.r2[j] >> (b * bl)) & bm) << bl); };
            uint32_t* off = &offs[(size_t)b * (nbk + 1)];
            for (uint32_t j : plain) off[value(j) + 1]++;
            for (uint32_t v = 0; v < nbk; ++v) off[v + 1] += off[v];
            std::vector<uint32_t> cur(off, off + nbk);
            for (uint32_t j : plain) {
              uint32_t* e = &list[((size_t)b * plain.size() + cur[value(j)]++) * 4];
              e[0] = H.r1[j];
              e[1] = H.r2[j];
              e[2] = j;
            }
          }
        };
        build_index(nb, blen, H.seed_off, H.seed_list);
        G.seed_nb = nb;
        G.seed_blen = blen;
        G.n_idx = (uint32_t)plain.size();
        G.n_odd = (uint32_t)H.odd_list.size();
        // coarser index for up to two mismatches (three blocks), in front of the full one
        if (nb > 3 && G.len / 3u >= 5u) {
          G.seed2_nb = 3;
          G.seed2_blen = std::min<uint32_t>(8u, G.len / 3u);
          build_index(G.seed2_nb, G.seed2_blen, H.seed2_off, H.seed2_list);
        }
        // ... and one for up to three (four blocks) between the two
        if (G.seed2_nb && nb > 4 && G.len / 4u >= 5u) {
          G.seed3_nb = 4;
          G.seed3_blen = std::min<uint32_t>(8u, G.len / 4u);
          build_index(G.seed3_nb, G.seed3_blen, H.seed3_off, H.seed3_list);
        }
        // first tier (see bc_device_plan.h): two blocks, one per half of the barcode
        const uint32_t tb = std::min<uint32_t>(8u, G.len / 2u);
        if (G.max_err >= 1 && tb >= 4 && H.odd_list.empty()) {
          const uint32_t tnbk = 1u << (2 * tb), tbm = (1u << tb) - 1, stride = G.len / 2u;
          H.tier_off.assign((size_t)2 * (tnbk + 1), 0);
          H.tier_list.assign((size_t)2 * plain.size() * 4, 0);
          for (uint32_t b = 0; b < 2; ++b) {
            const uint32_t sh = b * stride;
            auto value = [&](uint32_t j) { return ((H.r1[j] >> sh) & tbm) | (((H.r2[j] >> sh) & tbm) << tb); };
            uint32_t* off = &H.tier_off[(size_t)b * (tnbk + 1)];
            for (uint32_t j : plain) off[value(j) + 1]++;
            for (uint32_t v = 0; v < tnbk; ++v) off[v + 1] += off[v];
            std::vector<uint32_t> cur(off, off + tnbk);
            for (uint32_t j : plain) {
              uint32_t* e = &H.tier_list[((size_t)b * plain.size() + cur[value(j)]++) * 4];
              e[0] = H.r1[j];
              e[1] = H.r2[j];
              e[2] = j;
            }
          }
          // the head of every bucket, laid out so that a lookup is one 64-byte line
          H.tier_bkt.assign((size_t)2 * tnbk * 16, 0);
          for (uint32_t b = 0; b < 2; ++b) {
            const uint32_t* off = &H.tier_off[(size_t)b * (tnbk + 1)];
            for (uint32_t v = 0; v < tnbk; ++v) {
              const uint32_t n = off[v + 1] - off[v];
              uint32_t* line = &H.tier_bkt[((size_t)b * tnbk + v) * 16];
              for (uint32_t k = 0; k < 4; ++k) {
                if (k < n) {
                  const uint32_t* e = &H.tier_list[((size_t)b * plain.size() + off[v] + k) * 4];
                  line[k * 4 + 0] = e[0];
                  line[k * 4 + 1] = e[1];
                  line[k * 4 + 2] = e[2];
                }
                line[k * 4 + 3] = n;
              }
            }
          }
          G.tier_blen = tb;
          G.tier_stride = stride;
          // compact form when an entry fits 61 bits (bc_device_plan.h)
          uint32_t idx_bits = 1;
          while ((1ull << idx_bits) < (uint64_t)G.n_refs) ++idx_bits;
          if (2u * G.len + idx_bits <= 61u) {
            std::vector<uint32_t> compact((size_t)2 * tnbk * 8, 0);
            for (uint32_t b = 0; b < 2; ++b) {
              for (uint32_t v = 0; v < tnbk; ++v) {
                const uint32_t* line = &H.tier_bkt[((size_t)b * tnbk + v) * 16];
                uint32_t* out8 = &compact[((size_t)b * tnbk + v) * 8];
                const uint32_t n = line[3];
                for (uint32_t k = 0; k < 4; ++k) {
                  uint64_t e = (uint64_t)line[k * 4 + 0] | ((uint64_t)line[k * 4 + 1] << G.len) |
                               ((uint64_t)line[k * 4 + 2] << (2u * G.len));
                  if (k >= n) e = 0;
                  if (k == 0) e |= (uint64_t)std::min<uint32_t>(n, 7u) << 61;
                  out8[k * 2] = (uint32_t)e;
                  out8[k * 2 + 1] = (uint32_t)(e >> 32);
                }
              }
            }
            H.tier_bkt.swap(compact);
            G.tier_compact = 1;
          }
        }
      }
    }
  }
  // LDS exact-match tables in front of the correction tables (bc_device_plan.h), while they fit
  out.lhash.clear();
  P.lhash_vec = 0;
  for (size_t i = 0; i < P.n_groups; ++i) {
    DevGroup& G = P.groups[i];
    G.lhash_nb = 0;
    G.lhash_off = 0;
    G.lhash_complete = 0;
    G.index = (uint32_t)i;
    if (G.mode != kSetDirect) continue;
    const HostSet& H = out.sets[i];
    const uint32_t ibits = 32u - 2u * G.len;
    std::vector<uint32_t> plain;
    for (uint32_t j = 0; j < G.n_refs; ++j)
      if (!H.rn[j] && H.rlen[j] == G.len) plain.push_back(j);
    if (plain.empty() || (ibits < 32u && G.n_refs > (1u << ibits))) continue;
    auto key_of = [&](uint32_t j) { return H.r1[j] | (H.r2[j] << G.len); };
    // Two-choice placement with evictions into 4-entry buckets, aimed at 85 % occupancy (the scheme
    // holds up to ~97 %); should the random walk fail, the table grows by a sixth and is rebuilt.
    uint32_t nbuckets = std::max<uint32_t>(2u, (uint32_t)((plain.size() * 20 + 67) / 68));
    std::vector<uint32_t> slot_ref;
    bool complete = false;
    for (int attempt = 0; attempt < 6 && !complete; ++attempt, nbuckets += nbuckets / 6 + 1) {
      if (P.lhash_vec + nbuckets > kLhashMaxVec) break;
      slot_ref.assign((size_t)nbuckets * 4, kFail);
      std::vector<uint8_t> fill(nbuckets, 0);
      complete = true;
      uint64_t rng = 0x9E3779B97F4A7C15ull;
      for (uint32_t j0 : plain) {
        uint32_t j = j0;
        bool placed = false;
        for (int moves = 0; moves < 2000 && !placed; ++moves) {
          const uint32_t key = key_of(j);
          const uint32_t b1 = lhash_bucket(key, kLhashMul1, nbuckets), b2 = lhash_bucket(key, kLhashMul2, nbuckets);
          const uint32_t b = fill[b1] <= fill[b2] ? b1 : b2;
          if (fill[b] < 4) {
            slot_ref[(size_t)b * 4 + fill[b]++] = j;
            placed = true;
          } else {
            rng = rng * 6364136223846793005ull + 1442695040888963407ull;
            const uint32_t vb = (rng >> 33) & 1u ? b1 : b2;
            const uint32_t vs = (uint32_t)(rng >> 40) & 3u;
            std::swap(j, slot_ref[(size_t)vb * 4 + vs]);
          }
        }
        if (!placed) {
          complete = false;
          break;
        }
      }
      if (complete) break;
    }
    if (!complete) continue;  // no LDS table for this group: every capture goes to dtable
    const uint32_t filler = (key_of(plain[0]) << ibits) | plain[0];
    std::vector<uint32_t> img((size_t)nbuckets * 4, filler);
    for (size_t i = 0; i < slot_ref.size(); ++i)
      if (slot_ref[i] != kFail) img[i] = (key_of(slot_ref[i]) << ibits) | slot_ref[i];
    G.lhash_nb = nbuckets;
    G.lhash_off = P.lhash_vec;
    G.lhash_complete = 1u;
    out.lhash.insert(out.lhash.end(), img.begin(), img.end());
    P.lhash_vec += nbuckets;
  }
  if (!P.sparse && entries > ((unsigned __int128)1 << 40)) {
    set_error("unsupported plan: dense counter table above 2^40 entries");
    return false;
  }
  out.table_entries = (uint64_t)entries;  // dense plans: table size; sparse plans: size of the tuple-key space
  if (random_barcode) {
    for (const auto& g : groups) {
      if (g.type != kGroupRandom) continue;
      if (g.len == 0 || g.len > 27) {
        set_error("unsupported scheme: random barcodes must be 1..27 bases");
        return false;
      }
      unsigned __int128 space = 1;
      for (uint32_t i = 0; i < g.len; ++i) space *= 5;
      if (space * entries >= ((unsigned __int128)1 << 64) - 1) {
        set_error("unsupported plan: (barcode tuple, random barcode) does not fit a 64-bit key");
        return false;
      }
      P.has_random = 1;
      P.rnd_off = g.off;
      P.rnd_len = g.len;
      P.rspace = (uint64_t)space;
    }
  }
  return true;
}

// The plan of the wave-per-read kernel (bc_long.h): the same group order, key digits and strides as lower(), none of
// its width limits.
bool bc_plan::lower_long(LongHost& out) const {
  if (!unsupported.empty()) {
    set_error("unsupported scheme: " + unsupported);
    return false;
  }
  if (regex_length == 0 || length > 65535u) {
    set_error("unsupported scheme: format length must be 1..65535");
    return false;
  }
  LongPlan& P = out.plan;
  memset(&P, 0, sizeof(P));
  P.L = regex_length;  // bytes a match spans (shorter than format_string only for a token mixing 'N' and 'n')
  P.RL = (uint32_t)regions_string.size();
  P.max_const = max_constant;
  P.no_repair = lowercase_constants ? 1u : 0u;
  P.quality_on = min_quality > 0.0f ? 1u : 0u;
  for (uint32_t p = 0; p < pos.size(); ++p) {
    if (pos[p].kind == kPosConst) {
      out.const_pos.push_back(p);
      out.const_chr.push_back((uint8_t)pos[p].letter);
    } else if (pos[p].kind == kPosFmtN) {
      out.fmtn_pos.push_back(p);
    }
  }
  P.n_const = (uint32_t)out.const_pos.size();
  P.n_fmtn = (uint32_t)out.fmtn_pos.size();
  {
    uint32_t n = 0;
    size_t i = 0;
    const std::string& r = regions_string;
    while (i < r.size()) {
      size_t j = i;
      while (j < r.size() && r[j] == r[i]) ++j;
      if (r[i] != 'C' && j < r.size()) {
        if (n >= (uint32_t)kMaxRuns) {
          set_error("unsupported scheme: more than " + std::to_string(kMaxRuns) + " barcode runs");
          return false;
        }
        P.run_off[n] = (uint32_t)i;
        P.run_len[n] = (uint32_t)(j - i);
        P.run_thr[n] = quality_threshold((uint32_t)(j - i));
        ++n;
      }
      i = j;
    }
    P.n_runs = n;
  }
  struct Pending {
    const FormatGroup* g;
    const KnownSet* set;
    uint32_t max_err;
  };
  std::vector<Pending> order;
  for (const auto& g : groups)
    if (g.type == kGroupSample) order.push_back({&g, &samples, max_sample});
  for (uint32_t b = 1; b <= barcode_num; ++b)
    for (const auto& g : groups)
      if (g.type == kGroupBarcode && g.number == b) order.push_back({&g, &counted[b - 1], max_barcode[b - 1]});
  if (order.size() > (size_t)kMaxGroups) {
    set_error("unsupported scheme: too many barcode groups");
    return false;
  }
  P.n_groups = (uint32_t)order.size();
  out.ref_text.assign(order.size(), {});
  out.ref_off.assign(order.size(), {});
  out.n_samples = sample_barcode ? (uint32_t)samples.size() : 1u;
  P.discard_counts = (!sample_barcode && samples.size() > 0 && !random_barcode) ? 1u : 0u;
  // Do the captures fit the one 64-bit mixed-radix key (index of a known set, base-5 code of a capture kept raw, base-5
  // code of the random barcode)?  If not -- a raw capture or a random barcode above 27 bases, or several that overflow
  // together -- the plan counts under wide keys (bc_long.h).
  bool wide = false;
  {
    // (the very tests lower() makes before it gives a plan to the lane-per-read kernel: both must agree on what a
    // 64-bit key can hold, since an engine may run the two kernels side by side -- reads above 320 bases)
    unsigned __int128 space = 1;
    const unsigned __int128 limit = (unsigned __int128)1 << 63;
    for (const Pending& pd : order) {
      if (pd.set->size() == 0) {
        if (pd.g->len > 27) wide = true;
        for (uint32_t k = 0; k < pd.g->len && space < limit; ++k) space *= 5;
      } else {
        space *= pd.set->size();
      }
      if (space >= limit) wide = true;
    }
    if (random_barcode && !wide)
      for (const auto& g : groups) {
        if (g.type != kGroupRandom) continue;
        if (g.len > 27) {
          wide = true;
          continue;
        }
        unsigned __int128 rs = 1;
        for (uint32_t k = 0; k < g.len; ++k) rs *= 5;
        if (rs * space >= ((unsigned __int128)1 << 64) - 1) wide = true;
      }
  }
  unsigned __int128 entries = 1;
  uint32_t key_bits = 0;
  for (int i = (int)order.size() - 1; i >= 0; --i) {
    const Pending& pd = order[i];
    LongGroup& G = P.groups[i];
    G.type = pd.g->type;
    G.off = pd.g->off;
    G.len = pd.g->len;
    G.n_refs = (uint32_t)pd.set->size();
    G.max_err = pd.max_err;
    G.table_stride = wide ? 0 : (uint64_t)entries;
    G.key_bit = 0;
    if (G.n_refs == 0) {
      P.sparse = 1;
      if (!wide)
        for (uint32_t k = 0; k < G.len; ++k) entries *= 5;
    } else {
      if (!wide) entries *= G.n_refs;
      auto& off = out.ref_off[i];
      auto& text = out.ref_text[i];
      for (const std::string& s : pd.set->seqs) {
        off.push_back((uint32_t)text.size());
        text.insert(text.end(), s.begin(), s.end());
      }
      off.push_back((uint32_t)text.size());
    }
  }
  if (!P.sparse && !wide && entries > ((unsigned __int128)1 << 40)) {
    set_error("unsupported plan: dense counter table above 2^40 entries");
    return false;
  }
  P.wide = wide ? 1u : 0u;
  P.key_words = 1;
  P.rnd_bit = 0;
  if (wide) {
    // payload layout, group after group from bit 0: 32 bits of reference index, or the capture's three bit planes
    P.sparse = 1;  // counts live in a key map (rows are read as text, bc_engine_row_text), whatever the sets are
    for (uint32_t i = 0; i < P.n_groups; ++i) {
      P.groups[i].key_bit = key_bits;
      key_bits += P.groups[i].n_refs ? 32u : 3u * P.groups[i].len;
    }
  }
  out.table_entries = wide ? 0 : (uint64_t)entries;
  if (random_barcode) {
    for (const auto& g : groups) {
      if (g.type != kGroupRandom) continue;
      if (g.len == 0) {
        set_error("unsupported scheme: empty random barcode");
        return false;
      }
      P.has_random = 1;
      P.rnd_off = g.off;
      P.rnd_len = g.len;
      if (wide) {
        P.rnd_bit = key_bits;
        key_bits += 3u * g.len;
        P.rspace = 0;
      } else {
        unsigned __int128 space = 1;
        for (uint32_t i = 0; i < g.len; ++i) space *= 5;
        P.rspace = (uint64_t)space;
      }
    }
  }
  if (wide) {
    P.key_words = 1u + (key_bits + 63u) / 64u;
    if (P.key_words > (uint32_t)kMaxKeyWords) {
      set_error("unsupported plan: the captures kept raw (no conversion file) and the random barcode take " +
                std::to_string(key_bits) + " key bits; the engine's widest key holds " + std::to_string(64 * (kMaxKeyWords - 1)));
      return false;
    }
  }
  return true;
}

// ------------------------------------------------------------------------------------------------
// C ABI: plan
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* bc_version(void) { return "barcode-count-hip 0.1.0 (gfx950)"; }
const char* bc_last_error(void) { return bc::get_error(); }

bc_plan* bc_plan_create(const char* text, size_t len) {
  if (!text) {
    set_error("bc_plan_create: null scheme text");
    return nullptr;
  }
  // info.rs:218-222: drop lines starting with '#', concatenate the rest
  std::string data;
  for (const auto& l : rust_lines(text, len))
    if (l.empty() || l[0] != '#') data += l;

  bc_plan* p = new bc_plan();
  uint32_t off = 0;
  for (size_t i = 0; i < data.size();) {
    size_t tl = 0;
    const TokKind k = lex(data, i, tl);
    if (k == kTokNone) {
      ++i;
      continue;
    }
    const std::string tok = data.substr(i, tl);
    i += tl;
    if (k == kTokSample || k == kTokBarcode || k == kTokRandom) {
      const uint32_t digits = (uint32_t)strtoul(tok.c_str() + 1, nullptr, 10);  // info.rs:252-259
      std::string name;
      char region;
      uint32_t type;
      if (k == kTokSample) {
        if (p->sample_barcode) {  // second (?P<sample>..): Regex::new fails (info.rs:308)
          set_error("sequence format: duplicate capture group name 'sample'");
          delete p;
          return nullptr;
        }
        p->sample_barcode = true;
        p->sample_length = (int32_t)digits;
        name = "sample";
        region = 'S';
        type = kGroupSample;
      } else if (k == kTokBarcode) {
        p->barcode_num++;
        p->barcode_lengths.push_back(digits);
        name = "barcode" + std::to_string(p->barcode_num);
        region = 'B';
        type = kGroupBarcode;
      } else {
        if (p->random_barcode) {
          set_error("sequence format: duplicate capture group name 'random'");
          delete p;
          return nullptr;
        }
        p->random_barcode = true;
        name = "random";
        region = 'R';
        type = kGroupRandom;
      }
      p->regex_string += "(?P<" + name + ">.{" + std::to_string(digits) + "})";  // info.rs:263-267
      p->groups.push_back({type, k == kTokBarcode ? p->barcode_num : 0u, off, digits});
      for (uint32_t d = 0; d < digits; ++d) {  // info.rs:283-286
        p->regions_string.push_back(region);
        p->format_string.push_back('N');
        p->pos.push_back({kPosGroup, 0, (int)p->groups.size() - 1});
      }
      off += digits;
    } else if (tok.find('N') != std::string::npos) {
      // info.rs:287-295: only the upper-case N's are counted, nothing goes to regions_string
      const uint32_t n = (uint32_t)std::count(tok.begin(), tok.end(), 'N');
      p->regex_string += "[AGCT]{" + std::to_string(n) + "}";
      p->format_string += tok;
      // A token that mixes 'N' and 'n' ("NnN"): the regex asks for as many valid bases as there are UPPER-case N's
      // while format_string -- what a repair compares windows with and rebuilds the read from -- keeps the whole token
      // (info.rs:287-295): the format is longer than what the regex matches.  Anchoring follows the regex; whether a
      // repair can ever succeed is settled once the scheme is complete (below).
      if (n != tok.size()) p->mixed_n_token = true;
      for (uint32_t d = 0; d < n; ++d) p->pos.push_back({kPosFmtN, 0, -1});
      off += n;
    } else {
      // info.rs:296-305: constant region; the regex gets the upper-cased letters
      for (char c : tok) {
        const char u = (char)(c & ~0x20);
        p->regex_string.push_back(u);
        p->regions_string.push_back('C');
        p->pos.push_back({kPosConst, u, -1});
        // A lower-case constant: the regex has the upper-case letter (info.rs:298), format_string keeps the lower-case
        // one (info.rs:299).  Anchoring is unaffected; a repair can never succeed, because the repaired read carries
        // the lower-case letter (parse.rs:270-283), which the regex then does not match (parse.rs:92-95).
        if (c != u) p->lowercase_constants = true;
        // (a run of lower-case n's lands here too -- the token pattern is case-insensitive, contains('N') is not:
        // the regex then wants literal 'N' bases.  Only the wave-per-read kernel compares letters as they are.)
        if (u == 'N') p->literal_n_constant = true;
      }
      p->format_string += tok;
      p->constant_region_length += (uint32_t)tok.size();
      off += (uint32_t)tok.size();
    }
  }
  p->length = (uint32_t)p->format_string.size();  // info.rs:307
  p->regex_length = (uint32_t)p->pos.size();
  if (p->mixed_n_token) {
    // fix_constant_region replaces the read by the best window with every non-'N' format character written over it
    // (parse.rs:270-283) -- the lower-case n's among them -- and the regex is then searched in THAT string
    // (parse.rs:92-95), at any offset o <= length - regex_length.  A repair can never succeed when at every such
    // offset some regex position meets a character the format has put there and does not accept it; then the scheme
    // behaves like one with lower-case constants: anchoring only.  Otherwise a repaired read could match depending on
    // its bases at a shifted offset: not reproduced, the scheme is refused.
    const std::string& F = p->format_string;
    bool never = true;
    for (uint32_t o = 0; o + p->regex_length <= p->length && never; ++o) {
      bool rejected = false;
      for (uint32_t q = 0; q < p->regex_length && !rejected; ++q) {
        const char c = F[o + q];
        if (c == 'N') continue;  // the read's own base: unknown here
        const auto& rp = p->pos[q];
        if (rp.kind == kPosConst) rejected = c != rp.letter;
        else if (rp.kind == kPosFmtN) rejected = !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
        // (a capture group's '.' takes anything)
      }
      never = rejected;
    }
    if (never)
      p->lowercase_constants = true;  // (what that flag stands for: no repair can succeed)
    else
      p->unsupported = "a token mixing 'N' and 'n' whose repaired reads could still match the format at a shifted offset";
  }
  p->counted.resize(p->barcode_num);
  p->recompute_budgets();
  return p;
}

void bc_plan_destroy(bc_plan* p) { delete p; }
const char* bc_plan_format_string(const bc_plan* p) { return p->format_string.c_str(); }
const char* bc_plan_regions_string(const bc_plan* p) { return p->regions_string.c_str(); }
const char* bc_plan_regex_string(const bc_plan* p) { return p->regex_string.c_str(); }
uint32_t bc_plan_length(const bc_plan* p) { return p->length; }
uint32_t bc_plan_constant_region_length(const bc_plan* p) { return p->constant_region_length; }
uint32_t bc_plan_barcode_num(const bc_plan* p) { return p->barcode_num; }
uint32_t bc_plan_barcode_length(const bc_plan* p, uint32_t i) {
  return i < p->barcode_lengths.size() ? p->barcode_lengths[i] : 0;
}
int32_t bc_plan_sample_length(const bc_plan* p) { return p->sample_length; }
int bc_plan_has_random(const bc_plan* p) { return p->random_barcode; }
int bc_plan_has_sample(const bc_plan* p) { return p->sample_barcode; }

int bc_plan_load_sample_csv(bc_plan* p, const char* text, size_t len) {
  // info.rs:364-381: skip the header, first two comma-separated fields, no trimming
  bool first = true;
  for (const auto& l : rust_lines(text, len)) {
    if (first) {
      first = false;
      continue;
    }
    auto f = split_take(l, 2);
    if (f.size() == 2)
      p->samples.insert(f[0], f[1]);
    else
      p->samples.insert("", "");  // collect_tuple() == None (info.rs:374-375)
  }
  return BC_OK;
}

int bc_plan_load_counted_csv(bc_plan* p, const char* text, size_t len) {
  // info.rs:390-433
  std::vector<bool> seen(p->barcode_num, false);
  std::vector<bc::KnownSet> sets(p->barcode_num);
  bool first = true;
  for (const auto& l : rust_lines(text, len)) {
    if (first) {
      first = false;
      continue;
    }
    auto f = split_take(l, 3);
    if (f.size() != 3) f.assign(3, std::string());
    // usize::from_str: optional '+', then digits
    const std::string& t = f[2];
    size_t d0 = (!t.empty() && t[0] == '+') ? 1 : 0;
    bool ok = t.size() > d0;
    for (size_t i = d0; i < t.size(); ++i) ok = ok && t[i] >= '0' && t[i] <= '9';
    if (!ok) {
      set_error("Third column of barcode file contains something other than an integer: " + t);
      return BC_ERR_INVALID;
    }
    const unsigned long v = strtoul(t.c_str() + d0, nullptr, 10);
    if (v == 0 || v > p->barcode_num) {  // `0 - 1` / out-of-bounds index: the reference panics
      set_error("barcode number " + t + " outside 1..=" + std::to_string(p->barcode_num));
      return BC_ERR_INVALID;
    }
    seen[v - 1] = true;
    sets[v - 1].insert(f[0], f[1]);
  }
  for (uint32_t x = 0; x < p->barcode_num; ++x) {
    if (!seen[x]) {  // info.rs:420-431
      set_error("Barcode conversion file missing barcode numers [" + std::to_string(x) + "] in the third column");
      return BC_ERR_INVALID;
    }
  }
  for (uint32_t x = 0; x < p->barcode_num; ++x)
    for (size_t j = 0; j < sets[x].size(); ++j) p->counted[x].insert(sets[x].seqs[j], sets[x].ids[j]);
  p->counted_loaded = true;
  return BC_OK;
}

int bc_plan_add_sample(bc_plan* p, const char* seq, const char* id) {
  p->samples.insert(seq, id);
  return BC_OK;
}

int bc_plan_add_counted(bc_plan* p, uint32_t bi, const char* seq, const char* id) {
  if (bi >= p->barcode_num) {
    set_error("bc_plan_add_counted: barcode index out of range");
    return BC_ERR_INVALID;
  }
  p->counted[bi].insert(seq, id);
  p->counted_loaded = true;
  return BC_OK;
}

uint32_t bc_plan_n_samples(const bc_plan* p) { return (uint32_t)p->samples.size(); }
const char* bc_plan_sample_seq(const bc_plan* p, uint32_t i) {
  return i < p->samples.size() ? p->samples.seqs[i].c_str() : nullptr;
}
const char* bc_plan_sample_id(const bc_plan* p, uint32_t i) {
  return i < p->samples.size() ? p->samples.ids[i].c_str() : nullptr;
}
uint32_t bc_plan_n_counted(const bc_plan* p, uint32_t bi) {
  return bi < p->counted.size() ? (uint32_t)p->counted[bi].size() : 0;
}
const char* bc_plan_counted_seq(const bc_plan* p, uint32_t bi, uint32_t i) {
  return (bi < p->counted.size() && i < p->counted[bi].size()) ? p->counted[bi].seqs[i].c_str() : nullptr;
}
const char* bc_plan_counted_id(const bc_plan* p, uint32_t bi, uint32_t i) {
  return (bi < p->counted.size() && i < p->counted[bi].size()) ? p->counted[bi].ids[i].c_str() : nullptr;
}

void bc_max_seq_errors(int sample_errors, int sample_size, int barcode_errors, const uint16_t* barcode_sizes,
                       uint32_t n_barcodes, int constant_errors, uint16_t constant_region_size, uint16_t* out) {
  // info.rs:499-532: an explicit value wins, otherwise 20 % of the size (integer division)
  out[1] = sample_size >= 0 ? (uint16_t)(sample_errors >= 0 ? sample_errors : sample_size / 5) : (uint16_t)0;
  for (uint32_t i = 0; i < n_barcodes; ++i)
    out[2 + i] = (uint16_t)(barcode_errors >= 0 ? barcode_errors : barcode_sizes[i] / 5);
  out[0] = (uint16_t)(constant_errors >= 0 ? constant_errors : constant_region_size / 5);
}

int bc_plan_set_max_errors(bc_plan* p, int sample_errors, int barcode_errors, int constant_errors) {
  p->opt_sample = sample_errors;
  p->opt_barcode = barcode_errors;
  p->opt_constant = constant_errors;
  p->recompute_budgets();
  return BC_OK;
}
uint32_t bc_plan_max_constant_errors(const bc_plan* p) { return p->max_constant; }
uint32_t bc_plan_max_sample_errors(const bc_plan* p) { return p->max_sample; }
uint32_t bc_plan_max_barcode_errors(const bc_plan* p, uint32_t i) {
  return i < p->max_barcode.size() ? p->max_barcode[i] : 0;
}
int bc_plan_set_min_quality(bc_plan* p, float q) {
  p->min_quality = q;
  return BC_OK;
}
uint32_t bc_plan_quality_threshold(const bc_plan* p, uint32_t run_len) { return p->quality_threshold(run_len); }

uint64_t bc_plan_table_entries(const bc_plan* p) {
  bc::HostDevPlan h;
  if (p->lower(h)) return h.plan.sparse ? 0 : h.table_entries;
  bc::LongHost lh;  // plans only the wave-per-read kernel runs (bc_long.h)
  if (!p->lower_long(lh)) return 0;
  return lh.plan.sparse ? 0 : lh.table_entries;
}

int bc_plan_mode(const bc_plan* p) {
  bc::HostDevPlan h;
  if (p->lower(h)) return h.plan.sparse ? 2 : 1;
  bc::LongHost lh;
  if (!p->lower_long(lh)) return 0;
  return lh.plan.sparse ? 2 : 1;
}

}  // extern "C"
