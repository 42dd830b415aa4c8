// tests/emu/exchange_host.cpp -- TEST-ONLY: the exchange logic of a multi-GPU job's end (csrc/bc_exchange.hpp) run on
// HOST buffers by several processes over the message-file transport (csrc/bc_comm.hpp), so that the slicing, the
// byte-packing with its overflow side list, the owner partition of keys and the gather to the root can be checked in
// this GPU-less container.  HostOps below stands in for the HIP kernels of csrc/bc_comm.hip (the product path); nothing
// outside tests/ builds or loads this file.
//   exchange_host <dir> <rank> <world> <case> <n> <seed> <root> <out-file>
// case "tables": rank r builds a pseudo-random u32 table of n entries (a few counts above 255, some huge), the tables
//                are summed onto the root; the root writes the summed table to <out-file> (raw u32).
// case "keys":   rank r builds n pseudo-random 64-bit keys (overlapping between ranks) with a u32 each; they go to
//                their owners (root < 0: hashed; else all to root); every rank writes its received (key, val) pairs,
//                sorted, to <out-file>.<rank>.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../ngs-barcode-count_amd/csrc/bc_exchange.hpp"

namespace bc {
static std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace bc

namespace {

struct HostOps {
  void* alloc(size_t bytes) { return malloc(bytes ? bytes : 16); }
  void release(void* p) { free(p); }
  int sync() { return 0; }
  int pack_u8(const uint32_t* table, uint64_t n, uint8_t* out, std::vector<uint64_t>& oi, std::vector<uint32_t>& ov) {
    oi.clear();
    ov.clear();
    for (uint64_t i = 0; i < n; ++i) {
      const uint32_t c = table[i] + ((table == engine_table && engine_bits) ? (engine_bits[i >> 5] >> (i & 31)) & 1u : 0u);
      if (c <= 255u) {
        out[i] = (uint8_t)c;
      } else {
        out[i] = 0;
        oi.push_back(i);
        ov.push_back(c);
      }
    }
    return 0;
  }
  int sum_u8(const uint8_t* rows, uint32_t n_rows, uint64_t len, uint32_t* out) {
    for (uint64_t i = 0; i < len; ++i) {
      uint32_t a = 0;
      for (uint32_t r = 0; r < n_rows; ++r) a += rows[(uint64_t)r * len + i];
      out[i] = a;
    }
    return 0;
  }
  // (the test's stand-in for an engine's bit map beside its table: count = table + bit)
  const uint32_t* engine_table = nullptr;
  const uint32_t* engine_bits = nullptr;
  int sum_bits(const uint32_t* rows, uint32_t n_rows, uint64_t words, uint64_t len, uint32_t* out) {
    for (uint64_t i = 0; i < len; ++i) {
      uint32_t a = 0;
      for (uint32_t r = 0; r < n_rows; ++r) a += (rows[(uint64_t)r * words + (i >> 5)] >> (i & 31)) & 1u;
      out[i] = a;
    }
    return 0;
  }
  int table_nonzero(const uint32_t* table, uint64_t n, uint64_t cap, std::vector<uint64_t>& idx, std::vector<uint32_t>& val, bool& fits) {
    idx.clear();
    val.clear();
    for (uint64_t i = 0; i < n; ++i)
      if (table[i]) {
        idx.push_back(i);
        val.push_back(table[i]);
      }
    fits = idx.size() <= cap;
    return 0;
  }
  int pack_planes2(const uint32_t* counts, uint64_t len, uint32_t* planes, std::vector<uint64_t>& idx, std::vector<uint32_t>& val,
                   uint64_t cap, bool& fits) {
    idx.clear();
    val.clear();
    const uint64_t words = (len + 31) / 32;
    for (uint64_t w = 0; w < 2 * words; ++w) planes[w] = 0;
    for (uint64_t i = 0; i < len; ++i) {
      const uint32_t c = counts[i];
      if (c < 4u) {
        planes[i >> 5] |= (c & 1u) << (i & 31);
        planes[words + (i >> 5)] |= (c >> 1) << (i & 31);
      } else {
        idx.push_back(i);
        val.push_back(c);
      }
    }
    fits = idx.size() <= cap;
    return 0;
  }
  int unpack_planes2(const uint32_t* planes, uint64_t words, uint64_t len, uint32_t* dst) {
    for (uint64_t i = 0; i < len; ++i)
      dst[i] = ((planes[i >> 5] >> (i & 31)) & 1u) + 2u * ((planes[words + (i >> 5)] >> (i & 31)) & 1u);
    return 0;
  }
  int widen_u8(const uint8_t* src, uint64_t n, uint32_t* dst) {
    for (uint64_t i = 0; i < n; ++i) dst[i] = src[i];
    return 0;
  }
  int scatter_add(uint32_t* table, const uint64_t* idx, const uint32_t* val, uint64_t m) {
    for (uint64_t k = 0; k < m; ++k) table[idx[k]] += val[k];
    return 0;
  }
  int partition_keys(const uint64_t* keys, uint32_t words, const uint32_t* vals, uint64_t n, int world, int fixed_owner, uint64_t* ko,
                     uint32_t* vo, uint64_t* counts) {
    std::vector<uint64_t> at((size_t)world, 0);
    for (int r = 0; r < world; ++r) counts[r] = 0;
    auto own = [&](uint64_t k) { return fixed_owner >= 0 ? fixed_owner : bc::key_owner(k, world); };
    for (uint64_t i = 0; i < n; ++i) counts[own(keys[i * words])]++;
    for (int r = 1; r < world; ++r) at[(size_t)r] = at[(size_t)r - 1] + counts[r - 1];
    for (uint64_t i = 0; i < n; ++i) {
      const uint64_t p = at[(size_t)own(keys[i * words])]++;
      for (uint32_t w = 0; w < words; ++w) ko[p * words + w] = keys[i * words + w];
      if (vals) vo[p] = vals[i];
    }
    return 0;
  }
};

uint64_t mix(uint64_t x) {
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

}  // namespace

// what rank r's table holds at entry i (the test recomputes it in numpy)
static uint32_t table_value(uint64_t seed, int r, uint64_t i) {
  const uint64_t h = mix(seed * 1000003ull + (uint64_t)r * 7919ull + i);
  const uint32_t kind = (uint32_t)(h & 1023u);
  if (kind < 700) return 0;                                   // most tuples are never seen
  if (kind < 1000) return (uint32_t)((h >> 10) & 7u) + 1u;    // small counts
  if (kind < 1020) return 200u + (uint32_t)((h >> 10) & 127u); // around the byte limit: 200..327
  return 0xFFFFFF00u + (uint32_t)((h >> 10) & 255u);          // sums that wrap u32 must wrap the same way everywhere
}

int main(int argc, char** argv) {
  if (argc < 9) return 2;
  const std::string dir = argv[1], kind = argv[4], out = argv[8];
  const int rank = atoi(argv[2]), world = atoi(argv[3]), root = atoi(argv[7]);
  const uint64_t n = strtoull(argv[5], nullptr, 0), seed = strtoull(argv[6], nullptr, 0);
  bc::HostDirTransport t(dir, rank, world);
  HostOps ops;
  int rc = 0;
  if (kind == "tables") {
    // argv[9]: "" plain tables; "bits": the counts split into a first-occurrence bit map and a table that is nearly
    // empty (the exchange's bit-map form); "bits_dense": split the same way but with tables too full for that form on
    // the odd ranks, so that every rank has to fall back to bytes together
    const std::string how = argc > 9 ? argv[9] : "";
    std::vector<uint32_t> table(n + 4), bits((n + 31) / 32 + 4, 0u);
    for (uint64_t i = 0; i < n; ++i) {
      uint32_t c = table_value(seed, rank, i);
      if (how == "bits") c = (mix(seed ^ (i * 31 + (uint64_t)rank)) % 97 == 0) ? c : (c ? 1u : 0u);  // mostly 0 / 1
      if (!how.empty() && c) {
        const bool dense_here = how == "bits_dense" && (rank & 1);
        if (!dense_here || (mix(i + 5) & 1)) {
          bits[i >> 5] |= 1u << (i & 31);
          c -= 1u;
        }
      }
      table[i] = c;
    }
    if (!how.empty()) {
      ops.engine_table = table.data();
      ops.engine_bits = bits.data();
    }
    int form = 0;
    rc = bc::reduce_tables(t, ops, table.data(), how.empty() ? nullptr : bits.data(), n, root, &form);
    if (rank == root) printf("form %s\n", form == 0 ? "bytes" : (form == 1 ? "bits+bytes" : "bits+planes"));
    uint64_t counters[3] = {(uint64_t)rank + 1, 10, n};
    if (!rc) rc = t.reduce_sum_u64(counters, 3, root);
    if (!rc && rank == root) {
      FILE* f = fopen(out.c_str(), "wb");
      fwrite(table.data(), 4, n, f);
      fwrite(counters, 8, 3, f);
      fclose(f);
    }
  } else if (kind == "keys") {
    // (keys of `words` u64: word 0 as before, the further words derived from it -- a wide key travels whole)
    const uint32_t words = argc > 9 ? (uint32_t)atoi(argv[9]) : 1u;
    std::vector<uint64_t> keys(n * words);
    std::vector<uint32_t> vals(n);
    for (uint64_t i = 0; i < n; ++i) {
      keys[i * words] = mix(seed + (i * 3 + (uint64_t)rank) % (2 * n + 1));  // overlapping between ranks
      for (uint32_t w = 1; w < words; ++w) keys[i * words + w] = mix(keys[i * words] + w);
      vals[i] = (uint32_t)(rank * 1000 + i % 7);
    }
    uint64_t* gk = nullptr;
    uint32_t* gv = nullptr;
    uint64_t n_in = 0;
    rc = bc::exchange_keys(t, ops, keys.data(), words, vals.data(), n, root, &gk, &gv, &n_in);
    if (!rc) {
      std::vector<std::pair<uint64_t, uint32_t>> got(n_in);
      for (uint64_t i = 0; i < n_in; ++i) {
        got[i] = {gk[i * words], gv[i]};
        for (uint32_t w = 1; w < words; ++w)
          if (gk[i * words + w] != mix(gk[i * words] + w)) rc = 9;  // a key arrived torn
      }
      std::sort(got.begin(), got.end());
      FILE* f = fopen((out + "." + std::to_string(rank)).c_str(), "wb");
      for (auto& p : got) {
        fwrite(&p.first, 8, 1, f);
        fwrite(&p.second, 4, 1, f);
      }
      fclose(f);
      ops.release(gk);
      ops.release(gv);
    }
  } else {
    return 2;
  }
  if (rc) fprintf(stderr, "rank %d: status %d: %s\n", rank, rc, bc::g_err.c_str());
  return rc ? 1 : 0;
}
