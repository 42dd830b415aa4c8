// bc_exchange.hpp -- what the ranks of a multi-GPU job exchange at its end, written once over a
// Transport (bc_comm.hpp) and a memory space `Ops`, so that the very same logic runs on device buffers with HIP kernels
// (bc_comm.hip: the product) and on host buffers in the CPU-only test of the exchange (tests/emu/exchange_host.cpp).
//
// The reference keeps ONE Results map for all its workers (info.rs:661-808); per GPU that becomes a private table, and:
//  * no random barcode: counts add, so the dense u32 tables are summed onto the root (reduce_tables);
//  * random barcode: a tuple's count is the number of DISTINCT random barcodes (output.rs:265-270) -- set sizes do not
//    add -- so the (tuple, random) keys are first sent to one owner each (partition_and_exchange_keys), where
//    duplicates across ranks collapse; the per-tuple distinct counts of the owned keys then add like plain counts;
//  * captures kept raw (no conversion file): the (key, count) pairs / keys of every rank go to the root (gather_to_root).
//
// `Ops` provides, for its memory space (pointers below are in that space unless they are std::vectors):
//   void* alloc(size_t bytes) / void release(void*)                       nullptr on failure (message set)
//   int pack_u8(const uint32_t* table, uint64_t n, uint8_t* out, std::vector<uint64_t>& ovf_idx,
//               std::vector<uint32_t>& ovf_val)      out[i] = table[i] where it fits a byte, else 0 and (i, table[i])
//                                                    appended to the host lists
//   int sum_u8(const uint8_t* rows, uint32_t n_rows, uint64_t len, uint32_t* out)    out[i] = sum_r rows[r*len + i]
//   int widen_u8(const uint8_t* src, uint64_t n, uint32_t* dst)                      dst[i] = src[i]
//   int scatter_add(uint32_t* table, const uint64_t* idx, const uint32_t* val, uint64_t m)   host lists in
//   int partition_keys(const uint64_t* keys, uint32_t words, const uint32_t* vals /*or null*/, uint64_t n, int world,
//                      int fixed_owner, uint64_t* keys_out, uint32_t* vals_out, uint64_t* counts /*host, world*/)
//                                                    keys (`words` u64 each; the owner comes from the first word, which
//                                                    for a wide key is a fingerprint of the rest) grouped by owner rank
//                                                    (key_owner(), or fixed_owner >= 0)
//   int table_nonzero(const uint32_t* table, uint64_t n, uint64_t cap, std::vector<uint64_t>& idx,
//                     std::vector<uint32_t>& val, bool& fits)   the non-zero entries as host lists; fits = false (lists
//                                                    unspecified) when there are more than cap
//   int sum_bits(const uint32_t* rows, uint32_t n_rows, uint64_t words, uint64_t len, uint32_t* out)
//                                                    out[i] = number of the n_rows bit maps (words u32 each) with bit i
//   int pack_planes2(const uint32_t* counts, uint64_t len, uint32_t* planes, std::vector<uint64_t>& ovf_idx,
//                    std::vector<uint32_t>& ovf_val, uint64_t cap, bool& fits)
//                                                    planes[w] / planes[words + w] (words = (len + 31) / 32): bits 0 / 1 of
//                                                    the counts below 4; a count of 4 or more: both bits clear and
//                                                    (i, count) in the host lists; fits = false when more than cap such
//   int unpack_planes2(const uint32_t* planes, uint64_t words, uint64_t len, uint32_t* dst)   dst[i] = bit0 + 2 bit1
//   int sync()                                       everything above has finished
#pragma once
#include <algorithm>
#include <vector>

#include "bc_comm.hpp"

namespace bc {

// owner rank of a 64-bit key: any well-mixed function does, as long as every rank uses the same
inline int key_owner(uint64_t key, int world) {
  uint64_t x = key * 0x9E3779B97F4A7C15ull;
  x ^= x >> 32;
  return (int)((x >> 7) % (uint64_t)world);
}

// slice r of a table of n entries cut for `world` ranks: [slice_begin(r), slice_begin(r + 1)), multiples of 64 entries
// (whole words of a bit map, whole uint4s of bytes)
inline uint64_t slice_len(uint64_t n, int world) { return ((n + (uint64_t)world - 1) / (uint64_t)world + 63) & ~63ull; }
inline uint64_t slice_begin(uint64_t n, int world, int r) { return std::min(n, slice_len(n, world) * (uint64_t)r); }

template <class Ops>
struct Scoped {  // releases on every way out
  Ops& ops;
  std::vector<void*> held;
  explicit Scoped(Ops& o) : ops(o) {}
  ~Scoped() {
    for (void* p : held) ops.release(p);
  }
  template <typename T>
  T* get(size_t n) {
    void* p = ops.alloc(n * sizeof(T));
    if (p) held.push_back(p);
    return (T*)p;
  }
};

// overflow pairs (global index, count) to the rank given by owner_of(index): host lists in, host lists out
template <class OwnerFn>
int exchange_pairs(Transport& t, std::vector<uint64_t>& idx, std::vector<uint32_t>& val, OwnerFn owner_of,
                   std::vector<uint64_t>& idx_in, std::vector<uint32_t>& val_in) {
  const int W = t.world;
  std::vector<uint64_t> n_to((size_t)W, 0), n_from((size_t)W, 0);
  for (uint64_t i : idx) n_to[(size_t)owner_of(i)]++;
  int rc = t.exchange_counts(n_to.data(), n_from.data());
  if (rc) return rc;
  // (both data exchanges below are always made, empty or not: every rank must make the same sequence of transport
  // calls, and a message of zero bytes costs nothing)
  idx_in.clear();
  val_in.clear();
  std::vector<uint64_t> start((size_t)W + 1, 0);
  for (int r = 0; r < W; ++r) start[(size_t)r + 1] = start[(size_t)r] + n_to[(size_t)r];
  std::vector<uint64_t> s_idx(idx.size());
  std::vector<uint32_t> s_val(val.size());
  {
    std::vector<uint64_t> at(start.begin(), start.end() - 1);
    for (size_t k = 0; k < idx.size(); ++k) {
      const size_t p = (size_t)at[(size_t)owner_of(idx[k])]++;
      s_idx[p] = idx[k];
      s_val[p] = val[k];
    }
  }
  uint64_t total_in = 0;
  std::vector<uint64_t> sb((size_t)W), rb((size_t)W);
  for (int r = 0; r < W; ++r) total_in += n_from[(size_t)r];
  idx_in.resize(total_in);
  val_in.resize(total_in);
  for (int r = 0; r < W; ++r) {
    sb[(size_t)r] = n_to[(size_t)r] * 8;
    rb[(size_t)r] = n_from[(size_t)r] * 8;
  }
  if ((rc = t.host_all_to_all_v(s_idx.data(), sb.data(), idx_in.data(), rb.data()))) return rc;
  for (int r = 0; r < W; ++r) {
    sb[(size_t)r] = n_to[(size_t)r] * 4;
    rb[(size_t)r] = n_from[(size_t)r] * 4;
  }
  return t.host_all_to_all_v(s_val.data(), sb.data(), val_in.data(), rb.data());
}

// Sum of every rank's dense u32 table onto rank `root`, in place there (the other ranks' tables are left in an
// unspecified state).  On xGMI every GPU has its own link to every other, so this is not a ring:
//   1. all-to-all: rank r receives slice r of every table -- all peers at once, one link each -- as BYTES: one byte per
//      count, the rare count above 255 in a side list that travels separately (exact for any counts, a quarter of the
//      bytes);
//   2. each rank adds up the slices it received (one pass at HBM speed) and the side-list entries addressed to it;
//   3. the summed slices go to the root point to point, again as bytes + side list, one link each.
// `bits` (may be null): two-level counting -- the count of entry i is table[i] + bit i of this map.  After a short job
// nearly every count IS its bit (the table holds the repeats only), and step 1 then sends the bit map's slices as they
// stand -- an eighth of the bytes again, no pack pass -- with the table's few non-zero entries in the side list; used
// when every rank's table is that sparse (at most one entry in 64), else table + bit go into the bytes.  The sums of
// step 3 then travel as two bit planes.
template <class Ops>
int reduce_tables(Transport& t, Ops& ops, uint32_t* table, const uint32_t* bits, uint64_t n, int root, int* form = nullptr) {
  const int W = t.world, me = t.rank;
  if (form) *form = 0;  // 0: bytes both ways; 1: bit-map slices out, bytes to the root; 2: bit-map slices out, bit planes to the root
  if (W <= 1 || n == 0) return 0;
  Scoped<Ops> mem(ops);
  const uint64_t my_begin = slice_begin(n, W, me), my_len = slice_begin(n, W, me + 1) - my_begin;
  const uint64_t cut = slice_len(n, W);
  std::vector<uint64_t> o_idx, i_idx;
  std::vector<uint32_t> o_val, i_val;
  std::vector<uint64_t> sb((size_t)W), rb((size_t)W);
  int rc;
  // which form step 1 takes: every rank must agree
  bool as_bits = false;
  {
    bool fits = false;
    if (bits && (rc = ops.table_nonzero(table, n, std::max<uint64_t>(1024, n / 64), o_idx, o_val, fits))) return rc;
    std::vector<uint64_t> mine((size_t)W, (bits && fits) ? 1u : 0u), theirs((size_t)W, 0);
    if ((rc = t.exchange_counts(mine.data(), theirs.data()))) return rc;
    as_bits = true;
    for (uint64_t v : theirs) as_bits = as_bits && v != 0;
    if (form) *form = as_bits ? 1 : 0;
  }
  uint32_t* part = mem.template get<uint32_t>((size_t)my_len + 4);
  if (!part) return -4;
  uint8_t* packed = nullptr;  // n bytes: step 1's send buffer in the byte form, the root's view in step 3
  uint8_t* recv = nullptr;
  if (as_bits) {
    // 1b. bit-map slices to their owners (slice boundaries are multiples of 64 entries: whole words)
    const uint64_t my_words = (my_len + 31) / 32;
    recv = mem.template get<uint8_t>((size_t)(std::max(my_words * 4 * (uint64_t)W, my_len)) + 16);
    if (!recv) return -4;
    for (int r = 0; r < W; ++r) {
      sb[(size_t)r] = ((slice_begin(n, W, r + 1) - slice_begin(n, W, r) + 31) / 32) * 4;
      rb[(size_t)r] = my_words * 4;
    }
    if ((rc = ops.sync())) return rc;
    if ((rc = t.all_to_all_v(bits, sb.data(), recv, rb.data()))) return rc;
    if ((rc = exchange_pairs(t, o_idx, o_val, [&](uint64_t i) { return (int)(i / cut); }, i_idx, i_val))) return rc;
    // 2b. per entry, how many ranks have its bit
    if ((rc = ops.sum_bits(reinterpret_cast<const uint32_t*>(recv), (uint32_t)W, my_words, my_len, part))) return rc;
  } else {
    // 1. packed slices to their owners
    packed = mem.template get<uint8_t>((size_t)n + 16);
    recv = mem.template get<uint8_t>((size_t)(my_len * (uint64_t)W) + 16);
    if (!packed || !recv) return -4;
    if ((rc = ops.pack_u8(table, n, packed, o_idx, o_val))) return rc;  // (table + bit where a bit map is given to ops)
    for (int r = 0; r < W; ++r) {
      sb[(size_t)r] = slice_begin(n, W, r + 1) - slice_begin(n, W, r);
      rb[(size_t)r] = my_len;
    }
    if ((rc = ops.sync())) return rc;
    if ((rc = t.all_to_all_v(packed, sb.data(), recv, rb.data()))) return rc;
    if ((rc = exchange_pairs(t, o_idx, o_val, [&](uint64_t i) { return (int)(i / cut); }, i_idx, i_val))) return rc;
    // 2. one pass over what arrived
    if ((rc = ops.sum_u8(recv, (uint32_t)W, my_len, part))) return rc;
  }
  if (!i_idx.empty()) {
    for (uint64_t& i : i_idx) i -= my_begin;
    if ((rc = ops.scatter_add(part, i_idx.data(), i_val.data(), i_idx.size()))) return rc;
  }
  // 3. summed slices to the root.  In the bit-map form the sums are small (a tuple seen on every rank counts W): they
  // travel as two bit planes -- a quarter of a byte per count -- with counts of 4 and more in the side list; all ranks
  // vote again, and fall back to bytes together.
  o_idx.clear();
  o_val.clear();
  const uint64_t my_words = (my_len + 31) / 32;
  bool as_planes = false;
  uint32_t* planes = nullptr;
  if (as_bits) {
    planes = mem.template get<uint32_t>((size_t)(2 * my_words) + 4);
    if (!planes) return -4;
    bool fits = false;
    if ((rc = ops.pack_planes2(part, my_len, planes, o_idx, o_val, std::max<uint64_t>(1024, my_len / 64), fits))) return rc;
    std::vector<uint64_t> mine((size_t)W, fits ? 1u : 0u), theirs((size_t)W, 0);
    if ((rc = t.exchange_counts(mine.data(), theirs.data()))) return rc;
    as_planes = true;
    for (uint64_t v : theirs) as_planes = as_planes && v != 0;
    if (form && as_planes) *form = 2;
  }
  if (as_planes) {
    for (uint64_t& i : o_idx) i += my_begin;
    // the root's view: every rank's two planes back to back, rank after rank
    uint32_t* all_planes = nullptr;
    std::vector<uint64_t> words_of((size_t)W);
    uint64_t total_words = 0;
    for (int r = 0; r < W; ++r) {
      words_of[(size_t)r] = (slice_begin(n, W, r + 1) - slice_begin(n, W, r) + 31) / 32;
      total_words += 2 * words_of[(size_t)r];
    }
    if (me == root) {
      all_planes = mem.template get<uint32_t>((size_t)total_words + 4);
      if (!all_planes) return -4;
    }
    for (int r = 0; r < W; ++r) {
      sb[(size_t)r] = r == root ? 2 * my_words * 4 : 0;
      rb[(size_t)r] = me == root ? 2 * words_of[(size_t)r] * 4 : 0;
    }
    if ((rc = ops.sync())) return rc;
    if ((rc = t.all_to_all_v(planes, sb.data(), all_planes ? (void*)all_planes : (void*)planes, rb.data()))) return rc;
    if ((rc = exchange_pairs(t, o_idx, o_val, [&](uint64_t) { return root; }, i_idx, i_val))) return rc;
    if (me == root) {
      uint64_t at = 0;
      for (int r = 0; r < W; ++r) {
        const uint64_t b = slice_begin(n, W, r), len_r = slice_begin(n, W, r + 1) - b;
        if (len_r && (rc = ops.unpack_planes2(all_planes + at, words_of[(size_t)r], len_r, table + b))) return rc;
        at += 2 * words_of[(size_t)r];
      }
      if (!i_idx.empty() && (rc = ops.scatter_add(table, i_idx.data(), i_val.data(), i_idx.size()))) return rc;
    }
    return ops.sync();
  }
  o_idx.clear();
  o_val.clear();
  uint8_t* mine8 = recv;  // (my_len bytes of it: what arrived has been summed)
  if ((rc = ops.pack_u8(part, my_len, mine8, o_idx, o_val))) return rc;
  for (uint64_t& i : o_idx) i += my_begin;
  if (me == root && !packed) {
    packed = mem.template get<uint8_t>((size_t)n + 16);
    if (!packed) return -4;
  }
  for (int r = 0; r < W; ++r) {
    sb[(size_t)r] = r == root ? my_len : 0;
    rb[(size_t)r] = me == root ? slice_begin(n, W, r + 1) - slice_begin(n, W, r) : 0;
  }
  if ((rc = ops.sync())) return rc;
  if ((rc = t.all_to_all_v(mine8, sb.data(), packed ? packed : mine8, rb.data()))) return rc;
  if ((rc = exchange_pairs(t, o_idx, o_val, [&](uint64_t) { return root; }, i_idx, i_val))) return rc;
  if (me == root) {
    if ((rc = ops.widen_u8(packed, n, table))) return rc;
    if (!i_idx.empty() && (rc = ops.scatter_add(table, i_idx.data(), i_val.data(), i_idx.size()))) return rc;
  }
  return ops.sync();
}

// Keys (and, when vals != null, a u32 travelling with each) to their owner ranks: by key_owner(), or all of them to
// `fixed_owner` when that is >= 0.  keys / vals: n entries in the exchange's memory space; a key is `words` u64.  On return *keys_in (and
// *vals_in) hold the n_in entries this rank now owns -- allocated here from `ops`, released by the caller.
template <class Ops>
int exchange_keys(Transport& t, Ops& ops, const uint64_t* keys, uint32_t words, const uint32_t* vals, uint64_t n, int fixed_owner,
                  uint64_t** keys_in, uint32_t** vals_in, uint64_t* n_in) {
  const int W = t.world;
  *keys_in = nullptr;
  if (vals_in) *vals_in = nullptr;
  *n_in = 0;
  Scoped<Ops> mem(ops);
  uint64_t* sorted_k = mem.template get<uint64_t>((size_t)n * words + 2);
  uint32_t* sorted_v = vals ? mem.template get<uint32_t>((size_t)n + 4) : nullptr;
  if (!sorted_k || (vals && !sorted_v)) return -4;
  std::vector<uint64_t> n_to((size_t)W, 0), n_from((size_t)W, 0);
  int rc = ops.partition_keys(keys, words, vals, n, W, fixed_owner, sorted_k, sorted_v, n_to.data());
  if (rc) return rc;
  if ((rc = t.exchange_counts(n_to.data(), n_from.data()))) return rc;
  uint64_t total = 0;
  for (uint64_t v : n_from) total += v;
  uint64_t* got_k = (uint64_t*)ops.alloc((size_t)(total * words + 2) * 8);
  uint32_t* got_v = vals ? (uint32_t*)ops.alloc((size_t)(total + 4) * 4) : nullptr;
  if (!got_k || (vals && !got_v)) {
    if (got_k) ops.release(got_k);
    if (got_v) ops.release(got_v);
    return -4;
  }
  std::vector<uint64_t> sb((size_t)W), rb((size_t)W);
  for (int r = 0; r < W; ++r) {
    sb[(size_t)r] = n_to[(size_t)r] * 8 * words;
    rb[(size_t)r] = n_from[(size_t)r] * 8 * words;
  }
  rc = ops.sync();
  if (!rc) rc = t.all_to_all_v(sorted_k, sb.data(), got_k, rb.data());
  if (!rc && vals) {
    for (int r = 0; r < W; ++r) {
      sb[(size_t)r] = n_to[(size_t)r] * 4;
      rb[(size_t)r] = n_from[(size_t)r] * 4;
    }
    rc = t.all_to_all_v(sorted_v, sb.data(), got_v, rb.data());
  }
  if (rc) {
    ops.release(got_k);
    if (got_v) ops.release(got_v);
    return rc;
  }
  *keys_in = got_k;
  if (vals_in) *vals_in = got_v;
  *n_in = total;
  return 0;
}

}  // namespace bc
