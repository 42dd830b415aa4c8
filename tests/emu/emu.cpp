// tests/emu/emu.cpp -- TEST-ONLY host emulation of the lane code in csrc/bc_lane.h.
//
// Runs the SAME per-read logic the gfx950 kernel runs (pack -> anchor -> repair -> quality ->
// barcode lookup), one read at a time on the CPU, so that the bit tricks can be checked against
// the oracle in this GPU-less container.  It is not part of the product: the shipped library has
// no CPU path and nothing outside tests/ builds or loads this file.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../ngs-barcode-count_amd/csrc/bc_lane.h"
#include "../../ngs-barcode-count_amd/csrc/bc_plan.hpp"

namespace {

struct HostOps {
  const uint32_t* qual32 = nullptr;
  const bc::Quad* area = nullptr;
  const bc::Quad* lhash() const { return area; }
  bool tables() const { return area != nullptr; }
  const uint32_t* stage_quality(uint32_t) const { return qual32; }
  void sequence_consumed() const {}
  void mark(int) const {}
  void issued() const {}
  bool any(bool c) const { return c; }
  uint32_t tier_single_n(const bc::DevGroup& G, uint32_t q1, uint32_t q2, uint32_t qn, bool want, bool& settled) const {
    settled = false;
    return want ? bc::tier_lookup_single_n(G, q1, q2, qn, settled) : bc::kFail;
  }
  uint32_t nearest(const bc::DevGroup& G, uint32_t q1, uint32_t q2, uint32_t qn, uint32_t qx, bool need) const {
    if (!need) return bc::kFail;
    bc::Nearest s;
    bc::nearest_init(s);
    for (uint32_t j = 0; j < G.n_refs; ++j) {
      bool ex;
      const uint32_t d = bc::ref_distance(q1, q2, qn, qx, G.len, G.r1()[j], G.r2()[j], G.rn()[j], G.rlen()[j], ex);
      bc::nearest_add(s, d, j, ex);
    }
    return bc::nearest_result(s.key, s.idx, s.count, G.max_err);
  }
};

struct EmuPlan {
  bc::HostDevPlan h;
  std::vector<std::vector<uint32_t>> dtables;
  std::vector<bc::Quad> lhash;  // 16-byte aligned copy of h.lhash
};

template <int NW, int NWW>
void run(const EmuPlan& E, const uint8_t* seq, const uint8_t* qual, const uint16_t* lens, const uint16_t* qlens, uint32_t stride,
         uint32_t read_len, uint64_t n, uint8_t* outcomes, uint64_t* idx, uint64_t* rcode) {
  const uint32_t maxlen = lens ? stride : read_len;
  const uint32_t nd = (maxlen + 3) / 4;
  HostOps ops;
  std::vector<uint32_t> s32((stride + 512) / 4 + 4), q32((stride + 512) / 4 + 4);
  for (uint64_t i = 0; i < n; ++i) {
    // place the read at a rotating byte misalignment to exercise the alignbyte path; every other group of four
    // reads sits on a dword boundary and goes through the instantiation that knows it (kAligned)
    const bool aligned = (i & 4) != 0;
    const uint32_t base = aligned ? 4u * (uint32_t)(i & 3) : (uint32_t)(i & 3);
    memset(s32.data(), 'A', s32.size() * 4);
    memset(q32.data(), 'I', q32.size() * 4);
    memcpy((uint8_t*)s32.data() + base, seq + i * stride, stride);
    if (qual) memcpy((uint8_t*)q32.data() + base, qual + i * stride, stride);
    const uint32_t len = lens ? lens[i] : read_len;
    const uint32_t qlen = qlens ? qlens[i] : len;
    ops.qual32 = q32.data();
    ops.area = reinterpret_cast<const bc::Quad*>(E.lhash.data());
    bc::ReadResult r = aligned ? bc::process_read<HostOps, NW, NWW, true>(E.h.plan, ops, s32.data(), base, len, qlen, nd, true)
                               : bc::process_read<HostOps, NW, NWW, false>(E.h.plan, ops, s32.data(), base, len, qlen, nd, true);
    outcomes[i] = (uint8_t)r.outcome;
    idx[i] = r.dense_idx;
    if (rcode) rcode[i] = r.rcode;
  }
}

}  // namespace

extern "C" {

void* emu_plan_create(const bc_plan* p) {
  EmuPlan* E = new EmuPlan();
  if (!p->lower(E->h)) {
    delete E;
    return nullptr;
  }
  E->dtables.resize(E->h.plan.n_groups);
  E->lhash.resize(E->h.plan.lhash_vec);
  if (!E->lhash.empty()) memcpy(E->lhash.data(), E->h.lhash.data(), E->h.lhash.size() * 4);
  HostOps ops;
  for (uint32_t g = 0; g < E->h.plan.n_groups; ++g) {
    bc::DevGroup& G = E->h.plan.groups[g];
    bc::HostSet& H = E->h.sets[g];
    G.r1_a = (uint64_t)(uintptr_t)H.r1.data();
    G.r2_a = (uint64_t)(uintptr_t)H.r2.data();
    G.rn_a = (uint64_t)(uintptr_t)H.rn.data();
    G.rlen_a = (uint64_t)(uintptr_t)H.rlen.data();
    G.hkeys_a = (uint64_t)(uintptr_t)H.hkeys.data();
    G.hvals_a = (uint64_t)(uintptr_t)H.hvals.data();
    G.tier_off_a = (uint64_t)(uintptr_t)H.tier_off.data();
    G.tier_list_a = (uint64_t)(uintptr_t)H.tier_list.data();
    G.tier_bkt_a = (uint64_t)(uintptr_t)H.tier_bkt.data();
    if (G.mode == bc::kSetDirect) {
      const uint32_t nq = 1u << (2 * G.len);
      E->dtables[g].resize(nq);
      for (uint32_t q = 0; q < nq; ++q) E->dtables[g][q] = bc::dtable_entry(G, q);
      G.dtable_a = (uint64_t)(uintptr_t)E->dtables[g].data();
    }
  }
  return E;
}

void emu_plan_destroy(void* e) { delete (EmuPlan*)e; }

uint64_t emu_table_entries(void* e) { return ((EmuPlan*)e)->h.table_entries; }
int emu_discard_counts(void* e) { return (int)((EmuPlan*)e)->h.plan.discard_counts; }
uint64_t emu_rspace(void* e) { return ((EmuPlan*)e)->h.plan.has_random ? ((EmuPlan*)e)->h.plan.rspace : 0; }

int emu_process2(void* e, const uint8_t* seq, const uint8_t* qual, const uint16_t* lens, const uint16_t* qlens, uint32_t stride,
                 uint32_t read_len, uint64_t n, uint8_t* outcomes, uint64_t* idx, uint64_t* rcode) {
  const EmuPlan& E = *(EmuPlan*)e;
  const uint32_t maxlen = lens ? stride : read_len;
  const uint32_t L = E.h.plan.L;
  const uint32_t nww = maxlen >= L ? (maxlen - L + 1 + 31) / 32 : 1;
#define EMU_RUN(NW_, NWW_) run<NW_, NWW_>(E, seq, qual, lens, qlens, stride, read_len, n, outcomes, idx, rcode)
  if (maxlen <= 128) {
    if (nww <= 1) EMU_RUN(4, 1); else if (nww <= 2) EMU_RUN(4, 2); else EMU_RUN(4, 4);
  } else if (maxlen <= 256) {
    if (nww <= 2) EMU_RUN(8, 2); else if (nww <= 4) EMU_RUN(8, 4); else EMU_RUN(8, 8);
  } else if (maxlen <= 320) {
    if (nww <= 4) EMU_RUN(10, 4); else EMU_RUN(10, 10);
  } else {
    return -1;
  }
#undef EMU_RUN
  return 0;
}

int emu_process(void* e, const uint8_t* seq, const uint8_t* qual, const uint16_t* lens, uint32_t stride,
                uint32_t read_len, uint64_t n, uint8_t* outcomes, uint64_t* idx, uint64_t* rcode) {
  return emu_process2(e, seq, qual, lens, nullptr, stride, read_len, n, outcomes, idx, rcode);
}
}
