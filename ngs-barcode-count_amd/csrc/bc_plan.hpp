// bc_plan.hpp -- host-side static run description: the reference's SequenceFormat,
// BarcodeConversions and MaxSeqErrors (info.rs:176-659) plus their lowering to bc::DevPlan.
#pragma once
#include <stdint.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "bc_device_plan.h"
#include "bc_long.h"

namespace bc {

void set_error(const std::string& msg);
const char* get_error();

// HashMap<String, String> of info.rs:339/341 with a stable index per distinct sequence
struct KnownSet {
  std::vector<std::string> seqs;
  std::vector<std::string> ids;
  std::unordered_map<std::string, uint32_t> index;
  void insert(const std::string& seq, const std::string& id) {
    auto it = index.find(seq);
    if (it == index.end()) {
      index.emplace(seq, (uint32_t)seqs.size());
      seqs.push_back(seq);
      ids.push_back(id);
    } else {
      ids[it->second] = id;  // HashMap::insert: the later ID wins (info.rs:378, 418)
    }
  }
  size_t size() const { return seqs.size(); }
};

enum PosKind : uint8_t { kPosConst = 0, kPosFmtN = 1, kPosGroup = 2 };

struct FormatPos {
  uint8_t kind;
  char letter;  // upper-case constant
  int group;
};

struct FormatGroup {
  uint32_t type;    // GroupType
  uint32_t number;  // 1-based barcode number
  uint32_t off, len;
};

// host arrays of one known set, ready for upload
struct HostSet {
  std::vector<uint32_t> r1, r2, rn;
  std::vector<uint8_t> rlen;
  std::vector<uint64_t> hkeys;
  std::vector<uint32_t> hvals;
  std::vector<uint32_t> seed_off, seed_list, odd_list, tier_off, tier_list, tier_bkt, seed2_off, seed2_list, seed3_off, seed3_list;
};

struct HostDevPlan {
  DevPlan plan;                // device pointers still null
  std::vector<HostSet> sets;   // one per plan.groups entry
  std::vector<uint32_t> lhash; // image of the LDS exact-match area (plan.lhash_vec uint4s)
  uint64_t table_entries = 0;
  uint32_t n_samples = 1;
};

// host form of the wave-per-read kernel's plan (bc_long.h): device pointers still null
struct LongHost {
  LongPlan plan;
  std::vector<uint32_t> const_pos, fmtn_pos;
  std::vector<uint8_t> const_chr;
  std::vector<std::vector<uint8_t>> ref_text;   // per plan.groups entry
  std::vector<std::vector<uint32_t>> ref_off;
  uint64_t table_entries = 0;
  uint32_t n_samples = 1;
};

}  // namespace bc

struct bc_plan {
  // SequenceFormat (info.rs:176-187)
  std::string format_string, regions_string, regex_string;
  uint32_t length = 0;  // chars of format_string
  uint32_t constant_region_length = 0;
  uint32_t barcode_num = 0;
  std::vector<uint32_t> barcode_lengths;
  int32_t sample_length = -1;
  bool random_barcode = false, sample_barcode = false;
  std::vector<bc::FormatPos> pos;       // one per byte a match spans
  std::vector<bc::FormatGroup> groups;  // in order of appearance
  std::string unsupported;              // non-empty: why the engine cannot run this scheme
  bool lowercase_constants = false;     // some constant is lower-case in the scheme: no repair can succeed (bc_plan.cpp)
  bool literal_n_constant = false;      // ... and some of them are n's: the regex wants literal 'N' bases
  bool mixed_n_token = false;           // a token mixes 'N' and 'n': the regex is shorter than format_string (bc_plan.cpp)
  uint32_t regex_length = 0;            // bytes one regex match spans (= length unless mixed_n_token)

  // BarcodeConversions (info.rs:338-343)
  bc::KnownSet samples;
  std::vector<bc::KnownSet> counted;
  bool counted_loaded = false;

  // MaxSeqErrors (info.rs:461-472)
  int opt_sample = -1, opt_barcode = -1, opt_constant = -1;
  uint16_t max_constant = 0, max_sample = 0;
  std::vector<uint16_t> max_barcode;
  float min_quality = 0.0f;

  void recompute_budgets();
  // lowers to the device form; false + set_error() when the engine cannot run the plan
  bool lower(bc::HostDevPlan& out) const;
  // the same for the wave-per-read kernel: any read length, group length, reference length and constant budget
  bool lower_long(bc::LongHost& out) const;
  uint32_t quality_threshold(uint32_t run_len) const;
};
