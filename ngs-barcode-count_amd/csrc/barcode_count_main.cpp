// barcode_count_main.cpp -- the `barcode-count` command line, on top of the C ABI only.
//
// Drop-in for the reference binary: same flags and defaults
// (arguments.rs:27-124), same stdout blocks (main.rs:19-25, 65, 124-164; info.rs:141-172, 313-335,
// 618-659), same output files, headers and stats file (output.rs:74-576; info.rs:811-904).  The
// reader thread and the worker pool of main.rs:69-121 are replaced by bc_fastq_count + the gfx950
// engine.  Row order inside the CSVs is unspecified in the reference (HashMap iteration); here rows
// come out in the engine's index order.  `--threads` is accepted and ignored (no CPU workers exist).
//
// `--gpus N` (no counterpart in the reference, which is one process): the process started by the user never touches
// a GPU; it starts N rank processes of this same program, one per GPU, and passes rank 0's output through.  Every
// rank counts its share of the FASTQ file's records (bc_fastq_count_shard) and the job ends with the one exchange of
// bc_engine_finish_all -- RCCL over xGMI between the GPUs (rank 0's unique id travels through a file in a private
// temporary directory), or, with `--comm host`, message files in that directory (several ranks on one GPU).
#include <errno.h>
#include <fcntl.h>
#include <spawn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <dirent.h>
#include <signal.h>

#include <algorithm>
#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/barcode_count_hip.h"

extern char** environ;

namespace {

struct Args {  // arguments.rs:6-20
  std::string fastq, format, output_dir = "./", prefix;
  bool has_samples = false, has_counted = false;
  std::string sample_barcodes, counted_barcodes;
  int threads = 0;
  bool merge_output = false, enrich = false;
  int barcodes_errors = -1, sample_errors = -1, constant_errors = -1;
  float min_quality = 0.0f;
  int device = 0;
  // several GPUs
  int gpus = 1;
  std::vector<int> devices;        // --devices a,b,..: the HIP device of each rank (default 0 .. gpus-1)
  std::string comm = "rccl";       // --comm rccl | host
  int rank = -1, world = 1;        // (set by the launcher for its rank processes)
  std::string comm_dir;
};

[[noreturn]] void die(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "Error: ");
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
  exit(1);
}

void usage() {
  printf(
      "NGS-Barcode-Count (MI355X engine)\nCounts barcodes located in sequencing data\n\nUSAGE:\n"
      "    barcode-count [FLAGS] [OPTIONS] --fastq <fastq> --sequence-format <format_file>\n\nFLAGS:\n"
      "    -e, --enrich          Create output files of enrichment for single and double synthons/barcodes\n"
      "    -m, --merge-output    Merge sample output counts into a single file.  Not necessary when there is only one sample\n"
      "    -h, --help            Prints help information\n\nOPTIONS:\n"
      "    -c, --counted-barcodes <barcode_file>      Counted barcodes file\n"
      "    -o, --output-dir <dir>                     Directory to output the counts to [default: ./]\n"
      "    -f, --fastq <fastq>                        FastQ file\n"
      "    -q, --sequence-format <format_file>        Sequence format file\n"
      "        --max-errors-counted-barcode <n>       Maximimum number of sequence errors allowed within each counted barcode. Defaults to 20%% of the total.\n"
      "        --max-errors-constant <n>              Maximimum number of sequence errors allowed within constant region. Defaults to 20%% of the total.\n"
      "        --max-errors-sample <n>                Maximimum number of sequence errors allowed within sample barcode. Defaults to 20%% of the total.\n"
      "        --min-quality <min>                    Minimum average read quality score per barcode [default: 0]\n"
      "    -p, --prefix <prefix>                      File prefix name.  THe output will end with '_<sample_name>_counts.csv'\n"
      "    -s, --sample-barcodes <sample_file>        Sample barcodes file\n"
      "    -t, --threads <threads>                    Number of threads (ignored: the GPU engine has no CPU workers)\n"
      "        --device <id>                          HIP device to run on [default: 0]\n"
      "        --gpus <n>                             Run on n GPUs: one process per GPU, reads sharded, one exchange at the end [default: 1]\n"
      "        --devices <a,b,..>                     The HIP device of each of the n ranks [default: 0,1,..]\n"
      "        --comm <rccl|host>                     How the ranks exchange their results: RCCL over xGMI, or files in a temporary directory [default: rccl]\n");
}

int parse_u16(const char* what, const char* v) {
  char* end;
  const long x = strtol(v, &end, 10);
  if (*v == 0 || *end != 0 || x < 0 || x > 65535) die("Unable to convert %s to an integer", what);
  return (int)x;
}

Args parse_args(int argc, char** argv) {
  Args a;
  {
    char today[32];
    time_t t = time(nullptr);
    strftime(today, sizeof today, "%Y-%m-%d", localtime(&t));  // arguments.rs:25
    a.prefix = today;
  }
  bool have_fastq = false, have_format = false;
  for (int i = 1; i < argc; ++i) {
    std::string k = argv[i];
    std::string v;
    auto eq = k.find('=');
    bool inline_val = false;
    if (k.rfind("--", 0) == 0 && eq != std::string::npos) {
      v = k.substr(eq + 1);
      k = k.substr(0, eq);
      inline_val = true;
    }
    auto val = [&]() -> std::string {
      if (inline_val) return v;
      if (i + 1 >= argc) die("The argument '%s' requires a value", k.c_str());
      return argv[++i];
    };
    if (k == "-h" || k == "--help") {
      usage();
      exit(0);
    } else if (k == "-f" || k == "--fastq") {
      a.fastq = val();
      have_fastq = true;
    } else if (k == "-q" || k == "--sequence-format") {
      a.format = val();
      have_format = true;
    } else if (k == "-s" || k == "--sample-barcodes") {
      a.sample_barcodes = val();
      a.has_samples = true;
    } else if (k == "-c" || k == "--counted-barcodes") {
      a.counted_barcodes = val();
      a.has_counted = true;
    } else if (k == "-t" || k == "--threads") {
      a.threads = parse_u16("threads", val().c_str());
    } else if (k == "-o" || k == "--output-dir") {
      a.output_dir = val();
    } else if (k == "-p" || k == "--prefix") {
      a.prefix = val();
    } else if (k == "-m" || k == "--merge-output") {
      a.merge_output = true;
    } else if (k == "-e" || k == "--enrich") {
      a.enrich = true;
    } else if (k == "--max-errors-counted-barcode") {
      a.barcodes_errors = parse_u16("maximum barcode errors", val().c_str());
    } else if (k == "--max-errors-sample") {
      a.sample_errors = parse_u16("maximum sample errors", val().c_str());
    } else if (k == "--max-errors-constant") {
      a.constant_errors = parse_u16("maximum constant errors", val().c_str());
    } else if (k == "--min-quality") {
      const std::string s = val();
      char* end;
      a.min_quality = strtof(s.c_str(), &end);
      if (s.empty() || *end != 0) die("Unable to convert min score to a float");
    } else if (k == "--device") {
      a.device = parse_u16("device", val().c_str());
    } else if (k == "--gpus") {
      a.gpus = parse_u16("gpus", val().c_str());
      if (a.gpus < 1 || a.gpus > 64) die("--gpus must be between 1 and 64");
    } else if (k == "--devices") {
      const std::string list = val();
      size_t at = 0;
      while (at <= list.size()) {
        const size_t c = list.find(',', at);
        a.devices.push_back(parse_u16("device", list.substr(at, c == std::string::npos ? std::string::npos : c - at).c_str()));
        if (c == std::string::npos) break;
        at = c + 1;
      }
    } else if (k == "--comm") {
      a.comm = val();
      if (a.comm != "rccl" && a.comm != "host") die("--comm must be rccl or host");
    } else if (k == "--bc-rank") {  // (the three below are how the launcher tells a rank process who it is)
      a.rank = parse_u16("rank", val().c_str());
    } else if (k == "--bc-world") {
      a.world = parse_u16("world", val().c_str());
    } else if (k == "--bc-comm-dir") {
      a.comm_dir = val();
    } else {
      die("Found argument '%s' which wasn't expected, or isn't valid in this context", k.c_str());
    }
  }
  if (!have_fastq || !have_format) die("The following required arguments were not provided: --fastq <fastq> --sequence-format <format_file>");
  if (!a.devices.empty() && (int)a.devices.size() != a.gpus) die("--devices names %zu devices for --gpus %d", a.devices.size(), a.gpus);
  return a;
}

std::string read_file(const std::string& path, const char* ctx) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) die("%s %s", ctx, path.c_str());
  std::string s;
  char buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
  fclose(f);
  return s;
}

// num-format Locale::en
std::string commas(uint64_t v) {
  std::string s = std::to_string(v), out;
  for (size_t i = 0; i < s.size(); ++i) {
    if (i && (s.size() - i) % 3 == 0) out.push_back(',');
    out.push_back(s[i]);
  }
  return out;
}

// f32 Display: shortest decimal that round-trips, no trailing ".0"
std::string f32_display(float v) {
  char buf[64];
  for (int prec = 1; prec <= 9; ++prec) {
    snprintf(buf, sizeof buf, "%.*g", prec, (double)v);
    if (strtof(buf, nullptr) == v) break;
  }
  std::string s = buf;
  if (s.find('e') != std::string::npos) {  // Rust never uses exponents for Display
    snprintf(buf, sizeof buf, "%f", (double)v);
    s = buf;
    while (s.find('.') != std::string::npos && (s.back() == '0' || s.back() == '.')) {
      const bool dot = s.back() == '.';
      s.pop_back();
      if (dot) break;
    }
  }
  return s;
}

double now_ms() {
  struct timeval tv;
  gettimeofday(&tv, nullptr);
  return tv.tv_sec * 1000.0 + tv.tv_usec / 1000.0;
}

std::string elapsed_text(double ms_total) {  // main.rs:128-134, output.rs:579-588
  const long long ms = (long long)ms_total;
  const long long secs = ms / 1000;
  char buf[128];
  snprintf(buf, sizeof buf, "%lld hours, %lld minutes, %lld.%03lld seconds", secs / 3600, (secs / 60) % 60, secs % 60,
           ms - secs * 1000);
  return buf;
}

std::string vec_debug(const std::vector<uint32_t>& v) {  // Rust {:?} of a Vec<u16>
  std::string s = "[";
  for (size_t i = 0; i < v.size(); ++i) s += (i ? ", " : "") + std::to_string(v[i]);
  return s + "]";
}

struct Run {
  Args args;
  bc_plan* plan = nullptr;
  bc_engine* engine = nullptr;
  uint32_t barcode_num = 0;
  std::vector<std::pair<std::string, std::string>> samples;               // (sequence, id) in index order
  std::vector<std::vector<std::pair<std::string, std::string>>> counted;  // per barcode position
  // Results (info.rs:661-665) rebuilt from the engine's rows: sample key -> (tuple of sequences -> count)
  std::vector<std::string> sample_keys;
  struct Row {
    std::string code;     // "b1,b2,.." as sequences (the key Results holds)
    std::string written;  // the same converted to IDs when a counted-barcode file was given (output.rs:282-287)
    uint64_t count;
  };
  std::unordered_map<std::string, std::vector<Row>> results;
  std::unordered_map<std::string, std::unordered_map<std::string, uint64_t>> results_map;  // for the merged file
  std::vector<std::unordered_map<std::string, std::string>> counted_map;                  // sequence -> ID
  // WriteFiles state (output.rs:33-46)
  std::unordered_map<std::string, std::map<std::string, uint64_t>> single_hash, double_hash;
  std::unordered_set<std::string> compounds_written;
  std::vector<std::string> output_files;
  std::vector<uint64_t> output_counts;
  uint64_t merged_count = 0;
  std::string merge_text, sample_text;
};

std::string format_display(const Run& r) {  // info.rs:313-335
  const std::string regions = bc_plan_regions_string(r.plan);
  std::string key;
  std::set<char> seen;
  for (char c : regions) {
    if (!seen.insert(c).second) continue;
    if (c == 'S') key += "\nS: Sample barcode";
    if (c == 'B') key += "\nB: Counted barcode";
    if (c == 'C') key += "\nC: Constant region";
    if (c == 'R') key += "\nR: Random barcode";
  }
  return std::string("-FORMAT-\n") + bc_plan_format_string(r.plan) + "\n" + regions + key;
}

std::string max_errors_display(const Run& r) {  // info.rs:618-659
  std::vector<uint32_t> sizes, errs;
  for (uint32_t i = 0; i < r.barcode_num; ++i) {
    sizes.push_back(bc_plan_barcode_length(r.plan, i));
    errs.push_back(bc_plan_max_barcode_errors(r.plan, i));
  }
  std::string size_info, err_info;
  if (sizes.size() > 1) {
    size_info = "Barcode sizes: " + vec_debug(sizes);
    err_info = "Maximum mismatches allowed per barcode sequence: " + vec_debug(errs);
  } else if (sizes.size() == 1) {
    size_info = "Barcode size: " + std::to_string(sizes[0]);
    err_info = "Maximum mismatches allowed per barcode sequence: " + std::to_string(errs[0]);
  }
  const int32_t sl = bc_plan_sample_length(r.plan);
  const std::string dash = "--------------------------------------------------------------\n";
  return "-BARCODE INFO-\nConstant region size: " + std::to_string(bc_plan_constant_region_length(r.plan)) +
         "\nMaximum mismatches allowed per sequence: " + std::to_string(bc_plan_max_constant_errors(r.plan)) + "\n" + dash +
         "Sample barcode size: " + std::to_string(sl < 0 ? 0 : sl) +
         "\nMaximum mismatches allowed per sequence: " + std::to_string(bc_plan_max_sample_errors(r.plan)) + "\n" + dash +
         size_info + "\n" + err_info + "\n" + dash +
         "Minimum allowed average read quality score per barcode: " + f32_display(r.args.min_quality) + "\n";
}

std::string errors_display(const uint64_t c[BC_NCOUNTERS]) {  // info.rs:141-172 (AtomicU32: wraps at 2^32)
  auto g = [&](int k) { return commas((uint32_t)c[k]); };
  return "Correctly matched sequences: " + g(BC_MATCHED) + "\nConstant region mismatches:  " + g(BC_CONSTANT_REGION) +
         "\nSample barcode mismatches:   " + g(BC_SAMPLE_BARCODE) + "\nCounted barcode mismatches:  " + g(BC_BARCODE) +
         "\nDuplicates:                  " + g(BC_DUPLICATES) + "\nLow quality barcodes:        " + g(BC_LOW_QUALITY);
}

std::string sample_name(const Run& r, const std::string& key) {  // output.rs:136-143
  if (r.samples.empty()) return key;
  for (const auto& s : r.samples)
    if (s.first == key) return s.second;
  return "barcode";
}

std::string create_header(const Run& r) {  // output.rs:184-196
  if (r.barcode_num > 1) {
    std::string h = "Barcode_1";
    for (uint32_t n = 1; n < r.barcode_num; ++n) h += ",Barcode_" + std::to_string(n + 1);
    return h;
  }
  return "Barcode";
}

std::vector<std::string> split_commas(const std::string& s) {
  std::vector<std::string> out;
  size_t a = 0;
  for (size_t i = 0; i <= s.size(); ++i) {
    if (i == s.size() || s[i] == ',') {
      out.emplace_back(s, a, i - a);
      a = i + 1;
    }
  }
  return out;
}

std::string convert_code(const Run& r, const std::string& code) {  // output.rs:591-599
  const auto parts = split_commas(code);
  std::string out;
  for (size_t b = 0; b < parts.size(); ++b) {
    std::string id = parts[b];
    if (b < r.counted_map.size()) {
      auto it = r.counted_map[b].find(parts[b]);
      if (it != r.counted_map[b].end()) id = it->second;
    }
    out += (b ? "," : "") + id;
  }
  return out;
}

void add_single(Run& r, const std::string& sample, const std::string& barcode_string, uint64_t count) {  // info.rs:840-866
  const auto parts = split_commas(barcode_string);
  for (size_t index = 0; index < parts.size(); ++index) {
    std::string s;
    for (size_t x = 0; x < parts.size(); ++x) {
      if (x == index) s += parts[index];
      if (x != parts.size() - 1) s.push_back(',');
    }
    auto it = r.single_hash.find(sample);
    if (it != r.single_hash.end()) it->second[s] += count;  // else: the add lands in a temporary (info.rs:862)
  }
}

void add_double(Run& r, const std::string& sample, const std::string& barcode_string, uint64_t count) {  // info.rs:869-904
  const auto parts = split_commas(barcode_string);
  const size_t n = parts.size();
  for (size_t first = 0; first + 1 < n; ++first) {
    for (size_t add = 1; add < n - first; ++add) {
      std::string s;
      for (size_t col = 0; col < n; ++col) {
        if (col == first)
          s += parts[first];
        else if (col == first + add)
          s += parts[first + add];
        if (col != n - 1) s.push_back(',');
      }
      auto it = r.double_hash.find(sample);
      if (it != r.double_hash.end()) it->second[s] += count;
    }
  }
}

enum Enriched { kSingle, kDouble, kFull };

// add_counts_string (output.rs:199-361)
uint64_t add_counts_string(Run& r, const std::string& sample, const std::vector<std::string>& samples, Enriched type) {
  std::vector<Run::Row> enriched_rows;
  const std::unordered_map<std::string, std::map<std::string, uint64_t>> holder =
      type == kSingle ? r.single_hash : (type == kDouble ? r.double_hash : decltype(r.single_hash)());
  if (type != kFull)
    for (const auto& kv : holder.at(sample)) enriched_rows.push_back({kv.first, kv.first, kv.second});
  const std::vector<Run::Row>& rows = type == kFull ? r.results[sample] : enriched_rows;
  uint64_t barcode_num = 0;
  for (const auto& row : rows) {
    const std::string& code = row.code;
    const uint64_t count = row.count;
    ++barcode_num;
    if (barcode_num % 50000 == 0) {
      printf("Barcodes counted: %s\r", commas(barcode_num).c_str());
      fflush(stdout);
    }
    const std::string& written = row.written;
    if (r.args.merge_output) {
      if (r.compounds_written.insert(code).second) {
        r.merged_count++;
        std::string merged_row = written;
        for (const auto& sb : samples) {
          uint64_t c = 0;
          if (type == kFull) {
            auto ms = r.results_map.find(sb);
            if (ms != r.results_map.end()) {
              auto it = ms->second.find(code);
              if (it != ms->second.end()) c = it->second;
            }
          } else {
            auto it = holder.at(sb).find(code);
            if (it != holder.at(sb).end()) c = it->second;
          }
          merged_row += "," + std::to_string(c);
        }
        r.merge_text += merged_row + "\n";
      }
    }
    r.sample_text += written + "," + std::to_string(count) + "\n";
    if (type == kFull && r.args.enrich) {
      add_single(r, sample, written, count);
      if (r.barcode_num > 2) add_double(r, sample, written, count);
    }
  }
  printf("Barcodes counted: %s\r\n", commas(barcode_num).c_str());
  return barcode_num;
}

void write_file(const Run& r, const std::string& name, const std::string& text) {
  std::string path = r.args.output_dir;
  if (!path.empty() && path.back() != '/') path.push_back('/');
  path += name;
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) die("cannot create %s", path.c_str());
  fwrite(text.data(), 1, text.size(), f);
  fclose(f);
}

std::vector<std::string> ordered_samples(const Run& r, std::vector<std::string> keys) {
  // output.rs:91-97: with a sample file the keys are ordered by sample ID (for the merged columns)
  if (!r.samples.empty())
    std::stable_sort(keys.begin(), keys.end(),
                     [&](const std::string& a, const std::string& b) { return sample_name(r, a) < sample_name(r, b); });
  return keys;
}

void merged_header(Run& r, const std::vector<std::string>& samples, const std::string& header) {
  std::string h = header;
  for (const auto& sb : samples) h += "," + sample_name(r, sb);
  r.merge_text += h + "\n";
}

void write_enriched_files(Run& r, Enriched type) {  // output.rs:364-485
  auto& hash = type == kSingle ? r.single_hash : r.double_hash;
  std::vector<std::string> keys;
  for (const auto& k : r.sample_keys)
    if (hash.count(k)) keys.push_back(k);
  const auto samples = ordered_samples(r, keys);
  const char* descriptor = type == kSingle ? "Single" : "Double";
  std::string header = create_header(r);
  if (r.args.merge_output) merged_header(r, samples, header);
  header += ",Count\n";
  for (const auto& sb : samples) {
    const std::string file_name = r.args.prefix + "_" + sample_name(r, sb) + "_counts." + descriptor + ".csv";
    printf("%s\n", file_name.c_str());
    r.output_files.push_back(file_name);
    r.sample_text += header;
    const uint64_t count = add_counts_string(r, sb, samples, type);
    write_file(r, file_name, r.sample_text);
    r.sample_text.clear();
    r.output_counts.push_back(count);
  }
  if (r.args.merge_output) {
    const std::string merged = r.args.prefix + "_counts.all." + descriptor + ".csv";
    printf("%s\n", merged.c_str());
    r.output_files.push_back(merged);
    write_file(r, merged, r.merge_text);
    printf("Barcodes counted: %s\n", commas(r.merged_count).c_str());
    r.merge_text.clear();
    r.output_counts.insert(r.output_counts.end() - (long)samples.size(), r.merged_count);  // output.rs:478-481
    r.merged_count = 0;
  }
}

void write_counts_files(Run& r) {  // output.rs:74-181
  auto samples = ordered_samples(r, r.sample_keys);
  if (r.args.enrich)
    for (const auto& k : samples) {  // ResultsEnrichment::add_sample_barcodes, info.rs:829-837
      r.single_hash[k];
      r.double_hash[k];
    }
  std::string header = create_header(r);
  if (r.args.merge_output) {
    if (samples.size() == 1) {
      fprintf(stderr, "Merged file cannot be created without multiple sample barcodes\n");
      printf("\n");
      r.args.merge_output = false;
    } else {
      merged_header(r, samples, header);
    }
  }
  header += ",Count\n";
  for (const auto& sb : samples) {
    const std::string file_name = r.args.prefix + "_" + sample_name(r, sb) + "_counts.csv";
    printf("%s\n", file_name.c_str());
    r.output_files.push_back(file_name);
    r.sample_text += header;
    const uint64_t count = add_counts_string(r, sb, samples, kFull);
    write_file(r, file_name, r.sample_text);
    r.sample_text.clear();
    r.output_counts.push_back(count);
  }
  if (r.args.merge_output) {
    const std::string merged = r.args.prefix + "_counts.all.csv";
    printf("%s\n", merged.c_str());
    printf("Barcodes counted: %s\n", commas(r.merged_count).c_str());
    r.output_files.push_back(merged);
    write_file(r, merged, r.merge_text);
    r.merge_text.clear();
    r.output_counts.insert(r.output_counts.begin(), r.merged_count);  // output.rs:171 (the file name went to the back)
    r.merged_count = 0;
  }
  if (r.args.enrich) {
    write_enriched_files(r, kSingle);
    if (r.barcode_num > 2) write_enriched_files(r, kDouble);
  }
}

std::string time_text(time_t t) {
  char buf[64];
  strftime(buf, sizeof buf, "%Y-%m-%d %H:%M:%S", localtime(&t));
  return buf;
}

void write_stats_file(const Run& r, time_t start, double start_ms, const uint64_t counters[BC_NCOUNTERS], uint64_t total_reads) {
  // output.rs:488-576; the file is appended to
  std::string path = r.args.output_dir;
  if (!path.empty() && path.back() != '/') path.push_back('/');
  path += r.args.prefix + "_barcode_stats.txt";
  FILE* f = fopen(path.c_str(), "ab");
  if (!f) die("cannot open %s", path.c_str());
  std::string s;
  s += "-TIME INFORMATION-\nStart: " + time_text(start) + "\nFinish: " + time_text(time(nullptr)) +
       "\nTotal time: " + elapsed_text(now_ms() - start_ms) + "\n\n";
  s += "-INPUT FILES-\nFastq: " + r.args.fastq + "\nFormat: " + r.args.format +
       "\nSamples: " + (r.args.has_samples ? r.args.sample_barcodes : std::string("None")) +
       "\nBarcodes: " + (r.args.has_counted ? r.args.counted_barcodes : std::string("None")) + "\n\n";
  s += format_display(r) + "\n\n";
  s += max_errors_display(r) + "\n";
  s += "-RESULTS-\nTotal sequences:             " + commas((uint32_t)total_reads) + "\n" + errors_display(counters) + "\n\n";
  s += "-OUTPUT FILES-\n";
  for (size_t i = 0; i < r.output_files.size() && i < r.output_counts.size(); ++i)
    s += "File & barcodes counted: " + r.output_files[i] + "\t" + commas(r.output_counts[i]) + "\n";
  s += "\n";
  const std::string& fq = r.args.fastq;
  if (fq.size() >= 2 && fq.compare(fq.size() - 2, 2, "gz") == 0 && (uint32_t)total_reads < 1000000) {
    const char* warning =
        "WARNING: The program may have stopped early with the gzipped file.  Unzip the fastq.gz and rerun the algorithm "
        "on the unzipped fastq file if the number of reads is expected to be above 1,000,000 ";
    printf("\n%s\n\n", warning);
    s += std::string("\n") + warning + "\n";
  }
  s += "--------------------------------------------------------------------------------------------------\n\n\n";
  fwrite(s.data(), 1, s.size(), f);
  fclose(f);
}

void progress(uint64_t total, void*) {  // input.rs:151-159 (printed every 10,000 reads there)
  printf("Total sequences:             %s\r", commas(total).c_str());
  fflush(stdout);
}

// --gpus N: starts the N rank processes and waits for them.  Nothing here touches a GPU (no HIP call, and the engine
// library is not even asked for a device), and nothing is exec'ed by a process that has: the ranks are fresh children.
int launch_ranks(const Args& a, int argc, char** argv) {
  const char* base = getenv("TMPDIR");
  std::string dir = std::string(access("/dev/shm", W_OK) == 0 ? "/dev/shm" : (base && *base ? base : "/tmp")) + "/barcode-count-XXXXXX";
  if (!mkdtemp(&dir[0])) die("cannot create a temporary directory for the ranks: %s", strerror(errno));
  char self[4096];
  const ssize_t sl = readlink("/proc/self/exe", self, sizeof self - 1);
  if (sl <= 0) die("cannot find this program's own path");
  self[sl] = 0;
  std::vector<pid_t> pids;
  for (int r = 0; r < a.gpus; ++r) {
    std::vector<std::string> args(argv, argv + argc);
    args[0] = self;
    const int dev = a.devices.empty() ? r : a.devices[(size_t)r];
    for (const char* extra : {"--bc-rank", "", "--bc-world", "", "--bc-comm-dir", "", "--device", ""}) args.push_back(extra);
    args[args.size() - 7] = std::to_string(r);
    args[args.size() - 5] = std::to_string(a.gpus);
    args[args.size() - 3] = dir;
    args[args.size() - 1] = std::to_string(dev);
    std::vector<char*> av;
    for (auto& s : args) av.push_back(&s[0]);
    av.push_back(nullptr);
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    if (r != 0) posix_spawn_file_actions_addopen(&fa, 1, "/dev/null", O_WRONLY, 0);  // one voice on stdout: rank 0's
    pid_t pid = 0;
    const int rc = posix_spawn(&pid, self, &fa, nullptr, av.data(), environ);
    posix_spawn_file_actions_destroy(&fa);
    if (rc != 0) {
      for (pid_t p : pids) kill(p, SIGTERM);
      die("cannot start rank %d: %s", r, strerror(rc));
    }
    pids.push_back(pid);
  }
  int worst = 0;
  for (size_t r = 0; r < pids.size(); ++r) {
    int status = 0;
    while (waitpid(pids[r], &status, 0) < 0 && errno == EINTR) {
    }
    const int code = WIFEXITED(status) ? WEXITSTATUS(status) : 128 + (WIFSIGNALED(status) ? WTERMSIG(status) : 0);
    if (code != 0) {
      fprintf(stderr, "Error: rank %zu ended with status %d\n", r, code);
      if (worst == 0)  // its peers may be waiting for it in the exchange: do not leave them waiting
        for (size_t q = r + 1; q < pids.size(); ++q) kill(pids[q], SIGTERM);
    }
    worst = std::max(worst, code);
  }
  // whatever is left of the ranks' messages, and the directory
  const std::string cmd_dir = dir;
  if (DIR* d = opendir(cmd_dir.c_str())) {
    while (struct dirent* e = readdir(d))
      if (e->d_name[0] != '.') unlink((cmd_dir + "/" + e->d_name).c_str());
    closedir(d);
  }
  rmdir(cmd_dir.c_str());
  return worst;
}

// a rank's communicator: RCCL (the unique id goes from rank 0 to the others through a file) or message files
bc_comm* make_comm(const Args& a) {
  if (a.comm == "host") {
    bc_comm* c = bc_comm_create_host(a.comm_dir.c_str(), a.rank, a.world);
    if (!c) die("%s", bc_last_error());
    return c;
  }
  unsigned char id[BC_COMM_ID_BYTES];
  const std::string path = a.comm_dir + "/rccl_id";
  if (a.rank == 0) {
    if (bc_comm_unique_id(id)) die("%s", bc_last_error());
    const std::string tmp = path + ".part";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f || fwrite(id, 1, sizeof id, f) != sizeof id || fclose(f) != 0 || rename(tmp.c_str(), path.c_str()) != 0)
      die("cannot write %s", path.c_str());
  } else {
    const double t0 = now_ms();
    for (;;) {
      FILE* f = fopen(path.c_str(), "rb");
      if (f) {
        const size_t n = fread(id, 1, sizeof id, f);
        fclose(f);
        if (n == sizeof id) break;
      }
      if (now_ms() - t0 > 300e3) die("rank 0 did not publish the communicator id");
      usleep(2000);
    }
  }
  bc_comm* c = bc_comm_create(id, a.rank, a.world, a.device);
  if (!c) die("%s", bc_last_error());
  return c;
}

}  // namespace

int main(int argc, char** argv) {
  const double start_ms = now_ms();
  const time_t start = time(nullptr);
  Run r;
  r.args = parse_args(argc, argv);
  if (r.args.gpus > 1 && r.args.rank < 0) return launch_ranks(r.args, argc, argv);
  const bool multi = r.args.rank >= 0 && r.args.world > 1;
  const bool root = !multi || r.args.rank == 0;

  const std::string scheme = read_file(r.args.format, "Failed to open");
  r.plan = bc_plan_create(scheme.data(), scheme.size());
  if (!r.plan) die("%s", bc_last_error());
  r.barcode_num = bc_plan_barcode_num(r.plan);
  printf("%s\n\n", format_display(r).c_str());  // main.rs:19

  if (r.args.enrich && r.barcode_num < 2) {  // main.rs:22-25
    fprintf(stderr, "Fewer than 2 counted barcodes.  Too few for barcode enrichment.  Argument flag is ignored\n");
    r.args.enrich = false;
  }
  if (r.args.has_samples) {
    const std::string t = read_file(r.args.sample_barcodes, "Failed to open");
    if (bc_plan_load_sample_csv(r.plan, t.data(), t.size())) die("%s", bc_last_error());
  }
  if (r.args.has_counted) {
    const std::string t = read_file(r.args.counted_barcodes, "Failed to read");
    if (bc_plan_load_counted_csv(r.plan, t.data(), t.size())) die("%s", bc_last_error());
  }
  bc_plan_set_max_errors(r.plan, r.args.sample_errors, r.args.barcodes_errors, r.args.constant_errors);
  bc_plan_set_min_quality(r.plan, r.args.min_quality);
  printf("%s\n\n", max_errors_display(r).c_str());  // main.rs:65

  r.engine = bc_engine_create(r.plan, r.args.device, nullptr, nullptr);
  if (!r.engine) die("%s", bc_last_error());

  if (r.args.fastq.size() >= 8 && r.args.fastq.compare(r.args.fastq.size() - 8, 8, "fastq.gz") == 0) {
    // input.rs:60-61
    printf("If this program stops reading before the expected number of sequencing reads, unzip the gzipped fastq and rerun.\n\n");
  }
  uint64_t total_reads = 0;
  uint64_t counters[BC_NCOUNTERS];
  uint64_t n_rows = 0;
  if (!multi) {
    if (bc_fastq_count(r.engine, r.args.fastq.c_str(), &total_reads, progress, nullptr)) die("Read Fastq error: %s", bc_last_error());
    if (bc_engine_counters(r.engine, counters)) die("%s", bc_last_error());
  } else {
    // this rank's share of the records, then the job's one exchange; the root goes on to write the job's files
    bc_comm* comm = make_comm(r.args);
    if (bc_fastq_count_shard(r.engine, r.args.fastq.c_str(), (uint32_t)r.args.rank, (uint32_t)r.args.world, &total_reads,
                             root ? progress : nullptr, nullptr))
      die("Read Fastq error: %s", bc_last_error());
    if (bc_comm_sum_u64(comm, &total_reads, 1, 0)) die("%s", bc_last_error());
    if (bc_engine_finish_all(r.engine, comm, 0, counters, &n_rows)) die("%s", bc_last_error());
    if (bc_comm_barrier(comm)) die("%s", bc_last_error());  // nobody leaves while a peer still reads its messages
    bc_comm_destroy(comm);
    if (!root) {
      bc_engine_destroy(r.engine);
      bc_plan_destroy(r.plan);
      return 0;
    }
  }
  printf("Total sequences:             %s\r\n", commas((uint32_t)total_reads).c_str());  // input.rs:85-87
  printf("%s\n\n", errors_display(counters).c_str());                                      // main.rs:124
  printf("Compute time: %s\n\n", elapsed_text(now_ms() - start_ms).c_str());              // main.rs:127-135

  printf("-WRITING COUNTS-\n");
  // Results as the writers see it: every sample key that exists (info.rs:698-719) and its rows
  for (uint32_t i = 0; i < bc_plan_n_samples(r.plan); ++i) r.samples.emplace_back(bc_plan_sample_seq(r.plan, i), bc_plan_sample_id(r.plan, i));
  r.counted.resize(r.args.has_counted ? r.barcode_num : 0);
  for (uint32_t b = 0; b < r.counted.size(); ++b)
    for (uint32_t i = 0; i < bc_plan_n_counted(r.plan, b); ++i)
      r.counted[b].emplace_back(bc_plan_counted_seq(r.plan, b, i), bc_plan_counted_id(r.plan, b, i));
  const bool sample_group = bc_plan_has_sample(r.plan) != 0;
  for (const auto& s : r.samples) r.sample_keys.push_back(s.first);
  if (r.samples.empty() && !sample_group) r.sample_keys.push_back("barcode");
  if (!multi && bc_engine_finish(r.engine, &n_rows)) die("%s", bc_last_error());
  r.counted_map.resize(r.counted.size());
  for (size_t b = 0; b < r.counted.size(); ++b)
    for (const auto& kv : r.counted[b]) r.counted_map[b][kv.first] = kv.second;
  auto add_row = [&](const std::string& key, const std::string& code, const std::string& written, uint64_t cnt) {
    // keys that only exist once a read lands on them: raw sample barcodes (info.rs:742-757) and the
    // "barcode" entry of a random-barcode run with a sample file but no sample group (info.rs:792-801)
    auto it = r.results.find(key);
    if (it == r.results.end()) {
      if (std::find(r.sample_keys.begin(), r.sample_keys.end(), key) == r.sample_keys.end()) r.sample_keys.push_back(key);
      it = r.results.emplace(key, std::vector<Run::Row>()).first;
    }
    it->second.push_back({code, written, cnt});
    if (r.args.merge_output) r.results_map[key][code] = cnt;
  };
  if (bc_plan_mode(r.plan) == 1) {
    // dense plan: rows come as indices into the known sets; the sequence / ID strings are looked up
    const uint32_t nb = r.barcode_num ? r.barcode_num : 1;
    const uint64_t block = 1u << 20;
    std::vector<uint32_t> sidx(block), bidx(block * nb);
    std::vector<uint64_t> cnt(block);
    for (uint64_t first = 0; first < n_rows; first += block) {
      const uint64_t n = std::min(block, n_rows - first);
      if (bc_engine_rows(r.engine, first, n, sidx.data(), bidx.data(), cnt.data())) die("%s", bc_last_error());
      for (uint64_t i = 0; i < n; ++i) {
        std::string code, written;
        for (uint32_t b = 0; b < r.barcode_num; ++b) {
          const auto& kv = r.counted[b][bidx[i * nb + b]];
          if (b) {
            code.push_back(',');
            written.push_back(',');
          }
          code += kv.first;
          written += kv.second;
        }
        add_row(sample_group ? r.samples[sidx[i]].first : std::string("barcode"), code, written, cnt[i]);
      }
    }
  } else {
    for (uint64_t i = 0; i < n_rows; ++i) {
      char sample[64], tuple[2048];
      uint64_t cnt = 0;
      if (bc_engine_row_text(r.engine, i, sample, sizeof sample, tuple, sizeof tuple, &cnt)) die("%s", bc_last_error());
      add_row(sample, tuple, r.counted.empty() ? std::string(tuple) : convert_code(r, tuple), cnt);
    }
  }
  write_counts_files(r);
  write_stats_file(r, start, start_ms, counters, total_reads);
  printf("\nTotal time: %s\n", elapsed_text(now_ms() - start_ms).c_str());  // main.rs:156-164
  bc_engine_destroy(r.engine);
  bc_plan_destroy(r.plan);
  return 0;
}
