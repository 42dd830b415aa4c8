// bc_fastq.cpp -- FASTQ ingest for the engine.
//
// Replaces the reference's reader thread (input::read_fastq + FastqLineReader, input.rs:24-149):
// same 4-line framing, same "Total sequences" accounting (with its quirks), same first-record
// sanity check (RawSequenceRead::check_fastq_format, parse.rs:377-427) -- but instead of pushing one
// packed String per read onto a mutex-guarded VecDeque it fills fixed-stride byte batches that go to
// the GPU through bc_engine_submit_host (pinned double buffers, hipMemcpyAsync on a side stream).
#include <stdio.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/barcode_count_hip.h"
#include "bc_plan.hpp"

using namespace bc;

namespace {

bool ends_with(const std::string& s, const char* suf) {
  const size_t n = strlen(suf);
  return s.size() >= n && memcmp(s.data() + s.size() - n, suf, n) == 0;
}

// test_sequence (parse.rs:414-427): a line is "Sequence" unless fewer than half of its bytes are A,G,C,T,N
bool looks_like_sequence(const char* s, size_t n) {
  size_t dna = 0;
  for (size_t i = 0; i < n; ++i) dna += s[i] == 'A' || s[i] == 'G' || s[i] == 'C' || s[i] == 'T' || s[i] == 'N';
  return !(dna < n / 2);
}

struct Batch {
  std::vector<uint8_t> seq, qual;  // variable-length lines back to back while the batch fills
  std::vector<uint32_t> off;       // start of each read's line in seq / qual
  std::vector<uint16_t> len;
  uint32_t max_len = 0;
  bool uniform = true;
  void clear() {
    seq.clear();
    qual.clear();
    off.clear();
    len.clear();
    max_len = 0;
    uniform = true;
  }
};

struct Framer {
  bc_engine* engine;
  bool gz;           // the gz path keeps a '\r' that precedes '\n' (BufRead::read_line), the plain path drops it
  uint64_t total_reads = 0;
  uint32_t line_num = 0;
  bool test = true;  // first record still to be checked
  std::string l1, l2, l4;
  Batch b;
  std::vector<uint8_t> out_seq, out_qual;
  bc_progress_fn progress;
  void* user;
  uint64_t next_progress = 1000000;

  int flush() {
    const uint64_t n = b.len.size();
    if (n == 0) return BC_OK;
    const uint32_t stride = std::max<uint32_t>(4, (b.max_len + 3u) & ~3u);
    out_seq.assign((size_t)n * stride, (uint8_t)'N');
    out_qual.assign((size_t)n * stride, (uint8_t)'!');
    for (uint64_t i = 0; i < n; ++i) {
      memcpy(&out_seq[(size_t)i * stride], &b.seq[b.off[i]], b.len[i]);
      memcpy(&out_qual[(size_t)i * stride], &b.qual[b.off[i]], b.len[i]);
    }
    const int rc = bc_engine_submit_host(engine, out_seq.data(), out_qual.data(), b.uniform ? nullptr : b.len.data(),
                                         stride, b.max_len, n);
    b.clear();
    return rc;
  }

  // FastqLineReader::read + post (input.rs:115-148); `line` has no terminator
  int feed(const char* line, size_t n) {
    if (++line_num == 5) line_num = 1;
    if (line_num == 1) {
      ++total_reads;  // counted when line 1 is seen (input.rs:128-130)
      if (test) l1.assign(line, n);
      if (progress && total_reads >= next_progress) {
        progress(total_reads, user);
        next_progress += 1000000;
      }
    } else if (line_num == 2) {
      l2.assign(line, n);
    } else if (line_num == 4) {
      if (test) {  // parse.rs:377-394
        if (looks_like_sequence(l1.data(), l1.size())) {
          set_error("The first line within the FASTQ contains DNA sequences.  Check the FASTQ format");
          return BC_ERR_INVALID;
        }
        if (!looks_like_sequence(l2.data(), l2.size())) {
          set_error("The second line within the FASTQ file is not a sequence. Check the FASTQ format");
          return BC_ERR_INVALID;
        }
        test = false;
      }
      if (n != l2.size()) {
        set_error("read " + std::to_string(total_reads) +
                  ": quality line and sequence line differ in length (not supported by the engine)");
        return BC_ERR_UNSUPPORTED;
      }
      if (n > 320) {
        set_error("read " + std::to_string(total_reads) + ": longer than 320 bases (not supported by the engine)");
        return BC_ERR_UNSUPPORTED;
      }
      b.off.push_back((uint32_t)b.seq.size());
      b.seq.insert(b.seq.end(), l2.begin(), l2.end());
      b.qual.insert(b.qual.end(), line, line + n);
      if (!b.len.empty() && (uint32_t)n != b.len[0]) b.uniform = false;
      b.len.push_back((uint16_t)n);
      b.max_len = std::max<uint32_t>(b.max_len, (uint32_t)n);
      if (b.len.size() >= (4u << 20) || b.seq.size() >= (512u << 20)) return flush();
    }
    return BC_OK;
  }
};

}  // namespace

extern "C" int bc_fastq_count(bc_engine* e, const char* fastq_path, uint64_t* total_reads, bc_progress_fn progress,
                              void* user) {
  if (total_reads) *total_reads = 0;
  const std::string path = fastq_path ? fastq_path : "";
  const bool gz = ends_with(path, "fastq.gz");
  if (!gz && !ends_with(path, "fastq")) {  // input.rs:34-39
    set_error("This program only works with *.fastq files and *.fastq.gz files.  The latter is still experimental");
    return BC_ERR_INVALID;
  }
  gzFile f = gzopen(path.c_str(), "rb");  // transparent for plain files, multi-member aware for .gz
  if (!f) {
    set_error("Failed to open file: " + path);
    return BC_ERR_INVALID;
  }
  gzbuffer(f, 1 << 20);
  Framer fr;
  fr.engine = e;
  fr.gz = gz;
  fr.progress = progress;
  fr.user = user;
  std::vector<char> buf(8 << 20);
  std::string carry;
  int rc = BC_OK;
  for (;;) {
    const int got = gzread(f, buf.data(), (unsigned)buf.size());
    if (got < 0) {
      set_error("read error in " + path);
      rc = BC_ERR_INVALID;
      break;
    }
    if (got == 0) break;
    const char* p = buf.data();
    const char* end = p + got;
    while (p < end && rc == BC_OK) {
      const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
      if (!nl) {
        carry.append(p, end);
        break;
      }
      const char* ls = p;
      size_t ln = (size_t)(nl - p);
      if (!carry.empty()) {
        carry.append(p, nl);
        ls = carry.data();
        ln = carry.size();
      }
      // BufRead::lines() drops "\r\n" as well as "\n" (plain path, input.rs:44); read_line keeps the '\r' (gz path)
      if (!gz && ln && ls[ln - 1] == '\r') --ln;
      rc = fr.feed(ls, ln);
      carry.clear();
      p = nl + 1;
    }
    if (rc != BC_OK) break;
  }
  if (rc == BC_OK && !carry.empty()) {
    // last line of the file without '\n'
    size_t ln = carry.size();
    if (!gz && ln && carry[ln - 1] == '\r') --ln;
    if (gz && fr.line_num == 3) {
      // post() pops the record's last character unconditionally (input.rs:137): here that is the last
      // quality character, so the reference scores a quality line one short of the sequence line
      set_error("gz input without a final newline: the reference drops the last quality character "
                "(input.rs:137); not supported by the engine");
      rc = BC_ERR_UNSUPPORTED;
    } else {
      rc = fr.feed(carry.data(), ln);
    }
  }
  if (rc == BC_OK && gz) {
    // the gz loop calls read("") once more at EOF (input.rs:69-73): the line counter advances and,
    // when that starts a "record", the total does too (README.md:159 vs 176)
    if (++fr.line_num == 5) fr.line_num = 1;
    if (fr.line_num == 1) ++fr.total_reads;
  }
  if (rc == BC_OK) rc = fr.flush();
  gzclose(f);
  if (total_reads) *total_reads = fr.total_reads;
  return rc;
}
