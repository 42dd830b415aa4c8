// bc_fastq.cpp -- FASTQ ingest for the engine.
//
// Replaces the reference's reader thread (input::read_fastq + FastqLineReader, input.rs:24-149):
// same 4-line framing, same "Total sequences" accounting (with its quirks), same first-record
// sanity check (RawSequenceRead::check_fastq_format, parse.rs:377-427) -- but instead of pushing one
// packed String per read onto a mutex-guarded VecDeque it turns whole file chunks into fixed-stride
// byte batches for bc_engine_submit_host (pinned double buffers, hipMemcpyAsync on a side stream).
//
// Structure: a reader thread fills 128 MiB chunks (zlib for .gz, read(2) otherwise) while the
// previous chunk is framed by a team of threads: (A) every thread finds the newlines of its slice,
// (B) a prefix sum gives each newline its line number, (C) the sequence / quality lines are copied
// to their slots of the batch in parallel.  A chunk is cut at a record boundary, so records never
// straddle chunks.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
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

constexpr size_t kChunk = 128u << 20;

struct Source {
  bool gz = false;
  gzFile zf = nullptr;
  int fd = -1;
  // returns bytes read (0 at EOF), -1 on error
  long fill(char* dst, size_t cap) {
    size_t got = 0;
    while (got < cap) {
      long n;
      if (gz)
        n = gzread(zf, dst + got, (unsigned)std::min<size_t>(cap - got, 1u << 30));
      else
        n = read(fd, dst + got, cap - got);
      if (n < 0) return -1;
      if (n == 0) break;
      got += (size_t)n;
    }
    return (long)got;
  }
};

struct Team {
  unsigned n;
  explicit Team(unsigned n_) : n(n_ ? n_ : 1) {}
  template <class F>
  void run(F&& f) {
    if (n == 1) {
      f(0u);
      return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 1; t < n; ++t) th.emplace_back([&f, t] { f(t); });
    f(0u);
    for (auto& x : th) x.join();
  }
};

struct Ingest {
  bc_engine* engine;
  bool gz;
  bool test = true;
  uint64_t total_reads = 0;
  bc_progress_fn progress;
  void* user;
  uint64_t next_progress = 1000000;
  Team team;
  std::vector<std::vector<size_t>> nl;  // per thread: newline offsets of its slice
  std::vector<uint8_t> out_seq, out_qual;
  std::vector<uint16_t> lens;

  Ingest(unsigned threads) : team(threads), nl(threads ? threads : 1) {}

  // frames buf[0, len): `len` ends right after a newline.  Whole records are turned into one batch;
  // the bytes of a trailing incomplete record are left to the caller (returns how many bytes were used).
  int consume(const char* buf, size_t len, size_t* used) {
    *used = 0;
    const unsigned T = team.n;
    const size_t slice = (len + T - 1) / T;
    team.run([&](unsigned t) {  // (A) newline positions
      auto& v = nl[t];
      v.clear();
      const size_t a = std::min(len, t * slice), b = std::min(len, (t + 1) * slice);
      const char* p = buf + a;
      const char* e = buf + b;
      while (p < e) {
        const char* q = (const char*)memchr(p, '\n', (size_t)(e - p));
        if (!q) break;
        v.push_back((size_t)(q - buf));
        p = q + 1;
      }
    });
    std::vector<size_t> first(T + 1, 0);  // (B) global line number of each thread's first newline
    for (unsigned t = 0; t < T; ++t) first[t + 1] = first[t] + nl[t].size();
    const size_t n_lines = first[T];
    const size_t n_rec = n_lines / 4;
    if (n_rec == 0) return BC_OK;
    // end of the last whole record = newline number 4*n_rec - 1
    size_t end_off = 0;
    {
      const size_t g = 4 * n_rec - 1;
      unsigned t = 0;
      while (first[t + 1] <= g) ++t;
      end_off = nl[t][g - first[t]] + 1;
    }
    // start offset of the line that ends at global newline g
    auto prev_end = [&](unsigned t, size_t j) -> size_t {  // offset just past the previous newline
      if (j > 0) return nl[t][j - 1] + 1;
      while (t > 0) {
        --t;
        if (!nl[t].empty()) return nl[t].back() + 1;
      }
      return 0;
    };
    lens.assign(n_rec, 0);
    std::vector<uint16_t> qlens(n_rec, 0);
    std::atomic<int> bad{0};
    std::vector<uint32_t> tmax(T, 0), tmin(T, 0xFFFFFFFFu);
    team.run([&](unsigned t) {  // (C1) line lengths
      const auto& v = nl[t];
      for (size_t j = 0; j < v.size(); ++j) {
        const size_t g = first[t] + j;
        if (g >= 4 * n_rec) break;
        const unsigned k = (unsigned)(g & 3);
        if (k != 1 && k != 3) continue;
        const size_t s = prev_end(t, j);
        size_t n = v[j] - s;
        if (!gz && n && buf[v[j] - 1] == '\r') --n;  // lines() drops "\r\n" (input.rs:44); read_line keeps the '\r'
        if (n > 320) {
          bad = 1;
          n = 320;
        }
        if (k == 1) {
          lens[g >> 2] = (uint16_t)n;
          tmax[t] = std::max<uint32_t>(tmax[t], (uint32_t)n);
          tmin[t] = std::min<uint32_t>(tmin[t], (uint32_t)n);
        } else {
          qlens[g >> 2] = (uint16_t)n;
        }
      }
    });
    if (bad) {
      set_error("a read is longer than 320 bases (not supported by the engine)");
      return BC_ERR_UNSUPPORTED;
    }
    uint32_t max_len = 0, min_len = 0xFFFFFFFFu;
    for (unsigned t = 0; t < T; ++t) {
      max_len = std::max(max_len, tmax[t]);
      min_len = std::min(min_len, tmin[t]);
    }
    if (test) {  // first record only (input.rs:139-142, parse.rs:377-394)
      // lines 1 and 2 of the chunk's first record
      size_t e1 = 0, e2 = 0;
      {
        unsigned t = 0;
        while (first[t + 1] <= 0) ++t;
        e1 = nl[t][0 - first[t]];
        t = 0;
        while (first[t + 1] <= 1) ++t;
        e2 = nl[t][1 - first[t]];
      }
      size_t n1 = e1, n2 = e2 - (e1 + 1);
      if (!gz && n1 && buf[e1 - 1] == '\r') --n1;
      if (!gz && n2 && buf[e2 - 1] == '\r') --n2;
      if (looks_like_sequence(buf, n1)) {
        set_error("The first line within the FASTQ contains DNA sequences.  Check the FASTQ format");
        return BC_ERR_INVALID;
      }
      if (!looks_like_sequence(buf + e1 + 1, n2)) {
        set_error("The second line within the FASTQ file is not a sequence. Check the FASTQ format");
        return BC_ERR_INVALID;
      }
      test = false;
    }
    for (size_t r = 0; r < n_rec; ++r) {
      if (lens[r] != qlens[r]) {
        set_error("read " + std::to_string(total_reads + r + 1) +
                  ": quality line and sequence line differ in length (not supported by the engine)");
        return BC_ERR_UNSUPPORTED;
      }
    }
    const uint32_t stride = std::max<uint32_t>(4, (max_len + 3u) & ~3u);
    const bool uniform = min_len == max_len;
    out_seq.resize(n_rec * (size_t)stride);
    out_qual.resize(n_rec * (size_t)stride);
    team.run([&](unsigned t) {  // (C2) copy the sequence and quality lines to their slots
      const auto& v = nl[t];
      for (size_t j = 0; j < v.size(); ++j) {
        const size_t g = first[t] + j;
        if (g >= 4 * n_rec) break;
        const unsigned k = (unsigned)(g & 3);
        if (k != 1 && k != 3) continue;
        const size_t r = g >> 2;
        const size_t s = prev_end(t, j);
        uint8_t* dst = (k == 1 ? out_seq.data() : out_qual.data()) + r * (size_t)stride;
        const uint32_t n = lens[r];
        memcpy(dst, buf + s, n);
        if (n < stride) memset(dst + n, k == 1 ? 'N' : '!', stride - n);
      }
    });
    const int rc = bc_engine_submit_host(engine, out_seq.data(), out_qual.data(), uniform ? nullptr : lens.data(), stride,
                                         max_len, n_rec);
    if (rc != BC_OK) return rc;
    total_reads += n_rec;
    if (progress && total_reads >= next_progress) {
      progress(total_reads, user);
      next_progress = (total_reads / 1000000 + 1) * 1000000;
    }
    *used = end_off;
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
  Source src;
  src.gz = gz;
  if (gz) {
    src.zf = gzopen(path.c_str(), "rb");  // multi-member aware (flate2 MultiGzDecoder, input.rs:63)
    if (src.zf) gzbuffer(src.zf, 4 << 20);
  } else {
    src.fd = open(path.c_str(), O_RDONLY);
  }
  if ((gz && !src.zf) || (!gz && src.fd < 0)) {
    set_error("Failed to open file: " + path);
    return BC_ERR_INVALID;
  }
  unsigned threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char* ev = getenv("BC_INGEST_THREADS")) threads = (unsigned)std::max(1, atoi(ev));
  Ingest in(threads);
  in.engine = e;
  in.gz = gz;
  in.progress = progress;
  in.user = user;

  // double-buffered chunks: the reader thread fills one while the team frames the other.  Every
  // buffer has kHead bytes of headroom in front of where the reader writes; the unfinished record
  // left over from the previous chunk is copied there, so the data stays contiguous.
  constexpr size_t kHead = 4u << 20;
  std::vector<char> bufs[2];
  bufs[0].resize(kHead + kChunk);
  bufs[1].resize(kHead + kChunk);
  std::mutex mu;
  std::condition_variable cv;
  long filled[2] = {-2, -2};  // -2: not read yet, >= 0: bytes read, -1: error
  bool request[2] = {false, false};
  bool quit = false;
  std::thread reader([&] {
    for (;;) {
      int b = -1;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return quit || request[0] || request[1]; });
        if (quit) return;
        b = request[0] ? 0 : 1;
        request[b] = false;
      }
      const long n = src.fill(bufs[b].data() + kHead, kChunk);
      {
        std::lock_guard<std::mutex> lk(mu);
        filled[b] = n;
      }
      cv.notify_all();
    }
  });
  auto ask = [&](int b) {
    std::lock_guard<std::mutex> lk(mu);
    filled[b] = -2;
    request[b] = true;
    cv.notify_all();
  };
  auto wait_for = [&](int b) -> long {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return filled[b] != -2; });
    return filled[b];
  };

  int rc = BC_OK;
  int cur = 0;
  size_t carry = 0;  // bytes of an unfinished record sitting right in front of bufs[cur][kHead]
  ask(cur);
  std::string tail;  // what is left when the file ends
  for (;;) {
    const long got = wait_for(cur);
    if (got < 0) {
      set_error("read error in " + path);
      rc = BC_ERR_INVALID;
      break;
    }
    char* buf = bufs[cur].data() + kHead - carry;
    const size_t have = carry + (size_t)got;
    if (got == 0) {
      tail.assign(buf, have);
      break;
    }
    const int nxt = cur ^ 1;
    ask(nxt);  // the next chunk is read while this one is framed
    size_t cut = have;  // frame up to the last newline
    while (cut > 0 && buf[cut - 1] != '\n') --cut;
    size_t used = 0;
    if (cut) rc = in.consume(buf, cut, &used);
    const size_t rest = have - used;
    if (rc == BC_OK && rest > kHead) {
      set_error("a FASTQ record is longer than 4 MiB");
      rc = BC_ERR_INVALID;
    }
    if (rc != BC_OK) {
      (void)wait_for(nxt);  // let the reader finish before it is told to quit
      break;
    }
    memcpy(bufs[nxt].data() + kHead - rest, buf + used, rest);  // disjoint from what the reader is writing
    carry = rest;
    cur = nxt;
  }
  {
    std::lock_guard<std::mutex> lk(mu);
    quit = true;
  }
  cv.notify_all();
  reader.join();

  if (rc == BC_OK) {
    // what is left: fewer than four complete lines (+ possibly a last line without '\n')
    size_t lines = 0;
    for (char c : tail) lines += c == '\n';
    const bool partial_line = !tail.empty() && tail.back() != '\n';
    const size_t seen = lines + (partial_line ? 1 : 0);  // lines the reference's reader would have been handed
    if (seen == 4) {
      // a complete record whose quality line lacks the final '\n'
      if (gz) {
        // post() pops the record's last character unconditionally (input.rs:137): here that is the last
        // quality character, so the reference scores a quality line one short of the sequence line
        set_error("gz input without a final newline: the reference drops the last quality character "
                  "(input.rs:137); not supported by the engine");
        rc = BC_ERR_UNSUPPORTED;
      } else {
        std::string t2 = tail + "\n";
        size_t used = 0;
        rc = in.consume(t2.data(), t2.size(), &used);
      }
    } else if (seen > 0) {
      in.total_reads += 1;  // a trailing partial record is counted when its first line is seen (input.rs:128-130)
    }
    if (rc == BC_OK && gz) {
      // the gz loop calls read("") once more at EOF (input.rs:69-73): when that lands on "line 1" the
      // total grows by one (README.md:159 vs 176)
      const size_t line_num_after = seen % 4;  // lines handed over since the last whole record
      if (line_num_after == 0) in.total_reads += 1;
    }
  }
  if (gz)
    gzclose(src.zf);
  else
    close(src.fd);
  if (total_reads) *total_reads = in.total_reads;
  return rc;
}
