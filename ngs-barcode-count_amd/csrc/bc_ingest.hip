// bc_ingest.hip -- FASTQ ingest for the engine.
//
// Replaces the reference's reader thread (input::read_fastq + FastqLineReader, input.rs:24-149): same 4-line
// framing, same "Total sequences" accounting (with its quirks), same first-record sanity check
// (RawSequenceRead::check_fastq_format, parse.rs:377-427).  The reference pushes one packed String per read onto a
// mutex-guarded VecDeque; here the host only MOVES bytes and the device does the framing:
//
//   host     a reader team pread()s the file (page cache -> pinned chunk buffers, several threads; zlib for .gz),
//   PCIe     the raw text goes to the device as it is (hipMemcpyAsync on the ingest stream),
//   device   newline scan (count, prefix sum, positions), record table (where each record's sequence and quality
//            line start, how long they are), then a gather into the fixed-stride sequence / quality batch the match
//            kernel reads -- bc_engine_submit_device[_q] on the engine's stream.
//
// Chunks are arbitrary byte ranges of the file.  A record that straddles two chunks is finished in the second: the
// device keeps the file offset of the first unframed byte, and every device text buffer starts with a copy of the
// previous chunk's last kOverlap bytes (device to device), so the second chunk sees the whole record.
// Three slots rotate: while the device frames and counts chunk i, the team reads chunk i+1 and chunk i-1's batch may
// still be in the match kernel.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/barcode_count_hip.h"
#include "bc_plan.hpp"

using namespace bc;

namespace {

constexpr size_t kOverlap = 4u << 20;  // longest record tail that may be carried into the next chunk
constexpr int kSlots = 3;
constexpr uint32_t kScanBlock = 4096;  // text bytes per 256-thread block of the newline kernels (16 per thread)

// what the device reports per chunk (pinned host memory)
struct ChunkStats {
  unsigned long long n_lines;  // newlines in [start, len)
  unsigned long long n_rec;    // whole records among them
  unsigned long long end_pos;  // buffer offset just past the last whole record (= start when there is none)
  long long start;             // buffer offset of the first unframed byte; < 0: the overlap was too short
  unsigned int min_len, max_len;  // sequence-line lengths over the chunk's records
  unsigned int max_qlen;
  unsigned int qual_differs;   // some record's quality line is not as long as its sequence line
  unsigned int last_is_newline;
  unsigned int pad;
};

struct DevState {
  unsigned long long next_off;  // file offset of the first byte no record has been made of yet
};

__device__ __forceinline__ uint32_t newline_mask16(const uint4& v, uint32_t first_valid, uint32_t n_valid) {
  // bit i set: byte i of the 16 is '\n' and first_valid <= i < n_valid
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  uint32_t m = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t x = w[k] ^ 0x0A0A0A0Au;
    const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;  // 0x80 where the byte is '\n'
    m |= (((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u)) << (4 * k);
  }
  uint32_t keep = n_valid >= 16u ? 0xFFFFu : ((1u << n_valid) - 1u);
  keep &= ~((1u << (first_valid > 16u ? 16u : first_valid)) - 1u);
  return m & keep;
}

__global__ void ingest_begin_kernel(DevState* st, unsigned long long buf_file_off, unsigned long long len, ChunkStats* cs,
                                    const uint8_t* text) {
  const long long start = (long long)st->next_off - (long long)buf_file_off;
  cs->start = start;
  cs->n_lines = 0;
  cs->n_rec = 0;
  cs->end_pos = start < 0 ? 0ull : (unsigned long long)start;
  cs->min_len = 0xFFFFFFFFu;
  cs->max_len = 0;
  cs->max_qlen = 0;
  cs->qual_differs = 0;
  cs->last_is_newline = len ? (text[len - 1] == '\n') : 1u;
}

// newlines per block of kScanBlock bytes
__global__ __launch_bounds__(256) void ingest_count_kernel(const uint8_t* __restrict__ text, unsigned long long len,
                                                           const ChunkStats* __restrict__ cs, uint32_t* __restrict__ blk_cnt) {
  __shared__ uint32_t s_sum[4];
  const long long start = cs->start < 0 ? (long long)len : cs->start;
  const unsigned long long p = (unsigned long long)blockIdx.x * kScanBlock + threadIdx.x * 16u;
  uint32_t c = 0;
  if (p < len && p + 16 > (unsigned long long)start) {
    const uint4 v = *reinterpret_cast<const uint4*>(text + p);
    const uint32_t first = (unsigned long long)start > p ? (uint32_t)((unsigned long long)start - p) : 0u;
    const uint32_t valid = len - p >= 16 ? 16u : (uint32_t)(len - p);
    c = __popc(newline_mask16(v, first, valid));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

// exclusive prefix sum of the block counts (one workgroup), totals into the stats
__global__ __launch_bounds__(1024) void ingest_scan_kernel(const uint32_t* __restrict__ blk_cnt, uint32_t n_blk,
                                                           uint32_t* __restrict__ blk_off, ChunkStats* cs, uint64_t line_cap) {
  __shared__ uint32_t s_part[1024];
  const uint32_t per = (n_blk + 1023u) / 1024u;
  const uint32_t a = threadIdx.x * per, b = min(n_blk, a + per);
  uint32_t sum = 0;
  for (uint32_t i = a; i < b; ++i) sum += blk_cnt[i];
  s_part[threadIdx.x] = sum;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
    const uint32_t v = threadIdx.x >= d ? s_part[threadIdx.x - d] : 0u;
    __syncthreads();
    s_part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = s_part[threadIdx.x] - sum;
  for (uint32_t i = a; i < b; ++i) {
    blk_off[i] = run;
    run += blk_cnt[i];
  }
  if (threadIdx.x == 1023) {
    unsigned long long lines = s_part[1023];
    if (lines > line_cap) lines = line_cap;  // (never with the caps used: one position slot per two text bytes)
    cs->n_lines = lines;
    cs->n_rec = lines / 4;
  }
}

// position of every newline, in order
__global__ __launch_bounds__(256) void ingest_positions_kernel(const uint8_t* __restrict__ text, unsigned long long len,
                                                               const ChunkStats* __restrict__ cs,
                                                               const uint32_t* __restrict__ blk_off, uint32_t* __restrict__ nl_pos,
                                                               uint64_t line_cap) {
  __shared__ uint32_t s_wave[4];
  const long long start = cs->start < 0 ? (long long)len : cs->start;
  const unsigned long long p = (unsigned long long)blockIdx.x * kScanBlock + threadIdx.x * 16u;
  uint32_t m = 0;
  if (p < len && p + 16 > (unsigned long long)start) {
    const uint4 v = *reinterpret_cast<const uint4*>(text + p);
    const uint32_t first = (unsigned long long)start > p ? (uint32_t)((unsigned long long)start - p) : 0u;
    const uint32_t valid = len - p >= 16 ? 16u : (uint32_t)(len - p);
    m = newline_mask16(v, first, valid);
  }
  const uint32_t c = __popc(m);
  // exclusive scan of c over the block: within the wave by shuffles, across the four waves through LDS
  uint32_t incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)incl, o);
    if ((threadIdx.x & 63) >= (uint32_t)o) incl += t;
  }
  if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
  __syncthreads();
  uint32_t base = blk_off[blockIdx.x];
  for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) base += s_wave[w];
  uint32_t rank = base + incl - c;
  while (m) {
    const uint32_t i = __ffs(m) - 1u;
    m &= m - 1u;
    if (rank < line_cap) nl_pos[rank] = (uint32_t)(p + i);
    ++rank;
  }
}

// one thread per record: where its sequence and quality lines are, and how long
__global__ __launch_bounds__(256) void ingest_records_kernel(const uint8_t* __restrict__ text, const uint32_t* __restrict__ nl_pos,
                                                             ChunkStats* cs, int strip_cr, uint32_t* __restrict__ seq_at,
                                                             uint32_t* __restrict__ qual_at, uint16_t* __restrict__ lens,
                                                             uint16_t* __restrict__ qlens) {
  const unsigned long long n_rec = cs->n_rec;
  const unsigned long long r = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t sl = 0xFFFFFFFFu, ql = 0, sl_max = 0;
  bool differs = false;
  if (r < n_rec) {
    const uint32_t e0 = nl_pos[4 * r], e1 = nl_pos[4 * r + 1], e2 = nl_pos[4 * r + 2], e3 = nl_pos[4 * r + 3];
    uint32_t s = e1 - (e0 + 1u), q = e3 - (e2 + 1u);
    // BufReader::lines() drops a "\r\n" ending (input.rs:44); the gz path's read_line keeps the '\r' (input.rs:66-68)
    if (strip_cr && s && text[e1 - 1] == '\r') --s;
    if (strip_cr && q && text[e3 - 1] == '\r') --q;
    seq_at[r] = e0 + 1u;
    qual_at[r] = e2 + 1u;
    lens[r] = (uint16_t)(s > 65535u ? 65535u : s);
    qlens[r] = (uint16_t)(q > 65535u ? 65535u : q);
    sl = sl_max = s;
    ql = q;
    differs = s != q;
    if (r == n_rec - 1) cs->end_pos = (unsigned long long)e3 + 1ull;
  }
  // wave-level reduction, then one atomic per wave
  uint32_t mn = sl, mx = sl_max, mq = ql, df = differs ? 1u : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = min(mn, (uint32_t)__shfl_xor((int)mn, o));
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    mq = max(mq, (uint32_t)__shfl_xor((int)mq, o));
    df |= (uint32_t)__shfl_xor((int)df, o);
  }
  if ((threadIdx.x & 63) == 0 && mn != 0xFFFFFFFFu) {
    atomicMin(&cs->min_len, mn);
    atomicMax(&cs->max_len, mx);
    atomicMax(&cs->max_qlen, mq);
    if (df) atomicOr(&cs->qual_differs, 1u);
  }
}

// records [first, first + n) -> fixed-stride batch: one wavefront per record, lanes 0-31 move the sequence line,
// lanes 32-63 the quality line, a dword (four bytes gathered from the unaligned text) per lane and step
__global__ __launch_bounds__(256) void ingest_gather_kernel(const uint8_t* __restrict__ text, const uint32_t* __restrict__ seq_at,
                                                            const uint32_t* __restrict__ qual_at, const uint16_t* __restrict__ lens,
                                                            const uint16_t* __restrict__ qlens, unsigned long long first,
                                                            unsigned long long n, uint32_t stride, uint8_t* __restrict__ out_seq,
                                                            uint8_t* __restrict__ out_qual) {
  const unsigned long long r = (unsigned long long)blockIdx.x * 4u + (threadIdx.x >> 6);
  if (r >= n) return;
  const uint32_t lane = threadIdx.x & 63u;
  const bool is_qual = lane >= 32u;
  const uint32_t j0 = lane & 31u;
  const uint32_t at = is_qual ? qual_at[first + r] : seq_at[first + r];
  uint32_t len = is_qual ? (uint32_t)qlens[first + r] : (uint32_t)lens[first + r];
  if (len > stride) len = stride;
  const uint8_t* src = text + at;
  uint32_t* dst = reinterpret_cast<uint32_t*>((is_qual ? out_qual : out_seq) + r * (unsigned long long)stride);
  const uint32_t pad = is_qual ? (uint32_t)'!' : (uint32_t)'N';
  for (uint32_t d = j0; d < stride / 4u; d += 32u) {
    uint32_t w = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      const uint32_t i = 4u * d + k;
      w |= (i < len ? (uint32_t)src[i] : pad) << (8u * k);
    }
    dst[d] = w;
  }
}

// the unfinished tail of the previous chunk in front of this chunk's bytes
__global__ void ingest_overlap_kernel(const uint8_t* __restrict__ prev_end, uint8_t* __restrict__ dst_end, uint32_t bytes) {
  // copies the `bytes` bytes that end at prev_end to the `bytes` bytes that end at dst_end
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < bytes) dst_end[-(long long)bytes + i] = prev_end[-(long long)bytes + i];
}

__global__ void ingest_advance_kernel(DevState* st, const ChunkStats* cs, unsigned long long buf_file_off) {
  if (cs->start >= 0) st->next_off = buf_file_off + cs->end_pos;
}

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

#define HIP_TRY(expr)                                                     \
  do {                                                                    \
    hipError_t _e = (expr);                                               \
    if (_e != hipSuccess) {                                               \
      set_error(std::string(#expr) + ": " + hipGetErrorString(_e));       \
      return BC_ERR_HIP;                                                  \
    }                                                                     \
  } while (0)

struct Slot {
  uint8_t* pin = nullptr;       // [kOverlap headroom unused on the host | chunk bytes]
  uint8_t* d_text = nullptr;    // [kOverlap | chunk bytes]
  uint32_t* d_blk_cnt = nullptr;
  uint32_t* d_blk_off = nullptr;
  uint32_t* d_nl_pos = nullptr;
  uint32_t* d_seq_at = nullptr;
  uint32_t* d_qual_at = nullptr;
  uint16_t* d_lens = nullptr;
  uint16_t* d_qlens = nullptr;
  uint8_t* d_out_seq = nullptr;
  uint8_t* d_out_qual = nullptr;
  ChunkStats* stats = nullptr;  // pinned host
  ChunkStats* d_stats = nullptr;
  hipEvent_t uploaded = nullptr;   // the pinned text may be overwritten
  hipEvent_t framed = nullptr;     // the stats have arrived on the host
  hipEvent_t gathered = nullptr;   // the batch arrays are complete (ingest stream)
  hipEvent_t consumed = nullptr;   // the match kernel has read the batch arrays (engine stream)
  size_t len = 0;                  // text bytes of the chunk held now
  size_t ov = 0;                   // bytes of overlap in front of them on the device
  unsigned long long file_off = 0; // file offset of the chunk's first byte
  bool eof = false;
};

struct Ingest {
  bc_engine* engine = nullptr;
  hipStream_t st = nullptr, engine_stream = nullptr;
  size_t chunk = 0, out_cap = 0;
  uint64_t line_cap = 0, rec_cap = 0;
  uint32_t n_blk_cap = 0;
  Slot slot[kSlots];
  DevState* d_state = nullptr;
  bool gz = false;
  uint32_t stride = 0, ragged_stride = 0;

  int alloc() {
    n_blk_cap = (uint32_t)((kOverlap + chunk + kScanBlock - 1) / kScanBlock);
    line_cap = (kOverlap + chunk) / 2;
    rec_cap = line_cap / 4;
    out_cap = 2 * (kOverlap + chunk);
    HIP_TRY(hipMalloc((void**)&d_state, sizeof(DevState)));
    HIP_TRY(hipMemset(d_state, 0, sizeof(DevState)));
    for (Slot& s : slot) {
      HIP_TRY(hipHostMalloc((void**)&s.pin, chunk + 16, hipHostMallocDefault));
      HIP_TRY(hipHostMalloc((void**)&s.stats, sizeof(ChunkStats), hipHostMallocDefault));
      HIP_TRY(hipMalloc((void**)&s.d_stats, sizeof(ChunkStats)));
      HIP_TRY(hipMalloc((void**)&s.d_text, kOverlap + chunk + 64));
      HIP_TRY(hipMalloc((void**)&s.d_blk_cnt, (size_t)n_blk_cap * 4));
      HIP_TRY(hipMalloc((void**)&s.d_blk_off, (size_t)n_blk_cap * 4));
      HIP_TRY(hipMalloc((void**)&s.d_nl_pos, line_cap * 4));
      HIP_TRY(hipMalloc((void**)&s.d_seq_at, rec_cap * 4));
      HIP_TRY(hipMalloc((void**)&s.d_qual_at, rec_cap * 4));
      HIP_TRY(hipMalloc((void**)&s.d_lens, rec_cap * 2));
      HIP_TRY(hipMalloc((void**)&s.d_qlens, rec_cap * 2));
      HIP_TRY(hipMalloc((void**)&s.d_out_seq, out_cap));
      HIP_TRY(hipMalloc((void**)&s.d_out_qual, out_cap));
      HIP_TRY(hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&s.framed, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&s.gathered, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming));
    }
    return BC_OK;
  }

  void release() {
    for (Slot& s : slot) {
      if (s.pin) (void)hipHostFree(s.pin);
      if (s.stats) (void)hipHostFree(s.stats);
      void* dev[] = {s.d_stats, s.d_text, s.d_blk_cnt, s.d_blk_off, s.d_nl_pos, s.d_seq_at, s.d_qual_at, s.d_lens, s.d_qlens,
                     s.d_out_seq, s.d_out_qual};
      for (void* p : dev)
        if (p) (void)hipFree(p);
      for (hipEvent_t ev : {s.uploaded, s.framed, s.gathered, s.consumed})
        if (ev) (void)hipEventDestroy(ev);
    }
    if (d_state) (void)hipFree(d_state);
    if (st) (void)hipStreamDestroy(st);
  }

  // chunk in slot b (host side filled) -> device, framed; the stats travel back asynchronously
  int frame(int b, const Slot* prev) {
    Slot& s = slot[b];
    HIP_TRY(hipStreamWaitEvent(st, s.consumed, 0));  // the batch arrays of this slot may still be read by a match kernel
    HIP_TRY(hipMemcpyAsync(s.d_text + kOverlap, s.pin, s.len, hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(s.uploaded, st));
    s.ov = 0;
    if (prev) {
      s.ov = std::min(kOverlap, prev->ov + prev->len);
      hipLaunchKernelGGL(ingest_overlap_kernel, dim3((uint32_t)((s.ov + 255) / 256)), dim3(256), 0, st,
                         prev->d_text + kOverlap + prev->len, s.d_text + kOverlap, (uint32_t)s.ov);
    }
    const uint8_t* text = s.d_text + kOverlap - s.ov;  // 16-byte aligned: kOverlap and ov are multiples of 16 ...
    // ... unless the previous chunk was shorter than the overlap (only the file's first chunks can be): align down
    const size_t mis = (size_t)((uintptr_t)text & 15u);
    text -= mis;
    const unsigned long long buf_off = s.file_off - s.ov - mis;  // may wrap below zero only together with start >= mis
    const unsigned long long len = s.ov + mis + s.len;
    const uint32_t n_blk = (uint32_t)((len + kScanBlock - 1) / kScanBlock);
    hipLaunchKernelGGL(ingest_begin_kernel, dim3(1), dim3(1), 0, st, d_state, buf_off, len, s.d_stats, text);
    hipLaunchKernelGGL(ingest_count_kernel, dim3(n_blk), dim3(256), 0, st, text, len, s.d_stats, s.d_blk_cnt);
    hipLaunchKernelGGL(ingest_scan_kernel, dim3(1), dim3(1024), 0, st, s.d_blk_cnt, n_blk, s.d_blk_off, s.d_stats, line_cap);
    hipLaunchKernelGGL(ingest_positions_kernel, dim3(n_blk), dim3(256), 0, st, text, len, s.d_stats, s.d_blk_off, s.d_nl_pos,
                       line_cap);
    // the record kernel is sized for the most records the text can hold (a record has at least four bytes)
    const unsigned long long rec_max = std::min<unsigned long long>(rec_cap, len / 4 + 1);
    hipLaunchKernelGGL(ingest_records_kernel, dim3((uint32_t)((rec_max + 255) / 256)), dim3(256), 0, st, text, s.d_nl_pos,
                       s.d_stats, gz ? 0 : 1, s.d_seq_at, s.d_qual_at, s.d_lens, s.d_qlens);
    hipLaunchKernelGGL(ingest_advance_kernel, dim3(1), dim3(1), 0, st, d_state, s.d_stats, buf_off);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(s.stats, s.d_stats, sizeof(ChunkStats), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(s.framed, st));
    s_text[b] = text;
    s_text_len[b] = len;
    return BC_OK;
  }
  const uint8_t* s_text[kSlots] = {nullptr, nullptr, nullptr};
  unsigned long long s_text_len[kSlots] = {0, 0, 0};  // bytes of the framed text behind s_text

  // the framed chunk of slot b -> batches for the engine
  int submit(int b, uint64_t* n_rec_out) {
    Slot& s = slot[b];
    HIP_TRY(hipEventSynchronize(s.framed));
    const ChunkStats cs = *s.stats;
    *n_rec_out = 0;
    if (cs.start < 0) {
      set_error("a FASTQ record is longer than 4 MiB");
      return BC_ERR_INVALID;
    }
    if (cs.n_rec == 0) return BC_OK;
    if (cs.n_lines >= line_cap) {
      set_error("FASTQ text with lines of under two bytes on average: not supported by the engine");
      return BC_ERR_UNSUPPORTED;
    }
    if (cs.max_len > 65535u || cs.max_qlen > 65535u) {
      set_error("a FASTQ line is longer than 65535 bytes (not supported by the engine)");
      return BC_ERR_UNSUPPORTED;
    }
    const uint32_t want = std::max<uint32_t>(4u, (cs.max_len + 3u) & ~3u);
    const bool uniform = cs.min_len == cs.max_len && !cs.qual_differs;
    // fixed-length chunks get exactly their stride (the kernel is specialised for the shape); ragged ones keep the
    // widest stride seen so far, so that a file of varying lengths settles on one kernel shape
    if (!uniform) ragged_stride = std::max(ragged_stride, want);
    stride = uniform ? want : ragged_stride;
    // the batch arrays hold out_cap bytes: a chunk whose stride is far above its average line goes in several parts
    const uint64_t per = std::max<uint64_t>(256, (out_cap / stride) & ~255ull);
    for (uint64_t first = 0; first < cs.n_rec; first += per) {
      const uint64_t n = std::min<uint64_t>(per, cs.n_rec - first);
      if (first) HIP_TRY(hipStreamWaitEvent(st, s.consumed, 0));  // the previous part's kernel still reads the arrays
      hipLaunchKernelGGL(ingest_gather_kernel, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, st, s_text[b], s.d_seq_at, s.d_qual_at,
                         s.d_lens, s.d_qlens, first, n, stride, s.d_out_seq, s.d_out_qual);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipEventRecord(s.gathered, st));
      HIP_TRY(hipStreamWaitEvent(engine_stream, s.gathered, 0));
      int rc;
      if (uniform)
        rc = bc_engine_submit_device(engine, s.d_out_seq, s.d_out_qual, nullptr, stride, cs.max_len, n);
      else if (!cs.qual_differs)
        rc = bc_engine_submit_device(engine, s.d_out_seq, s.d_out_qual, s.d_lens + first, stride, stride, n);
      else
        rc = bc_engine_submit_device_q(engine, s.d_out_seq, s.d_out_qual, s.d_lens + first, s.d_qlens + first, stride, n);
      if (rc != BC_OK) return rc;
      HIP_TRY(hipEventRecord(s.consumed, engine_stream));
    }
    *n_rec_out = cs.n_rec;
    return BC_OK;
  }
};

struct Source {
  bool gz = false;
  gzFile zf = nullptr;
  int fd = -1;
  unsigned long long pos = 0;  // next byte to read (plain files)
  unsigned long long size = 0; // plain files
  unsigned threads = 4;
  // nothing left after what fill() has returned so far
  bool at_end() {
    if (!gz) return pos >= size;
    const int c = gzgetc(zf);
    if (c < 0) return true;
    gzungetc(c, zf);
    return false;
  }
  // fills dst with up to cap bytes; returns the count (0 at end of file), -1 on error
  long fill(uint8_t* dst, size_t cap) {
    if (gz) {
      size_t got = 0;
      while (got < cap) {
        const int n = gzread(zf, dst + got, (unsigned)std::min<size_t>(cap - got, 1u << 30));
        if (n < 0) return -1;
        if (n == 0) break;
        got += (size_t)n;
      }
      return (long)got;
    }
    // (size = where this reader's share of the file ends: the file's end, or the shard's)
    if (pos >= size) return 0;
    cap = (size_t)std::min<unsigned long long>(cap, size - pos);
    // page cache -> pinned memory, one slice per thread
    const size_t slice = (((cap + threads - 1) / threads) + 4095) & ~(size_t)4095;  // (never 0: cap may be a few bytes)
    std::vector<long> got(threads, 0);
    std::vector<std::thread> team;
    auto work = [&](unsigned t) {
      const size_t a = std::min(cap, (size_t)t * slice), b = std::min(cap, a + slice);
      size_t done = 0;
      while (a + done < b) {
        const ssize_t n = pread(fd, dst + a + done, b - a - done, (off_t)(pos + a + done));
        if (n < 0) {
          got[t] = -1;
          return;
        }
        if (n == 0) break;
        done += (size_t)n;
      }
      got[t] = (long)done;
    };
    for (unsigned t = 1; t < threads; ++t) team.emplace_back(work, t);
    work(0);
    for (auto& th : team) th.join();
    size_t total = 0;
    for (unsigned t = 0; t < threads; ++t) {
      if (got[t] < 0) return -1;
      total += (size_t)got[t];
      if ((size_t)got[t] < std::min(cap, (size_t)(t + 1) * slice) - std::min(cap, (size_t)t * slice)) break;  // end of file inside this slice
    }
    pos += total;
    return (long)total;
  }
};

// First record of a plain FASTQ file that starts at or after byte `off`: the first line start p >= off whose line
// begins with '@' while the line two further down begins with '+' (a quality line may begin with '@', but then the line
// two further down is a sequence line, which never begins with '+').  `size` when there is none; -1 on a read error or
// when no record boundary is found within 16 MiB (no FASTQ record is that long: the framing kernels allow 4 MiB).
long long record_start_at_or_after(int fd, unsigned long long off, unsigned long long size) {
  if (off == 0) return 0;
  if (off >= size) return (long long)size;
  const unsigned long long from = off - 1;  // (the byte before tells whether `off` itself starts a line)
  std::vector<char> buf;
  const size_t step = 1u << 20, limit = 16u << 20;
  for (;;) {
    const size_t have = buf.size();
    if (from + have >= size || have >= limit) break;
    const size_t want = (size_t)std::min<unsigned long long>(step, size - (from + have));
    buf.resize(have + want);
    size_t got = 0;
    while (got < want) {
      const ssize_t n = pread(fd, buf.data() + have + got, want - got, (off_t)(from + have + got));
      if (n < 0) return -1;
      if (n == 0) break;
      got += (size_t)n;
    }
    buf.resize(have + got);
    const bool at_end = from + buf.size() >= size;
    // line starts inside the window (buffer offsets), from the first one at or after `off`
    size_t p = 0;
    if (buf[0] != '\n') {
      const char* nl = (const char*)memchr(buf.data(), '\n', buf.size());
      if (!nl) {
        if (at_end) return (long long)size;
        continue;  // one long line so far
      }
      p = (size_t)(nl - buf.data());
    }
    p += 1;  // first byte after a newline that sits at or after off - 1
    bool need_more = false;
    while (p < buf.size()) {
      const char* e1 = (const char*)memchr(buf.data() + p, '\n', buf.size() - p);
      const char* e2 = e1 ? (const char*)memchr(e1 + 1, '\n', buf.size() - (size_t)(e1 + 1 - buf.data())) : nullptr;
      if (!e1 || !e2 || (size_t)(e2 + 1 - buf.data()) >= buf.size()) {
        need_more = true;  // the line two further down is not in the window yet
        break;
      }
      if (buf[p] == '@' && e2[1] == '+') return (long long)(from + p);
      p = (size_t)(e1 + 1 - buf.data());
    }
    if (at_end) return (long long)size;  // fewer than three lines left: no whole record starts here
    if (!need_more && p >= buf.size()) continue;
    if (buf.size() >= limit) return -1;
  }
  return from + buf.size() >= size ? (long long)size : -1;
}

}  // namespace

static int fastq_count_impl(bc_engine* e, const char* fastq_path, uint32_t shard, uint32_t n_shards, uint64_t* total_reads,
                            bc_progress_fn progress, void* user);

// where bc_fastq_count_shard would start a shard that nominally begins at byte `offset` (host logic only, no GPU)
extern "C" int bc_fastq_record_start(const char* fastq_path, uint64_t offset, uint64_t* start) {
  *start = 0;
  const int fd = open(fastq_path ? fastq_path : "", O_RDONLY);
  if (fd < 0) {
    set_error(std::string("Failed to open file: ") + (fastq_path ? fastq_path : ""));
    return BC_ERR_INVALID;
  }
  const off_t end = lseek(fd, 0, SEEK_END);
  const long long at = record_start_at_or_after(fd, offset, end > 0 ? (unsigned long long)end : 0ull);
  close(fd);
  if (at < 0) {
    set_error("no FASTQ record boundary found (read error, or not 4-line FASTQ)");
    return BC_ERR_INVALID;
  }
  *start = (uint64_t)at;
  return BC_OK;
}

extern "C" int bc_fastq_count(bc_engine* e, const char* fastq_path, uint64_t* total_reads, bc_progress_fn progress,
                              void* user) {
  return fastq_count_impl(e, fastq_path, 0, 1, total_reads, progress, user);
}

extern "C" int bc_fastq_count_shard(bc_engine* e, const char* fastq_path, uint32_t shard, uint32_t n_shards,
                                    uint64_t* total_reads, bc_progress_fn progress, void* user) {
  if (n_shards == 0 || shard >= n_shards) {
    set_error("bc_fastq_count_shard: shard outside 0 .. n_shards-1");
    return BC_ERR_INVALID;
  }
  return fastq_count_impl(e, fastq_path, shard, n_shards, total_reads, progress, user);
}

static int fastq_count_impl(bc_engine* e, const char* fastq_path, uint32_t shard, uint32_t n_shards, uint64_t* total_reads,
                            bc_progress_fn progress, void* user) {
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
  if (!gz) {
    const off_t end = lseek(src.fd, 0, SEEK_END);
    src.size = end > 0 ? (unsigned long long)end : 0ull;
  }
  // One shard of several (one per GPU of a job): the records that START inside this shard's share of the bytes.  A gz
  // stream cannot be entered in the middle: its first shard takes all of it, the others have nothing to read.
  const bool last_shard = shard + 1 == n_shards;
  if (n_shards > 1) {
    if (gz) {
      if (shard != 0) {
        gzclose(src.zf);
        return BC_OK;
      }
    } else {
      const unsigned long long size = src.size;
      const long long a = record_start_at_or_after(src.fd, size / n_shards * shard, size);
      const long long b = last_shard ? (long long)size : record_start_at_or_after(src.fd, size / n_shards * (shard + 1), size);
      if (a < 0 || b < 0) {
        close(src.fd);
        set_error("no FASTQ record boundary found near a shard boundary of " + path + " (read error, or not 4-line FASTQ)");
        return BC_ERR_INVALID;
      }
      src.pos = (unsigned long long)a;
      src.size = (unsigned long long)std::max(a, b);
    }
  }
  src.threads = std::min(8u, std::max(1u, std::thread::hardware_concurrency() / 2));
  if (const char* ev = getenv("BC_INGEST_THREADS")) src.threads = (unsigned)std::min(64, std::max(1, atoi(ev)));

  // chunk size: a multiple of 16 (the device reads the text 16 bytes at a time), no larger than the file needs;
  // BC_INGEST_CHUNK is for tests, which want records to straddle chunks in small files
  size_t chunk = gz ? (32u << 20) : (size_t)std::min<unsigned long long>(128u << 20, ((src.size >> 20) + 1) << 20);
  if (const char* ev = getenv("BC_INGEST_CHUNK")) chunk = (size_t)std::max(4096L, atol(ev));
  chunk = (chunk + 15) & ~(size_t)15;
  // The pinned and device buffers of the last call are kept for the next one on the same device with the same
  // chunk size (pinning a few hundred MiB costs more than reading a small file); one call at a time per process.
  static std::mutex g_mu;
  static Ingest* g_cached = nullptr;
  static int g_device = -1;
  std::unique_lock<std::mutex> whole_call(g_mu);
  const int device = bc_engine_device(e);
  int rc = BC_OK;
  if (hipSetDevice(device) != hipSuccess) {
    set_error("bc_fastq_count: no HIP device");
    return BC_ERR_HIP;
  }
  if (g_cached && (g_device != device || g_cached->chunk != chunk)) {
    g_cached->release();
    delete g_cached;
    g_cached = nullptr;
  }
  if (!g_cached) {
    g_cached = new Ingest();
    g_cached->chunk = chunk;
    g_device = device;
    if (hipStreamCreateWithFlags(&g_cached->st, hipStreamNonBlocking) != hipSuccess) {
      set_error("bc_fastq_count: could not create a stream");
      rc = BC_ERR_HIP;
    }
    if (rc == BC_OK) rc = g_cached->alloc();
    if (rc != BC_OK) {
      g_cached->release();
      delete g_cached;
      g_cached = nullptr;
      if (gz) gzclose(src.zf); else close(src.fd);
      return rc;
    }
  }
  Ingest& in = *g_cached;
  in.engine = e;
  in.gz = gz;
  in.engine_stream = (hipStream_t)bc_engine_hip_stream(e);
  in.stride = in.ragged_stride = 0;
  for (Slot& sl : in.slot) {
    sl.len = sl.ov = 0;
    sl.file_off = 0;
    sl.eof = false;
  }
  auto finish = [&](int code) {
    (void)hipStreamSynchronize(in.st);
    (void)bc_engine_sync(e);  // the match kernels read the batch arrays, which the next call reuses
    if (gz)
      gzclose(src.zf);
    else
      close(src.fd);
    return code;
  };
  if (hipMemsetAsync(in.d_state, 0, sizeof(DevState), in.st) != hipSuccess) {
    set_error("bc_fastq_count: hipMemsetAsync failed");
    return finish(BC_ERR_HIP);
  }

  // the reader team runs ahead of the device by the slots that are free: a producer thread fills, this thread frames
  std::mutex mu;
  std::condition_variable cv;
  int filled_upto = 0;   // chunks [0, filled_upto) are in their slots
  int released_upto = kSlots;  // the producer may fill chunks [.., released_upto)
  bool read_error = false, stop = false;
  std::thread producer([&] {
    unsigned long long off = 0;
    for (int i = 0;; ++i) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || i < released_upto; });
        if (stop) return;
      }
      Slot& s = in.slot[i % kSlots];
      (void)hipEventSynchronize(s.uploaded);  // the slot's previous text has left for the device
      const long n = src.fill(s.pin, in.chunk);
      const bool last = n <= 0 || (size_t)n < in.chunk || src.at_end();
      {
        std::lock_guard<std::mutex> lk(mu);
        if (n < 0) read_error = true;
        s.len = n > 0 ? (size_t)n : 0;
        s.file_off = off;
        s.eof = last;
        filled_upto = i + 1;
      }
      cv.notify_all();
      if (last) return;  // end of file (or error)
      off += (unsigned long long)n;
    }
  });

  uint64_t total = 0, lines_after_last_record = 0;
  bool last_byte_newline = true, any_bytes = false, appended_newline = false, gz_last_char_dropped = false;
  bool test = shard == 0;  // (the file's first record is the first shard's)
  int pending = -1;  // chunk framed but not yet submitted
  int last_counted_slot = -1;  // slot of the last chunk whose records were counted (its text is still on the device)
  for (int i = 0;; ++i) {
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return filled_upto > i; });
      if (read_error) {
        set_error("read error in " + path);
        rc = BC_ERR_INVALID;
      }
    }
    if (rc != BC_OK) break;
    Slot& s = in.slot[i % kSlots];
    const bool eof = s.eof;
    if (s.len) {
      any_bytes = true;
      if (test) {  // first record only (input.rs:139-142, parse.rs:377-394): lines 1 and 2 of the file
        const char* t = (const char*)s.pin;
        const char* e1 = (const char*)memchr(t, '\n', s.len);
        const char* e2 = e1 ? (const char*)memchr(e1 + 1, '\n', s.len - (size_t)(e1 + 1 - t)) : nullptr;
        // (a file of fewer than four whole lines never posts a record, so the reference never looks at it)
        const char* e3 = e2 ? (const char*)memchr(e2 + 1, '\n', s.len - (size_t)(e2 + 1 - t)) : nullptr;
        const bool whole = e3 && (memchr(e3 + 1, '\n', s.len - (size_t)(e3 + 1 - t)) || (eof && !gz && (size_t)(e3 + 1 - t) < s.len));
        if (whole) {
          size_t n1 = (size_t)(e1 - t), n2 = (size_t)(e2 - (e1 + 1));
          if (!gz && n1 && t[n1 - 1] == '\r') --n1;
          if (!gz && n2 && e2[-1] == '\r') --n2;
          if (looks_like_sequence(t, n1)) {
            set_error("The first line within the FASTQ contains DNA sequences.  Check the FASTQ format");
            rc = BC_ERR_INVALID;
          } else if (!looks_like_sequence(e1 + 1, n2)) {
            set_error("The second line within the FASTQ file is not a sequence. Check the FASTQ format");
            rc = BC_ERR_INVALID;
          }
        }
        test = false;
        if (rc != BC_OK) break;
      }
      if (eof && s.pin[s.len - 1] != '\n') {
        last_byte_newline = false;
        if (!gz) {  // lines() hands the last line over without its newline (input.rs:44): framing-wise it has one
          s.pin[s.len++] = '\n';
          appended_newline = true;
        } else {
          // read_line hands the unterminated last line over as it is, and post() pops the record's last character
          // whatever it is (input.rs:137): when that line is a record's fourth, the record is scored with a quality
          // line one character short.  Turning the character into the missing newline is exactly that; when the line is
          // a record's first, second or third, no record comes of it and only the line count matters.
          s.pin[s.len - 1] = '\n';
          gz_last_char_dropped = true;
        }
      }
      rc = in.frame(i % kSlots, i > 0 ? &in.slot[(i - 1) % kSlots] : nullptr);
      if (rc != BC_OK) break;
    }
    // the chunk before this one: its stats are in (or about to be); count it while this one is being framed
    if (pending >= 0) {
      uint64_t n_rec = 0;
      rc = in.submit(pending % kSlots, &n_rec);
      if (rc != BC_OK) break;
      total += n_rec;
      if (progress && n_rec) progress(total - total % 10000, user);  // the reference prints every 10,000 reads (input.rs:54-57)
      {
        const ChunkStats& cs = *in.slot[pending % kSlots].stats;
        lines_after_last_record = cs.n_lines - 4 * cs.n_rec;
        last_counted_slot = pending % kSlots;
      }
      pending = -1;
      std::lock_guard<std::mutex> lk(mu);
      released_upto = i + kSlots - 1;  // slot (i - 1) % kSlots may be refilled once its upload event has fired
      cv.notify_all();
    }
    if (s.len) pending = i;
    if (eof) {
      if (pending >= 0) {
        uint64_t n_rec = 0;
        rc = in.submit(pending % kSlots, &n_rec);
        if (rc != BC_OK) break;
        total += n_rec;
        if (progress && n_rec) progress(total - total % 10000, user);  // the reference prints every 10,000 reads (input.rs:54-57)
        const ChunkStats& cs = *in.slot[pending % kSlots].stats;
        lines_after_last_record = cs.n_lines - 4 * cs.n_rec;
        last_counted_slot = pending % kSlots;
      }
      break;
    }
  }
  {
    std::lock_guard<std::mutex> lk(mu);
    stop = true;
  }
  cv.notify_all();
  producer.join();
  if (rc != BC_OK) return finish(rc);

  // what is left after the last whole record: fewer than four complete lines (+ possibly a last line without '\n')
  (void)any_bytes;
  {
    // lines the reference's reader would have been handed after the last whole record
    size_t seen = (size_t)lines_after_last_record;
    // (gz without a final newline: the unterminated last line was given its newline above, so it is among the lines
    // the device counted)
    (void)gz_last_char_dropped;
    (void)appended_newline;
    if (!last_shard && !(gz && shard == 0) && seen != 0) {
      // a shard that does not end the file ends on a record boundary; lines left over mean the file's lines do not
      // come in fours from where this shard started -- the reference, framing from the file's first line, would read
      // it differently from here on
      set_error("the lines of " + path + " do not come in records of four: run it on one GPU");
      return finish(BC_ERR_INVALID);
    }
    if (gz && seen == 3 && last_counted_slot >= 0) {
      // The gz loop hands read() one more, empty line at the end of the stream (input.rs:69-73).  After three lines of
      // a record that makes "line 4": the reference posts the partial record -- header, sequence, '+' line and an EMPTY
      // quality line (post() pops the last character, unpack() fills what lines there are: parse.rs:236-267) -- and its
      // workers score it like any other read (an empty quality line passes the quality filter: nothing is zipped,
      // parse.rs:340-345).  Here: the record's second line, fetched back from the device text, goes through the engine
      // as one read with a quality line of length 0.
      const Slot& ls = in.slot[last_counted_slot];
      const unsigned long long from = ls.stats->end_pos, upto = in.s_text_len[last_counted_slot];
      std::vector<char> tail((size_t)(upto > from ? upto - from : 0));
      if (!tail.empty() && hipMemcpy(tail.data(), in.s_text[last_counted_slot] + from, tail.size(), hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("bc_fastq_count: reading the stream's last lines back failed");
        return finish(BC_ERR_HIP);
      }
      const char* l1 = (const char*)memchr(tail.data(), '\n', tail.size());
      const char* l2 = l1 ? (const char*)memchr(l1 + 1, '\n', tail.size() - (size_t)(l1 + 1 - tail.data())) : nullptr;
      if (l1 && l2) {
        const size_t n = (size_t)(l2 - (l1 + 1));
        if (n > 65535) {
          set_error("a FASTQ line is longer than 65535 bytes (not supported by the engine)");
          return finish(BC_ERR_UNSUPPORTED);
        }
        const uint32_t one_stride = std::max<uint32_t>(16u, (uint32_t)((n + 15) & ~(size_t)15));
        uint8_t* d_one = nullptr;
        if (hipMalloc((void**)&d_one, (size_t)one_stride * 2 + 32) != hipSuccess) {
          (void)hipGetLastError();
          set_error("bc_fastq_count: out of device memory");
          return finish(BC_ERR_NOMEM);
        }
        std::vector<uint8_t> host((size_t)one_stride * 2 + 32, (uint8_t)'\n');
        memcpy(host.data(), l1 + 1, n);
        const uint16_t len16 = (uint16_t)n, qlen16 = 0;
        memcpy(host.data() + 2 * (size_t)one_stride, &len16, 2);
        memcpy(host.data() + 2 * (size_t)one_stride + 16, &qlen16, 2);
        int rc1 = hipMemcpy(d_one, host.data(), host.size(), hipMemcpyHostToDevice) == hipSuccess ? BC_OK : BC_ERR_HIP;
        if (rc1 == BC_OK)
          rc1 = bc_engine_submit_device_q(e, d_one, d_one + one_stride, d_one + 2 * (size_t)one_stride, d_one + 2 * (size_t)one_stride + 16,
                                          one_stride, 1);
        if (rc1 == BC_OK) rc1 = bc_engine_sync(e);
        (void)hipFree(d_one);
        if (rc1 != BC_OK) return finish(rc1);
      }
    }
    if (seen > 0 && seen < 4) total += 1;  // a trailing partial record is counted when its first line is seen (input.rs:128-130)
    if (gz) {
      // the gz loop calls read("") once more at EOF (input.rs:69-73): when that lands on "line 1" the total grows
      // by one (README.md:159 vs 176)
      if (seen % 4 == 0) total += 1;
    }
  }
  if (total_reads) *total_reads = total;
  return finish(BC_OK);
}
