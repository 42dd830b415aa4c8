// bc_comm.hip -- the multi-GPU end of a job behind the C ABI (include/barcode_count_hip.h: bc_comm_*,
// bc_engine_reduce_all, bc_engine_finish_all): one process per GPU, reads sharded by the caller, ONE exchange at the end
// (SURVEY.md 8(e)).  The exchange logic is bc_exchange.hpp; this file gives it
//   * its two transports for device buffers: RCCL over xGMI (ncclSend / ncclRecv groups on the engine's stream; librccl
//     is loaded when the first communicator is made, so single-GPU users never touch it), and the message-file
//     transport of bc_comm.hpp behind pinned-free host staging (several ranks on one GPU, machines without peer access);
//   * its device memory space (HipOps: pack / sum / widen / scatter / key partition kernels on the engine's stream).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <memory>
#include <string>
#include <vector>

#include "../../include/barcode_count_hip.h"
#include "bc_exchange.hpp"
#include "bc_plan.hpp"

using namespace bc;

extern "C" {  // bc_engine.hip (not part of the documented ABI)
void* bc_internal_table_unfolded(bc_engine* e, const void** bits);
int bc_internal_table_now_plain(bc_engine* e);
int bc_internal_table_pack_u8(const void* d_table_u32, const void* d_bits, uint64_t n, void* d_out_u8, void* d_ovf_idx_u64,
                              void* d_ovf_val_u32, uint64_t ovf_capacity, uint64_t* n_ovf, int device_id, void* hip_stream);
}

#define HIPC(expr)                                                                   \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) {                                                          \
      set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                  \
      if (_e == hipErrorOutOfMemory) (void)hipGetLastError();                        \
      return _e == hipErrorOutOfMemory ? BC_ERR_NOMEM : BC_ERR_HIP;                  \
    }                                                                                \
  } while (0)

namespace {

// ---- kernels of the exchange's device memory space ------------------------------------------------------------
// out[i] = sum over the n_rows byte slices rows[r * len + i]
__global__ void comm_sum_u8_kernel(const uint8_t* __restrict__ rows, uint32_t n_rows, uint64_t len, uint32_t* __restrict__ out) {
  const uint64_t n4 = len >> 2;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  const bool aligned = (len & 3u) == 0u && ((uintptr_t)rows & 3u) == 0u;
  if (aligned) {
    for (; i < n4; i += step) {
      uint32_t a = 0, b = 0, c = 0, d = 0;
      for (uint32_t r = 0; r < n_rows; ++r) {
        const uint32_t v = reinterpret_cast<const uint32_t*>(rows + (uint64_t)r * len)[i];
        a += v & 255u;
        b += (v >> 8) & 255u;
        c += (v >> 16) & 255u;
        d += v >> 24;
      }
      out[4 * i] = a;
      out[4 * i + 1] = b;
      out[4 * i + 2] = c;
      out[4 * i + 3] = d;
    }
  } else {
    for (; i < len; i += step) {
      uint32_t a = 0;
      for (uint32_t r = 0; r < n_rows; ++r) a += rows[(uint64_t)r * len + i];
      out[i] = a;
    }
  }
}

// out[i] = how many of the n_rows bit maps (`words` u32 each, back to back) have bit i, i < len: one word column per thread
__global__ void comm_sum_bits_kernel(const uint32_t* __restrict__ rows, uint32_t n_rows, uint64_t words, uint64_t len,
                                     uint32_t* __restrict__ out) {
  uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (; j < words; j += step) {
    uint32_t c[32];
#pragma unroll
    for (int b = 0; b < 32; ++b) c[b] = 0;
    for (uint32_t r = 0; r < n_rows; ++r) {
      const uint32_t w = rows[(uint64_t)r * words + j];
#pragma unroll
      for (int b = 0; b < 32; ++b) c[b] += (w >> b) & 1u;
    }
    const uint64_t first = j * 32;
    if (first + 32 <= len) {
      uint4* o = reinterpret_cast<uint4*>(out + first);
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = make_uint4(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]);
    } else {
#pragma unroll
      for (int b = 0; b < 32; ++b)
        if (first + (uint64_t)b < len) out[first + b] = c[b];
    }
  }
}

// counts below 4 as two bit planes (planes[w]: bit 0, planes[words + w]: bit 1, of entries 32 w .. 32 w + 31); a count
// of 4 or more leaves both bits clear and goes to the list (one thread per plane word)
__global__ void comm_pack_planes2_kernel(const uint32_t* __restrict__ counts, uint64_t len, uint64_t words, uint32_t* __restrict__ planes,
                                         unsigned long long* cursor, uint64_t capacity, unsigned long long* __restrict__ out_idx,
                                         uint32_t* __restrict__ out_val) {
  uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (; j < words; j += step) {
    uint32_t p0 = 0, p1 = 0;
    for (uint32_t b = 0; b < 32u; ++b) {
      const uint64_t i = j * 32 + b;
      if (i >= len) break;
      const uint32_t c = counts[i];
      if (c < 4u) {
        p0 |= (c & 1u) << b;
        p1 |= (c >> 1) << b;
      } else {
        const unsigned long long p = atomicAdd(cursor, 1ull);
        if (p < capacity) {
          out_idx[p] = i;
          out_val[p] = c;
        }
      }
    }
    planes[j] = p0;
    planes[words + j] = p1;
  }
}
__global__ void comm_unpack_planes2_kernel(const uint32_t* __restrict__ planes, uint64_t words, uint64_t len, uint32_t* __restrict__ dst) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (; i < len; i += step) {
    const uint32_t sh = (uint32_t)(i & 31u);
    dst[i] = ((planes[i >> 5] >> sh) & 1u) + 2u * ((planes[words + (i >> 5)] >> sh) & 1u);
  }
}

// the non-zero entries of a table as (index, value) pairs, in no particular order (one cursor add per wavefront)
__global__ void comm_table_nonzero_kernel(const uint32_t* __restrict__ table, uint64_t n, unsigned long long* cursor, uint64_t capacity,
                                          unsigned long long* __restrict__ out_idx, uint32_t* __restrict__ out_val) {
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < n; base += step) {
    const uint64_t i = base + threadIdx.x;
    const uint32_t v = i < n ? table[i] : 0u;
    const unsigned long long m = __ballot(v != 0u);
    if (m == 0ull) continue;
    unsigned long long first = 0;
    const unsigned leader = (unsigned)(__ffsll((long long)m) - 1);
    if (__lane_id() == leader) first = atomicAdd(cursor, (unsigned long long)__popcll(m));
    first = ((unsigned long long)(unsigned)__shfl((int)(first >> 32), (int)leader) << 32) |
            (unsigned long long)(unsigned)__shfl((int)first, (int)leader);
    if (v) {
      const unsigned long long p = first + (unsigned long long)__popcll(m & ((1ull << __lane_id()) - 1ull));
      if (p < capacity) {
        out_idx[p] = i;
        out_val[p] = v;
      }
    }
  }
}

__global__ void comm_widen_u8_kernel(const uint8_t* __restrict__ src, uint64_t n, uint32_t* __restrict__ dst) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) dst[i] = src[i];
}

__global__ void comm_scatter_add_kernel(uint32_t* __restrict__ table, const unsigned long long* __restrict__ idx,
                                        const uint32_t* __restrict__ val, uint64_t m) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (; i < m; i += step) atomicAdd(&table[idx[i]], val[i]);
}

__device__ __forceinline__ int dev_key_owner(unsigned long long key, int world) {
  unsigned long long x = key * 0x9E3779B97F4A7C15ull;
  x ^= x >> 32;
  return (int)((x >> 7) % (unsigned long long)world);
}
// keys per owner (one cursor add per wavefront and owner)
__global__ void comm_owner_count_kernel(const unsigned long long* __restrict__ keys, uint32_t words, uint64_t n, int world,
                                        int fixed_owner, unsigned long long* __restrict__ counts) {
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < n; base += step) {
    const uint64_t i = base + threadIdx.x;
    const int o = i < n ? (fixed_owner >= 0 ? fixed_owner : dev_key_owner(keys[i * words], world)) : -1;
    for (int r = 0; r < world; ++r) {
      const unsigned long long m = __ballot(o == r);
      if (m && __lane_id() == (unsigned)(__ffsll((long long)m) - 1)) atomicAdd(&counts[r], (unsigned long long)__popcll(m));
    }
  }
}
// ... and into their owner's range of the output (cursor[r] starts at the range's first slot)
__global__ void comm_owner_scatter_kernel(const unsigned long long* __restrict__ keys, uint32_t words,
                                          const uint32_t* __restrict__ vals, uint64_t n, int world, int fixed_owner,
                                          unsigned long long* __restrict__ cursor,
                                          unsigned long long* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < n; base += step) {
    const uint64_t i = base + threadIdx.x;
    const unsigned long long k = i < n ? keys[i * words] : 0ull;
    const int o = i < n ? (fixed_owner >= 0 ? fixed_owner : dev_key_owner(k, world)) : -1;
    for (int r = 0; r < world; ++r) {
      const unsigned long long m = __ballot(o == r);
      if (!m) continue;
      const unsigned leader = (unsigned)(__ffsll((long long)m) - 1);
      unsigned long long first = 0;
      if (__lane_id() == leader) first = atomicAdd(&cursor[r], (unsigned long long)__popcll(m));
      first = ((unsigned long long)(unsigned)__shfl((int)(first >> 32), (int)leader) << 32) |
              (unsigned long long)(unsigned)__shfl((int)first, (int)leader);
      if (o == r) {
        const unsigned long long p = first + (unsigned long long)__popcll(m & ((1ull << __lane_id()) - 1ull));
        keys_out[p * words] = k;
        for (uint32_t w = 1; w < words; ++w) keys_out[p * words + w] = keys[i * words + w];
        if (vals) vals_out[p] = vals[i];
      }
    }
  }
}

uint32_t grid_of(uint64_t n) { return (uint32_t)std::min<uint64_t>((n + 255) / 256 + 1, 256ull * 32); }

struct HipOps {
  int device;
  hipStream_t st;
  // the engine's own table may come with its first-occurrence bit map still apart (two-level counting): packed as
  // table + bit, no fold pass first
  const uint32_t* engine_table = nullptr;
  const void* engine_bits = nullptr;
  void* alloc(size_t bytes) {
    void* p = nullptr;
    const hipError_t rc = hipMalloc(&p, bytes ? bytes : 16);
    if (rc != hipSuccess) {
      set_error(std::string("exchange: hipMalloc of ") + std::to_string(bytes) + " bytes: " + hipGetErrorString(rc));
      (void)hipGetLastError();
      return nullptr;
    }
    return p;
  }
  void release(void* p) { (void)hipFree(p); }
  int sync() {
    HIPC(hipStreamSynchronize(st));
    return 0;
  }
  int pack_u8(const uint32_t* table, uint64_t n, uint8_t* out, std::vector<uint64_t>& ovf_idx, std::vector<uint32_t>& ovf_val) {
    ovf_idx.clear();
    ovf_val.clear();
    if (n == 0) return 0;
    uint64_t cap = std::max<uint64_t>(1024, n / 4096);
    for (;;) {
      unsigned long long* d_idx = nullptr;
      uint32_t* d_val = nullptr;
      HIPC(hipMalloc((void**)&d_idx, cap * 8));
      if (hipMalloc((void**)&d_val, cap * 4) != hipSuccess) {
        (void)hipFree(d_idx);
        set_error("exchange: out of device memory for the overflow list");
        (void)hipGetLastError();
        return BC_ERR_NOMEM;
      }
      uint64_t need = 0;
      int rc = bc_internal_table_pack_u8(table, table == engine_table ? engine_bits : nullptr, n, out, d_idx, d_val, cap, &need, device,
                                         st);  // (waits for the stream)
      if (rc == BC_OK && need <= cap && need) {
        ovf_idx.resize(need);
        ovf_val.resize(need);
        if (hipMemcpy(ovf_idx.data(), d_idx, need * 8, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(ovf_val.data(), d_val, need * 4, hipMemcpyDeviceToHost) != hipSuccess) {
          set_error("exchange: reading the overflow list back failed");
          rc = BC_ERR_HIP;
        }
      }
      (void)hipFree(d_idx);
      (void)hipFree(d_val);
      if (rc != BC_OK) return rc;
      if (need <= cap) return 0;
      cap = need;
    }
  }
  int sum_u8(const uint8_t* rows, uint32_t n_rows, uint64_t len, uint32_t* out) {
    if (len == 0) return 0;
    hipLaunchKernelGGL(comm_sum_u8_kernel, dim3(grid_of(len / 4 + 1)), dim3(256), 0, st, rows, n_rows, len, out);
    HIPC(hipGetLastError());
    return 0;
  }
  int sum_bits(const uint32_t* rows, uint32_t n_rows, uint64_t words, uint64_t len, uint32_t* out) {
    if (len == 0) return 0;
    hipLaunchKernelGGL(comm_sum_bits_kernel, dim3(grid_of(words)), dim3(256), 0, st, rows, n_rows, words, len, out);
    HIPC(hipGetLastError());
    return 0;
  }
  int table_nonzero(const uint32_t* table, uint64_t n, uint64_t cap, std::vector<uint64_t>& idx, std::vector<uint32_t>& val, bool& fits) {
    idx.clear();
    val.clear();
    fits = true;
    if (n == 0) return 0;
    unsigned long long *d_n = nullptr, *d_idx = nullptr;
    uint32_t* d_val = nullptr;
    int rc = 0;
    if (hipMalloc((void**)&d_n, 8) != hipSuccess || hipMalloc((void**)&d_idx, cap * 8) != hipSuccess ||
        hipMalloc((void**)&d_val, cap * 4) != hipSuccess) {
      (void)hipGetLastError();
      set_error("exchange: out of device memory for the table's non-zero entries");
      rc = BC_ERR_NOMEM;
    }
    unsigned long long need = 0;
    if (!rc) {
      if (hipMemsetAsync(d_n, 0, 8, st) != hipSuccess) rc = BC_ERR_HIP;
      hipLaunchKernelGGL(comm_table_nonzero_kernel, dim3(grid_of(n)), dim3(256), 0, st, table, n, d_n, cap, d_idx, d_val);
      if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&need, d_n, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipStreamSynchronize(st) != hipSuccess)
        rc = BC_ERR_HIP;
    }
    if (!rc) {
      fits = need <= cap;
      if (fits && need) {
        idx.resize(need);
        val.resize(need);
        if (hipMemcpy(idx.data(), d_idx, need * 8, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(val.data(), d_val, need * 4, hipMemcpyDeviceToHost) != hipSuccess)
          rc = BC_ERR_HIP;
      }
    }
    if (d_n) (void)hipFree(d_n);
    if (d_idx) (void)hipFree(d_idx);
    if (d_val) (void)hipFree(d_val);
    if (rc == BC_ERR_HIP) set_error("exchange: collecting the table's non-zero entries failed");
    return rc;
  }
  int pack_planes2(const uint32_t* counts, uint64_t len, uint32_t* planes, std::vector<uint64_t>& idx, std::vector<uint32_t>& val,
                   uint64_t cap, bool& fits) {
    idx.clear();
    val.clear();
    fits = true;
    if (len == 0) return 0;
    const uint64_t words = (len + 31) / 32;
    unsigned long long *d_n = nullptr, *d_idx = nullptr;
    uint32_t* d_val = nullptr;
    int rc = 0;
    if (hipMalloc((void**)&d_n, 8) != hipSuccess || hipMalloc((void**)&d_idx, cap * 8) != hipSuccess ||
        hipMalloc((void**)&d_val, cap * 4) != hipSuccess) {
      (void)hipGetLastError();
      set_error("exchange: out of device memory for the side list");
      rc = BC_ERR_NOMEM;
    }
    unsigned long long need = 0;
    if (!rc) {
      if (hipMemsetAsync(d_n, 0, 8, st) != hipSuccess) rc = BC_ERR_HIP;
      hipLaunchKernelGGL(comm_pack_planes2_kernel, dim3(grid_of(words)), dim3(256), 0, st, counts, len, words, planes, d_n, cap, d_idx,
                         d_val);
      if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&need, d_n, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipStreamSynchronize(st) != hipSuccess)
        rc = BC_ERR_HIP;
    }
    if (!rc) {
      fits = need <= cap;
      if (fits && need) {
        idx.resize(need);
        val.resize(need);
        if (hipMemcpy(idx.data(), d_idx, need * 8, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(val.data(), d_val, need * 4, hipMemcpyDeviceToHost) != hipSuccess)
          rc = BC_ERR_HIP;
      }
    }
    if (d_n) (void)hipFree(d_n);
    if (d_idx) (void)hipFree(d_idx);
    if (d_val) (void)hipFree(d_val);
    if (rc == BC_ERR_HIP) set_error("exchange: packing the summed counts into bit planes failed");
    return rc;
  }
  int unpack_planes2(const uint32_t* planes, uint64_t words, uint64_t len, uint32_t* dst) {
    if (len == 0) return 0;
    hipLaunchKernelGGL(comm_unpack_planes2_kernel, dim3(grid_of(len)), dim3(256), 0, st, planes, words, len, dst);
    HIPC(hipGetLastError());
    return 0;
  }
  int widen_u8(const uint8_t* src, uint64_t n, uint32_t* dst) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(comm_widen_u8_kernel, dim3(grid_of(n)), dim3(256), 0, st, src, n, dst);
    HIPC(hipGetLastError());
    return 0;
  }
  int scatter_add(uint32_t* table, const uint64_t* idx, const uint32_t* val, uint64_t m) {
    if (m == 0) return 0;
    unsigned long long* d_idx = nullptr;
    uint32_t* d_val = nullptr;
    HIPC(hipMalloc((void**)&d_idx, m * 8));
    if (hipMalloc((void**)&d_val, m * 4) != hipSuccess) {
      (void)hipFree(d_idx);
      (void)hipGetLastError();
      set_error("exchange: out of device memory for the overflow list");
      return BC_ERR_NOMEM;
    }
    int rc = 0;
    if (hipMemcpyAsync(d_idx, idx, m * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(d_val, val, m * 4, hipMemcpyHostToDevice, st) != hipSuccess)
      rc = BC_ERR_HIP;
    if (!rc) {
      hipLaunchKernelGGL(comm_scatter_add_kernel, dim3(grid_of(m)), dim3(256), 0, st, table, d_idx, d_val, m);
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = BC_ERR_HIP;
    }
    (void)hipFree(d_idx);
    (void)hipFree(d_val);
    if (rc) set_error("exchange: adding the overflow list failed");
    return rc;
  }
  int partition_keys(const uint64_t* keys, uint32_t words, const uint32_t* vals, uint64_t n, int world, int fixed_owner,
                     uint64_t* keys_out, uint32_t* vals_out, uint64_t* counts) {
    for (int r = 0; r < world; ++r) counts[r] = 0;
    if (n == 0) return 0;
    unsigned long long* d_cnt = nullptr;
    HIPC(hipMalloc((void**)&d_cnt, (size_t)world * 16));
    int rc = 0;
    std::vector<unsigned long long> h((size_t)world, 0);
    if (hipMemsetAsync(d_cnt, 0, (size_t)world * 16, st) != hipSuccess) rc = BC_ERR_HIP;
    if (!rc) {
      hipLaunchKernelGGL(comm_owner_count_kernel, dim3(grid_of(n)), dim3(256), 0, st, (const unsigned long long*)keys, words, n, world,
                         fixed_owner, d_cnt);
      if (hipMemcpyAsync(h.data(), d_cnt, (size_t)world * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipStreamSynchronize(st) != hipSuccess)
        rc = BC_ERR_HIP;
    }
    if (!rc) {
      std::vector<unsigned long long> start((size_t)world, 0);
      for (int r = 1; r < world; ++r) start[(size_t)r] = start[(size_t)r - 1] + h[(size_t)r - 1];
      if (hipMemcpyAsync(d_cnt + world, start.data(), (size_t)world * 8, hipMemcpyHostToDevice, st) != hipSuccess) rc = BC_ERR_HIP;
      if (!rc) {
        hipLaunchKernelGGL(comm_owner_scatter_kernel, dim3(grid_of(n)), dim3(256), 0, st, (const unsigned long long*)keys, words, vals,
                           n, world, fixed_owner, d_cnt + world, (unsigned long long*)keys_out, vals_out);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = BC_ERR_HIP;
      }
    }
    (void)hipFree(d_cnt);
    if (rc) {
      set_error("exchange: partitioning the keys by owner failed");
      return rc;
    }
    for (int r = 0; r < world; ++r) counts[r] = h[(size_t)r];
    return 0;
  }
};

// ---- RCCL, loaded on first use ----------------------------------------------------------------------------------
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  std::string why;
};
Rccl* rccl() {
  static Rccl* r = []() {
    Rccl* x = new Rccl();
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      x->lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (x->lib) break;
    }
    if (!x->lib) {
      x->why = std::string("RCCL (librccl.so) could not be loaded: ") + (dlerror() ? dlerror() : "not found");
      return x;
    }
#define BC_SYM(field, sym)                                                     \
  x->field = reinterpret_cast<decltype(x->field)>(dlsym(x->lib, sym));           \
  if (!x->field && x->why.empty()) x->why = std::string("librccl.so has no ") + sym
    BC_SYM(get_unique_id, "ncclGetUniqueId");
    BC_SYM(comm_init_rank, "ncclCommInitRank");
    BC_SYM(comm_destroy, "ncclCommDestroy");
    BC_SYM(send, "ncclSend");
    BC_SYM(recv, "ncclRecv");
    BC_SYM(group_start, "ncclGroupStart");
    BC_SYM(group_end, "ncclGroupEnd");
    BC_SYM(error_string, "ncclGetErrorString");
#undef BC_SYM
    return x;
  }();
  return r;
}
#define NCCLC(expr)                                                                           \
  do {                                                                                        \
    ncclResult_t _r = (expr);                                                                 \
    if (_r != ncclSuccess) {                                                                  \
      set_error(std::string(#expr) + ": " + (rccl()->error_string ? rccl()->error_string(_r) : "RCCL error")); \
      return BC_ERR_HIP;                                                                      \
    }                                                                                         \
  } while (0)

struct RcclTransport : Transport {
  ncclComm_t comm = nullptr;
  hipStream_t st = nullptr;  // set per exchange: the engine's stream
  int device = 0;
  ~RcclTransport() override {
    if (comm && rccl()->comm_destroy) (void)rccl()->comm_destroy(comm);
  }
  int all_to_all_v(const void* send, const uint64_t* sb, void* recv, const uint64_t* rb) override {
    Rccl* R = rccl();
    const uint8_t* sp = (const uint8_t*)send;
    uint8_t* rp = (uint8_t*)recv;
    uint64_t so = 0, ro = 0;
    bool any = false;
    for (int r = 0; r < world; ++r) any = any || (r != rank && (sb[r] || rb[r]));
    if (any) NCCLC(R->group_start());
    for (int r = 0; r < world; ++r) {
      if (r == rank) {
        if (sb[r] != rb[r]) {
          set_error("exchange: a rank's slice for itself differs in length from what it expects");
          return BC_ERR_STATE;
        }
        if (sb[r]) HIPC(hipMemcpyAsync(rp + ro, sp + so, sb[r], hipMemcpyDeviceToDevice, st));
      } else {
        if (sb[r]) NCCLC(R->send(sp + so, (size_t)sb[r], ncclUint8, r, comm, st));
        if (rb[r]) NCCLC(R->recv(rp + ro, (size_t)rb[r], ncclUint8, r, comm, st));
      }
      so += sb[r];
      ro += rb[r];
    }
    if (any) NCCLC(R->group_end());
    HIPC(hipStreamSynchronize(st));
    return 0;
  }
  int host_all_to_all_v(const void* send, const uint64_t* sb, void* recv, const uint64_t* rb) override {
    uint64_t ts = 0, tr = 0;
    for (int r = 0; r < world; ++r) {
      ts += sb[r];
      tr += rb[r];
    }
    uint8_t *ds = nullptr, *dr = nullptr;
    HIPC(hipMalloc((void**)&ds, ts + 16));
    if (hipMalloc((void**)&dr, tr + 16) != hipSuccess) {
      (void)hipFree(ds);
      (void)hipGetLastError();
      set_error("exchange: out of device memory for a control message");
      return BC_ERR_NOMEM;
    }
    int rc = 0;
    if (ts && hipMemcpyAsync(ds, send, ts, hipMemcpyHostToDevice, st) != hipSuccess) rc = BC_ERR_HIP;
    if (!rc) rc = all_to_all_v(ds, sb, dr, rb);
    if (!rc && tr && hipMemcpy(recv, dr, tr, hipMemcpyDeviceToHost) != hipSuccess) rc = BC_ERR_HIP;
    (void)hipFree(ds);
    (void)hipFree(dr);
    if (rc == BC_ERR_HIP) set_error("exchange: staging a control message through the device failed");
    return rc;
  }
};

// device buffers over the message-file transport: staged through host memory
struct StagedTransport : Transport {
  HostDirTransport inner;
  hipStream_t st = nullptr;
  StagedTransport(const std::string& dir, int r, int w) : inner(dir, r, w) {
    rank = r;
    world = w;
  }
  int all_to_all_v(const void* send, const uint64_t* sb, void* recv, const uint64_t* rb) override {
    uint64_t ts = 0, tr = 0;
    for (int r = 0; r < world; ++r) {
      ts += sb[r];
      tr += rb[r];
    }
    std::vector<uint8_t> hs, hr;
    try {
      hs.resize(ts + 1);
      hr.resize(tr + 1);
    } catch (const std::bad_alloc&) {
      set_error("exchange: out of host memory for the staged transport");
      return BC_ERR_NOMEM;
    }
    HIPC(hipStreamSynchronize(st));
    if (ts) HIPC(hipMemcpy(hs.data(), send, ts, hipMemcpyDeviceToHost));
    const int rc = inner.host_all_to_all_v(hs.data(), sb, hr.data(), rb);
    if (rc) return rc;
    if (tr) HIPC(hipMemcpy(recv, hr.data(), tr, hipMemcpyHostToDevice));
    return 0;
  }
  int host_all_to_all_v(const void* send, const uint64_t* sb, void* recv, const uint64_t* rb) override {
    return inner.host_all_to_all_v(send, sb, recv, rb);
  }
};

}  // namespace

struct bc_comm {
  std::unique_ptr<Transport> t;
  RcclTransport* rccl_t = nullptr;     // (views of t, by kind)
  StagedTransport* staged_t = nullptr;
  int device = -1;  // RCCL: the device the communicator was made on
};

extern "C" {

int bc_comm_unique_id(void* id128) {
  Rccl* R = rccl();
  if (!R->why.empty()) {
    set_error(R->why);
    return BC_ERR_UNSUPPORTED;
  }
  ncclUniqueId id;
  NCCLC(R->get_unique_id(&id));
  static_assert(sizeof(id) == BC_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(id128, &id, sizeof(id));
  return BC_OK;
}

bc_comm* bc_comm_create(const void* id128, int rank, int world, int device_id) {
  if (!id128 || world < 1 || rank < 0 || rank >= world) {
    set_error("bc_comm_create: bad rank / world / id");
    return nullptr;
  }
  Rccl* R = rccl();
  if (!R->why.empty()) {
    set_error(R->why);
    return nullptr;
  }
  if (hipSetDevice(device_id) != hipSuccess) {
    set_error("bc_comm_create: no HIP device " + std::to_string(device_id));
    return nullptr;
  }
  auto t = std::make_unique<RcclTransport>();
  t->rank = rank;
  t->world = world;
  t->device = device_id;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  const ncclResult_t rc = R->comm_init_rank(&t->comm, world, id, rank);
  if (rc != ncclSuccess) {
    set_error(std::string("ncclCommInitRank: ") + R->error_string(rc));
    t->comm = nullptr;
    return nullptr;
  }
  bc_comm* c = new bc_comm();
  c->rccl_t = t.get();
  c->device = device_id;
  c->t = std::move(t);
  return c;
}

bc_comm* bc_comm_create_host(const char* dir, int rank, int world) {
  if (!dir || !*dir || world < 1 || rank < 0 || rank >= world) {
    set_error("bc_comm_create_host: bad directory / rank / world");
    return nullptr;
  }
  if (access(dir, W_OK) != 0) {
    set_error(std::string("bc_comm_create_host: cannot write to ") + dir);
    return nullptr;
  }
  auto t = std::make_unique<StagedTransport>(dir, rank, world);
  bc_comm* c = new bc_comm();
  c->staged_t = t.get();
  c->t = std::move(t);
  return c;
}

void bc_comm_destroy(bc_comm* c) { delete c; }
int bc_comm_rank(const bc_comm* c) { return c ? c->t->rank : 0; }
int bc_comm_world(const bc_comm* c) { return c ? c->t->world : 1; }

static int bind_stream(bc_engine* e, bc_comm* c, HipOps& ops) {
  ops.device = bc_engine_device(e);
  ops.st = (hipStream_t)bc_engine_hip_stream(e);
  if (c->rccl_t) {
    if (c->device != ops.device) {
      set_error("the communicator was made on device " + std::to_string(c->device) + ", the engine runs on device " +
                std::to_string(ops.device));
      return BC_ERR_INVALID;
    }
    c->rccl_t->st = ops.st;
  }
  if (c->staged_t) c->staged_t->st = ops.st;
  HIPC(hipSetDevice(ops.device));
  return BC_OK;
}

int bc_comm_barrier(bc_comm* c) {
  if (!c) return BC_OK;
  if (c->rccl_t) HIPC(hipSetDevice(c->device));  // (before any exchange the communicator works on that device's NULL stream)
  const int rc = c->t->barrier();
  return rc > 0 ? BC_ERR_HIP : rc;
}

int bc_comm_sum_u64(bc_comm* c, uint64_t* vals, int n, int root) {
  if (!c || c->t->world <= 1) return BC_OK;
  if (c->rccl_t) HIPC(hipSetDevice(c->device));
  const int rc = c->t->reduce_sum_u64(vals, n, root);
  return rc > 0 ? BC_ERR_HIP : rc;
}

int bc_engine_reduce_all(bc_engine* e, bc_comm* c, int root, uint64_t counters[BC_NCOUNTERS]) {
  uint64_t local[BC_NCOUNTERS];
  int rc = bc_engine_counters(e, local);  // (syncs the engine)
  if (rc) return rc;
  if (!c || c->t->world <= 1) {
    if (counters) memcpy(counters, local, sizeof(local));
    return BC_OK;
  }
  Transport& t = *c->t;
  if (root < 0 || root >= t.world) {
    set_error("bc_engine_reduce_all: root outside the communicator");
    return BC_ERR_INVALID;
  }
  HipOps ops;
  if ((rc = bind_stream(e, c, ops))) return rc;
  const bc_plan* p = bc_engine_plan(e);
  const int mode = bc_plan_mode(p);
  const bool sparse = mode == 2, random = bc_plan_has_random(p) != 0;
  auto status = [](int r) { return r > 0 ? BC_ERR_HIP : r; };

  if (random || sparse) {
    // keys (with their counts, for raw-key plans without a random barcode) to their owners: hashed over the ranks for
    // a dense table -- duplicates across ranks then collapse on the owner --, all to the root for a key map, whose
    // result the root has to hold whole
    uint64_t n = 0;
    const bool with_counts = sparse && !random;
    if (with_counts)
      rc = bc_engine_export_counts(e, nullptr, nullptr, 0, &n);
    else
      rc = bc_engine_key_count(e, &n);
    if (rc) return rc;
    const uint32_t kw = bc_engine_key_words(e);  // u64 words per key: 1, or a plan with wide keys
    unsigned long long* d_keys = (unsigned long long*)ops.alloc((size_t)(n * kw + 2) * 8);
    uint32_t* d_cnts = with_counts ? (uint32_t*)ops.alloc((size_t)(n + 4) * 4) : nullptr;
    uint64_t* got_k = nullptr;
    uint32_t* got_v = nullptr;
    uint64_t n_in = 0, n_new = 0;
    if (!d_keys || (with_counts && !d_cnts)) rc = BC_ERR_NOMEM;
    if (!rc && n) rc = with_counts ? bc_engine_export_counts(e, d_keys, d_cnts, n, &n) : bc_engine_export_keys(e, d_keys, n, &n);
    if (!rc)
      rc = status(exchange_keys(t, ops, (const uint64_t*)d_keys, kw, d_cnts, n, sparse ? root : -1, &got_k,
                                with_counts ? &got_v : nullptr, &n_in));
    if (!rc) rc = bc_engine_clear_keys(e);
    if (!rc && n_in) rc = with_counts ? bc_engine_import_counts(e, got_k, got_v, n_in) : bc_engine_import_keys(e, got_k, n_in, &n_new);
    if (d_keys) ops.release(d_keys);
    if (d_cnts) ops.release(d_cnts);
    if (got_k) ops.release(got_k);
    if (got_v) ops.release(got_v);
    if (rc) return rc;
    if (random) {
      // every read that passed all tests here is either the one survivor of its key on the key's owner, or a duplicate
      // (parse.rs:65-69 applied to the whole job): summed over the ranks, matched = distinct keys overall
      local[BC_DUPLICATES] += local[BC_MATCHED] - n_new;
      local[BC_MATCHED] = n_new;
    }
    if (random && !sparse && (rc = bc_engine_materialize_table(e))) return rc;  // the owned keys' per-tuple distinct counts
  }
  if (!sparse) {
    // (two-level counting: the bit map travels inside the packed bytes -- table + bit --, not through a fold pass)
    uint32_t* table = (uint32_t*)bc_internal_table_unfolded(e, &ops.engine_bits);
    ops.engine_table = table;
    int form = 0;
    if ((rc = status(reduce_tables(t, ops, table, (const uint32_t*)ops.engine_bits, bc_engine_table_entries(e), root, &form))))
      return rc;
    if (t.rank == root && getenv("BC_COMM_VERBOSE"))
      fprintf(stderr, "[barcode-count] table exchange over %d ranks: %s\n", t.world,
              form == 0 ? "byte-packed slices" : (form == 1 ? "bit-map slices + the tables' non-zero entries, sums as bytes"
                                                            : "bit-map slices + the tables' non-zero entries, sums as two bit planes"));
    if ((rc = bc_internal_table_now_plain(e))) return rc;  // root: the table is the job's sum; others: unspecified anyway
  }
  if ((rc = status(t.reduce_sum_u64(local, BC_NCOUNTERS, root)))) return rc;
  if (counters) {
    if (t.rank == root)
      memcpy(counters, local, sizeof(local));
    else
      memset(counters, 0, sizeof(local));
  }
  return BC_OK;
}

int bc_engine_finish_all(bc_engine* e, bc_comm* c, int root, uint64_t counters[BC_NCOUNTERS], uint64_t* n_rows) {
  if (n_rows) *n_rows = 0;
  int rc = bc_engine_reduce_all(e, c, root, counters);
  if (rc) return rc;
  if (!c || c->t->world <= 1 || c->t->rank == root) return bc_engine_finish(e, n_rows);
  return BC_OK;
}

}  // extern "C"
