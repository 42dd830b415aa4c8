// tools/read_rate.hip -- what this box's HBM gives a pure streaming READ (the match kernel's stream is read-only;
// a device copy, the usual yardstick, is half reads and half writes).  Three forms: plain 16-byte loads, the same
// non-temporal, and global_load_lds (the tile DMA the match kernel uses).
//   hipcc --offload-arch=gfx950 -O3 -o tools/read_rate tools/read_rate.hip && tools/read_rate [GiB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
template <bool kNT>
__global__ __launch_bounds__(256) void read_kernel(const v4u* __restrict__ src, uint64_t n16, uint32_t* out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; i + 3 * step < n16; i += 4 * step) {
    v4u v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = kNT ? __builtin_nontemporal_load(&src[i + k * step]) : src[i + k * step];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
  }
  for (; i < n16; i += step) {
    const v4u v = src[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;  // (keeps the loads alive)
}

// wave-tile DMA into LDS, 6400 bytes per wave per round as the match kernel does (no use of the data)
__global__ __launch_bounds__(256) void dma_kernel(const uint8_t* __restrict__ src, uint64_t bytes, uint32_t* out) {
  extern __shared__ uint4 smem[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint8_t* tile = reinterpret_cast<uint8_t*>(smem) + wave * 2u * 6656u;
  const uint64_t n_tiles = bytes / 6400u;
  const uint64_t n_waves = (uint64_t)gridDim.x * 4u;
  uint32_t acc = 0;
  uint32_t buf = 0;
  for (uint64_t t = (uint64_t)blockIdx.x * 4u + wave; t < n_tiles; t += n_waves, buf ^= 1u) {
    const uint8_t* s = src + t * 6400u;
    uint8_t* dst = tile + buf * 6656u;
    for (uint32_t off0 = 0; off0 < 6400u; off0 += 1024u) {
      const uint32_t off = off0 + lane * 16u;
      if (off < 6400u)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + off),
                                         (__attribute__((address_space(3))) void*)(dst + off0), 16, 0, 2);
    }
    // the previous tile (other buffer) has landed once at most this tile's 7 loads are outstanding
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    acc ^= reinterpret_cast<const uint32_t*>(tile + (buf ^ 1u) * 6656u)[lane];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 20.0;
  const uint64_t bytes = ((uint64_t)(gib * (1ull << 30)) / 6400u) * 6400u;
  uint8_t* d = nullptr;
  uint32_t* out = nullptr;
  CK(hipMalloc((void**)&d, bytes));
  CK(hipMalloc((void**)&out, 64));
  CK(hipMemset(d, 0x41, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  auto time_it = [&](const char* name, auto&& launch) {
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(e0, 0));
      launch();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
      sum += ms;
    }
    printf("%-28s %.3f ms best, %.3f ms mean: %.0f GB/s best\n", name, best, sum / 5, bytes / (best * 1e-3) / 1e9);
  };
  for (int per_cu : {4, 8}) {
    printf("grid = %d workgroups per CU x %d CUs\n", per_cu, cus);
    time_it("read, 16-byte loads", [&]() { hipLaunchKernelGGL(read_kernel<false>, dim3(cus * per_cu), dim3(256), 0, 0, (const v4u*)d, bytes / 16, out); });
    time_it("read, non-temporal", [&]() { hipLaunchKernelGGL(read_kernel<true>, dim3(cus * per_cu), dim3(256), 0, 0, (const v4u*)d, bytes / 16, out); });
  }
  for (int per_cu : {2, 3}) {
    printf("tile DMA (global_load_lds), %d workgroups per CU\n", per_cu);
    time_it("dma", [&]() { hipLaunchKernelGGL(dma_kernel, dim3(cus * per_cu), dim3(256), 4 * 2 * 6656, 0, d, bytes, out); });
  }
  uint8_t* d2 = nullptr;
  if (hipMalloc((void**)&d2, bytes / 2) == hipSuccess) {
    time_it("hipMemcpy D2D (half the size)", [&]() { CK(hipMemcpyAsync(d2, d, bytes / 2, hipMemcpyDeviceToDevice, 0)); });
    printf("  (a copy moves twice its size: x2 = read+write rate)\n");
  }
  return 0;
}
