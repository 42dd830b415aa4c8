// Micro-benchmark behind DESIGN.md's counting section: throughput of no-return u32 atomic adds at random
// addresses as a function of the table range they fall in (is a locality-restoring partition pass worth it?)
//   hipcc --offload-arch=gfx950 -O3 -o tools/atomic_rate tools/atomic_rate.hip && tools/atomic_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  return x;
}

// every lane adds to `per` random counters inside [0, range)
__global__ void scatter_add(uint32_t* table, uint64_t range, uint32_t per, uint64_t seed) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (uint32_t i = 0; i < per; ++i) {
    const uint64_t h = mix(seed + t * per + i);
    atomicAdd(&table[h % range], 1u);
  }
}

// the same number of adds over the whole table, but region by region (64 regions): every line is touched at
// most about once, so this isolates what address ORDER buys (DRAM pages, TLB) from what cache REUSE buys
__global__ void scatter_add_by_region(uint32_t* table, uint64_t range, uint32_t per, uint64_t seed, uint32_t regions) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t total = (uint64_t)gridDim.x * blockDim.x * per;
  const uint64_t rsize = range / regions;
  for (uint32_t i = 0; i < per; ++i) {
    const uint64_t op = (uint64_t)i * ((uint64_t)gridDim.x * blockDim.x) + t;  // consecutive threads, consecutive ops
    const uint64_t region = op / (total / regions);
    atomicAdd(&table[region * rsize + mix(seed + op) % rsize], 1u);
  }
}

// the same with an explicit cache policy on the atomic instruction
template <int kPolicy>
__global__ void scatter_add_policy(uint32_t* table, uint64_t range, uint32_t per, uint64_t seed) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (uint32_t i = 0; i < per; ++i) {
    const uint64_t h = mix(seed + t * per + i);
    uint32_t* p = &table[h % range];
    const uint32_t one = 1u;
    if (kPolicy == 0) asm volatile("global_atomic_add %0, %1, off" ::"v"(p), "v"(one) : "memory");
    if (kPolicy == 1) asm volatile("global_atomic_add %0, %1, off nt" ::"v"(p), "v"(one) : "memory");
    if (kPolicy == 2) asm volatile("global_atomic_add %0, %1, off sc1" ::"v"(p), "v"(one) : "memory");
    if (kPolicy == 3) asm volatile("global_atomic_add %0, %1, off sc1 nt" ::"v"(p), "v"(one) : "memory");
  }
}

// scope / return-value variants: kScope 0 = agent (what atomicAdd is), 1 = workgroup (the atomic may be done in the
// XCD's own L2: only correct when no other XCD touches the address during the kernel); kRet: the old value is used
template <int kScope, int kRet, int kOr>
__global__ void scatter_scope(uint32_t* table, uint64_t range, uint32_t per, uint64_t seed, uint32_t* sink) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < per; ++i) {
    const uint64_t h = mix(seed + t * per + i);
    uint32_t* p = &table[h % range];
    uint32_t old;
    if (kOr) {
      old = kScope ? __hip_atomic_fetch_or(p, 1u << (h >> 59), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                   : __hip_atomic_fetch_or(p, 1u << (h >> 59), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      old = kScope ? __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                   : __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (kRet) acc += old;
  }
  if (kRet && acc == 0xFFFFFFFFu) sink[0] = acc;
}

// same address stream, plain loads (what a gather of that locality costs)
__global__ void scatter_load(const uint32_t* table, uint64_t range, uint32_t per, uint64_t seed, uint32_t* sink) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < per; ++i) acc += table[mix(seed + t * per + i) % range];
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

// a reader that streams `bytes` once (plain or non-temporal loads), for the "atomics under streaming load" test
template <int kNT>
__global__ void stream_read(const uint4* __restrict__ src, uint64_t n16, uint32_t* sink) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; i < n16; i += step) {
    uint4 v;
    if (kNT) {
      typedef uint32_t u4 __attribute__((ext_vector_type(4)));
      const u4 t = __builtin_nontemporal_load(reinterpret_cast<const u4*>(src) + i);
      v = make_uint4(t.x, t.y, t.z, t.w);
    } else {
      v = src[i];
    }
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const uint64_t max_entries = 4ull << 30;  // 16 GiB of u32
  uint32_t* table = nullptr;
  if (hipMalloc((void**)&table, max_entries * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(table, 0, max_entries * 4);
  uint32_t* sink = nullptr;
  hipMalloc((void**)&sink, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const uint32_t per = 16, threads = 256, blocks = 256 * 64;  // 67 M operations per launch
  const double n = (double)per * threads * blocks;
  const uint64_t ranges[] = {1ull << 18, 1ull << 20, 1ull << 22, 1ull << 24, 1ull << 26, 1ull << 27, 1ull << 28, 1ull << 30, 4ull << 30};
  for (uint64_t r : ranges) {
    float ms_a = 0, ms_l = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(scatter_add, dim3(blocks), dim3(threads), 0, 0, table, r, per, 12345ull + rep);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms_a, e0, e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(scatter_load, dim3(blocks), dim3(threads), 0, 0, table, r, per, 12345ull + rep, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms_l, e0, e1);
    }
    float ms_p[4] = {0, 0, 0, 0};
    for (int pol = 0; pol < 4; ++pol) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (pol == 0) hipLaunchKernelGGL(scatter_add_policy<0>, dim3(blocks), dim3(threads), 0, 0, table, r, per, 777ull + rep);
        if (pol == 1) hipLaunchKernelGGL(scatter_add_policy<1>, dim3(blocks), dim3(threads), 0, 0, table, r, per, 777ull + rep);
        if (pol == 2) hipLaunchKernelGGL(scatter_add_policy<2>, dim3(blocks), dim3(threads), 0, 0, table, r, per, 777ull + rep);
        if (pol == 3) hipLaunchKernelGGL(scatter_add_policy<3>, dim3(blocks), dim3(threads), 0, 0, table, r, per, 777ull + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms_p[pol], e0, e1);
      }
    }
    float ms_s[6] = {0, 0, 0, 0, 0, 0};
    for (int v = 0; v < 6; ++v) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (v == 0) hipLaunchKernelGGL((scatter_scope<0, 0, 0>), dim3(blocks), dim3(threads), 0, 0, table, r, per, 555ull + rep, sink);
        if (v == 1) hipLaunchKernelGGL((scatter_scope<1, 0, 0>), dim3(blocks), dim3(threads), 0, 0, table, r, per, 555ull + rep, sink);
        if (v == 2) hipLaunchKernelGGL((scatter_scope<0, 1, 0>), dim3(blocks), dim3(threads), 0, 0, table, r, per, 555ull + rep, sink);
        if (v == 3) hipLaunchKernelGGL((scatter_scope<1, 1, 0>), dim3(blocks), dim3(threads), 0, 0, table, r, per, 555ull + rep, sink);
        if (v == 4) hipLaunchKernelGGL((scatter_scope<0, 1, 1>), dim3(blocks), dim3(threads), 0, 0, table, r, per, 555ull + rep, sink);
        if (v == 5) hipLaunchKernelGGL((scatter_scope<1, 1, 1>), dim3(blocks), dim3(threads), 0, 0, table, r, per, 555ull + rep, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms_s[v], e0, e1);
      }
    }
    printf("range %8.0f MiB  atomics %7.2f G/s   loads %7.2f G/s   asm: plain %6.2f  nt %6.2f  sc1 %6.2f  sc1+nt %6.2f G/s\n",
           r * 4.0 / (1 << 20), n / ms_a / 1e6, n / ms_l / 1e6, n / ms_p[0] / 1e6, n / ms_p[1] / 1e6, n / ms_p[2] / 1e6,
           n / ms_p[3] / 1e6);
    printf("                     add agent %6.2f  add wg-scope %6.2f | returning: add agent %6.2f  add wg %6.2f  or agent %6.2f  or wg %6.2f G/s\n",
           n / ms_s[0] / 1e6, n / ms_s[1] / 1e6, n / ms_s[2] / 1e6, n / ms_s[3] / 1e6, n / ms_s[4] / 1e6, n / ms_s[5] / 1e6);
    fflush(stdout);
  }
  for (uint32_t regions : {1u, 64u, 1024u, 16384u}) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(scatter_add_by_region, dim3(blocks), dim3(threads), 0, 0, table, max_entries, per, 4242ull + rep, regions);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("67 M adds over 16 GiB, ordered by region (%5u regions of %7.1f MiB): %6.2f G/s\n", regions,
           max_entries * 4.0 / regions / (1 << 20), n / ms / 1e6);
  }
  // atomics confined to 256 MB while another stream reads 12 GB: does the Infinity Cache keep serving them?
  {
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    const uint64_t n16 = (12ull << 30) / 16;
    const uint4* src = reinterpret_cast<const uint4*>(table + (1ull << 30));  // 4 GB into the table, clear of the atomics' range
    for (int nt = 0; nt < 2; ++nt) {
      for (uint64_t r : {1ull << 26, 4ull << 30}) {
        hipDeviceSynchronize();
        hipEventRecord(e0, sb);
        for (int k = 0; k < 4; ++k) {
          if (nt) hipLaunchKernelGGL(stream_read<1>, dim3(256 * 4), dim3(256), 0, sa, src, n16, sink);
          else hipLaunchKernelGGL(stream_read<0>, dim3(256 * 4), dim3(256), 0, sa, src, n16, sink);
        }
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(scatter_add, dim3(blocks), dim3(threads), 0, sb, table, r, per, 999ull + k);
        hipEventRecord(e1, sb);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipDeviceSynchronize();
        printf("under a concurrent %s 12 GB x4 read: atomics over %6.0f MiB at %.2f G/s\n", nt ? "non-temporal" : "plain",
               r * 4.0 / (1 << 20), 4 * n / ms / 1e6);
      }
    }
  }
  return 0;
}
