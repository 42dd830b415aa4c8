// tools/skeleton.hip -- the memory-system floor of the match kernel's access pattern, with no matching at all: per
// wave-tile of 64 reads the tile DMA of the sequence (and quality) lines into LDS, and one counter atomic per
// "matched" read at a random place -- exactly what bc_kernel.h issues, software-pipelined the same way -- so that
// what is left of the real kernel's time above this is instruction issue and latency, not memory.
//   hipcc --offload-arch=gfx950 -O3 -o tools/skeleton tools/skeleton.hip && tools/skeleton
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  return x;
}
__device__ __forceinline__ void dma(uint8_t* lds, const uint8_t* src, uint32_t lane) {
  for (uint32_t off0 = 0; off0 < 6400u; off0 += 1024u) {
    const uint32_t off = off0 + lane * 16u;
    if (off < 6400u)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                       (__attribute__((address_space(3))) void*)(lds + off0), 16, 0, 2);
  }
}

// kQual: a second 6400-byte stream per tile.  kMode: 0 no atomics, 1 returning OR into a bit map (two-level counting's
// first level), 2 no-return add into the u32 table.  pct: lanes out of 64 that count (89 % of config 3's reads match).
template <int kQual, int kMode, int kWaves>
__global__ __launch_bounds__(kWaves * 64) void skeleton(const uint8_t* __restrict__ seq, const uint8_t* __restrict__ qual,
                                                        uint64_t n_tiles, uint32_t* table, uint64_t entries, uint32_t matched,
                                                        uint32_t* sink) {
  extern __shared__ uint4 smem[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint8_t* tile = reinterpret_cast<uint8_t*>(smem) + wave * (kQual ? 2u : 1u) * 6656u;
  uint8_t* qtile = tile + 6656u;
  const uint64_t n_waves = (uint64_t)gridDim.x * kWaves;
  uint64_t t = (uint64_t)blockIdx.x * kWaves + wave;
  uint32_t acc = 0, old = 0;
  if (t < n_tiles) {
    dma(tile, seq + t * 6400u, lane);
    if (kQual) dma(qtile, qual + t * 6400u, lane);
  }
  for (; t < n_tiles; t += n_waves) {
    const uint64_t tn = t + n_waves;
    // this tile's sequence lines have landed (its quality lines and the previous atomic may still be in flight)
    if (kQual) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    acc ^= reinterpret_cast<const uint32_t*>(tile)[lane * 25u];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (tn < n_tiles) dma(tile, seq + tn * 6400u, lane);  // "planes built": the next tile's sequence lines
    if (kQual) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // quality lines of this tile
      acc ^= reinterpret_cast<const uint32_t*>(qtile)[lane * 25u];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    acc += old;  // the previous tile's returned word, looked at one tile later
    if (kQual && tn < n_tiles) dma(qtile, qual + tn * 6400u, lane);
    if (kMode && lane < matched) {
      const uint64_t idx = mix(t * 64u + lane) % entries;
      if (kMode == 1) old = atomicOr(&table[idx >> 5], 1u << (idx & 31u));
      else atomicAdd(&table[idx], 1u);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc + old == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
  const uint64_t reads = argc > 1 ? strtoull(argv[1], nullptr, 0) : 100000000ull;
  const uint64_t n_tiles = reads / 64, bytes = n_tiles * 6400;
  const uint64_t entries = 4000000000ull;
  uint8_t *seq = nullptr, *qual = nullptr;
  uint32_t *table = nullptr, *sink = nullptr;
  CK(hipMalloc((void**)&seq, bytes));
  CK(hipMalloc((void**)&qual, bytes));
  CK(hipMalloc((void**)&table, entries * 4));
  CK(hipMalloc((void**)&sink, 64));
  CK(hipMemset(seq, 0x41, bytes));
  CK(hipMemset(qual, 0x49, bytes));
  CK(hipMemset(table, 0, entries * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  auto run = [&](const char* name, auto kern, int waves, int per_cu, int qual_on, uint64_t ent, uint32_t matched) {
    const size_t lds = (size_t)waves * (qual_on ? 2 : 1) * 6656;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
      CK(hipMemsetAsync(table, 0, 600000000, 0));
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(kern, dim3(cus * per_cu), dim3(waves * 64), lds, 0, seq, qual, n_tiles, table, ent, matched, sink);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r) best = ms < best ? ms : best;
    }
    const double gb = (double)bytes * (qual_on ? 2 : 1) / 1e9;
    printf("%-58s %d waves/WG x %d WG/CU: %.3f ms  %.2f G reads/s  stream %.0f GB/s\n", name, waves, per_cu, best,
           reads / (best * 1e-3) / 1e9, gb / (best * 1e-3));
    fflush(stdout);
  };
  printf("%llu reads of 100 bases; table / bit map of 4e9 tuples\n", (unsigned long long)reads);
  for (int per_cu : {2, 3, 4}) {
    run("seq+qual stream, no counting", skeleton<1, 0, 4>, 4, per_cu, 1, entries, 57);
    run("seq+qual stream, 57/64 returning OR into the bit map", skeleton<1, 1, 4>, 4, per_cu, 1, entries, 57);
    run("seq+qual stream, 57/64 adds into the 16 GB table", skeleton<1, 2, 4>, 4, per_cu, 1, entries, 57);
  }
  for (int per_cu : {3, 4, 6}) {
    run("seq stream only, no counting", skeleton<0, 0, 4>, 4, per_cu, 0, entries, 64);
    run("seq stream, 64/64 returning OR into the bit map (config 2)", skeleton<0, 1, 4>, 4, per_cu, 0, entries, 64);
    run("seq stream, 64/64 adds into the 16 GB table", skeleton<0, 2, 4>, 4, per_cu, 0, entries, 64);
    run("seq stream, 64/64 adds into a 400 KB table (config 5)", skeleton<0, 2, 4>, 4, per_cu, 0, 100000, 64);
  }
  return 0;
}
