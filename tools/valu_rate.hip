// perf-debug microbenchmark: issue rate of the integer VALU ops the lane code uses, per SIMD,
// at 1..8 waves per SIMD.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ void k(uint32_t* out, uint32_t seed, int iters) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a * 7u + 1u, d = b + 3u;
  uint32_t e = a + 11u, f = b + 13u, g = c + 17u, h = d + 19u;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (OP == 0) { a &= b ^ 0x55555555u; c &= d ^ 0x33333333u; e &= f ^ 0x0f0f0f0fu; g &= h ^ 0x00ff00ffu; b ^= a; d ^= c; f ^= e; h ^= g; }
      if (OP == 1) { a = __builtin_amdgcn_alignbit(b, a, 3); c = __builtin_amdgcn_alignbit(d, c, 5); e = __builtin_amdgcn_alignbit(f, e, 7); g = __builtin_amdgcn_alignbit(h, g, 9);
                     b = __builtin_amdgcn_alignbit(a, b, 11); d = __builtin_amdgcn_alignbit(c, d, 13); f = __builtin_amdgcn_alignbit(e, f, 15); h = __builtin_amdgcn_alignbit(g, h, 17); }
      if (OP == 2) { a = __builtin_amdgcn_udot4(b, 0x08040201u, a, false); c = __builtin_amdgcn_udot4(d, 0x08040201u, c, false); e = __builtin_amdgcn_udot4(f, 0x08040201u, e, false); g = __builtin_amdgcn_udot4(h, 0x08040201u, g, false);
                     b = __builtin_amdgcn_udot4(a, 0x80402010u, b, false); d = __builtin_amdgcn_udot4(c, 0x80402010u, d, false); f = __builtin_amdgcn_udot4(e, 0x80402010u, f, false); h = __builtin_amdgcn_udot4(g, 0x80402010u, h, false); }
      if (OP == 3) { a = __builtin_amdgcn_perm(b, a, c); c = __builtin_amdgcn_perm(d, c, e); e = __builtin_amdgcn_perm(f, e, g); g = __builtin_amdgcn_perm(h, g, a);
                     b = __builtin_amdgcn_perm(a, b, d); d = __builtin_amdgcn_perm(c, d, f); f = __builtin_amdgcn_perm(e, f, h); h = __builtin_amdgcn_perm(g, h, b); }
      if (OP == 4) { a = __builtin_amdgcn_sad_u8(b, c, a); c = __builtin_amdgcn_sad_u8(d, e, c); e = __builtin_amdgcn_sad_u8(f, g, e); g = __builtin_amdgcn_sad_u8(h, a, g);
                     b = __builtin_amdgcn_sad_u8(a, d, b); d = __builtin_amdgcn_sad_u8(c, f, d); f = __builtin_amdgcn_sad_u8(e, h, f); h = __builtin_amdgcn_sad_u8(g, b, h); }
      if (OP == 5) { a = (a & b) | (~a & c); c = (c & d) | (~c & e); e = (e & f) | (~e & g); g = (g & h) | (~g & a); b = (b ^ a ^ d); d = (d ^ c ^ f); f = (f ^ e ^ h); h = (h ^ g ^ b); }
      if (OP == 6) { a += __builtin_popcount(b); c += __builtin_popcount(d); e += __builtin_popcount(f); g += __builtin_popcount(h); b += __builtin_popcount(a); d += __builtin_popcount(c); f += __builtin_popcount(e); h += __builtin_popcount(g); }
      if (OP == 7) { a = (a > b) ? c : a; c = (c > d) ? e : c; e = (e > f) ? g : e; g = (g > h) ? a : g; b = (b > a) ? d : b; d = (d > c) ? f : d; f = (f > e) ? h : f; h = (h > g) ? b : h; }
      if (OP == 8) { float x = __uint_as_float(a), y = __uint_as_float(b), z = __uint_as_float(c), w = __uint_as_float(d);
                     x = __builtin_fmaf(x, y, z); z = __builtin_fmaf(z, w, x); y = __builtin_fmaf(y, x, w); w = __builtin_fmaf(w, z, y);
                     x = __builtin_fmaf(x, y, z); z = __builtin_fmaf(z, w, x); y = __builtin_fmaf(y, x, w); w = __builtin_fmaf(w, z, y);
                     a = __float_as_uint(x); b = __float_as_uint(y); c = __float_as_uint(z); d = __float_as_uint(w); }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}

template <int OP>
int run(const char* name, int ops_per_unroll) {
  uint32_t* out;
  CHECK(hipMalloc(&out, 256 * 8 * 256 * 4 * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 4000;
  for (int wps = 1; wps <= 8; wps *= 2) {
    // 256 CUs x 4 SIMDs x wps waves: blocks of 256 threads (4 waves -> one per SIMD), wps blocks per CU
    dim3 grid(256 * wps), block(256);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, 1u, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, 2u, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_simd = (double)wps * iters * 8 * ops_per_unroll;
    printf("%-10s waves/SIMD %d: %.3f ms, %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, wps, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
  }
  CHECK(hipFree(out));
  return 0;
}

int main() {
  run<0>("and/xor", 8); run<1>("alignbit", 8); run<2>("dot4", 8); run<3>("perm", 8); run<4>("sad_u8", 8);
  run<5>("bitop3", 8); run<6>("popc+add", 16); run<7>("cmp+cndmask", 16); run<8>("fma_f32", 8);
  return 0;
}
