// perf-debug microbenchmark: issue cost of single gfx950 VALU/SALU instructions (exact opcodes through inline
// asm, register operands, eight independent chains), per SIMD, at 1/2/3/4 waves per SIMD.
// Prints cycles per wave-instruction per SIMD at the measured shader clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define R8(X) X(a, b, c) X(b, c, d) X(c, d, e) X(d, e, f) X(e, f, g) X(f, g, h) X(g, h, a) X(h, a, b)
// one test = a macro I(dst, s0, s1) expanding to one asm statement
#define DEF(NAME, ASMSTR)                                                                          \
  __global__ void k_##NAME(uint32_t* out, uint32_t seed, int iters) {                              \
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a * 7u + 1u, d = b + 3u; \
    uint32_t e = a + 11u, f = b + 13u, g = c + 17u, h = d + 19u;                                   \
    for (int i = 0; i < iters; ++i) {                                                              \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                              \
        R8(I_##NAME)                                                                               \
      }                                                                                            \
    }                                                                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;                    \
  }

#define I_and(x, y, z) asm volatile("v_and_b32 %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
#define I_xor(x, y, z) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));
#define I_or(x, y, z) asm volatile("v_or_b32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));
#define I_xorlit(x, y, z) asm volatile("v_xor_b32 %0, 0x12345678, %1" : "=v"(x) : "v"(y));
#define I_andinl(x, y, z) asm volatile("v_and_b32 %0, 15, %1" : "=v"(x) : "v"(y));
#define I_not(x, y, z) asm volatile("v_not_b32 %0, %1" : "=v"(x) : "v"(y));
#define I_mov(x, y, z) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y));
#define I_add(x, y, z) asm volatile("v_add_u32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));
#define I_sub(x, y, z) asm volatile("v_sub_u32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));
#define I_shl(x, y, z) asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(x) : "v"(y));
#define I_shr(x, y, z) asm volatile("v_lshrrev_b32 %0, 5, %1" : "=v"(x) : "v"(y));
#define I_shrv(x, y, z) asm volatile("v_lshrrev_b32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));
#define I_alignbit(x, y, z) asm volatile("v_alignbit_b32 %0, %1, %2, 7" : "=v"(x) : "v"(y), "v"(z));
#define I_alignbyte(x, y, z) asm volatile("v_alignbyte_b32 %0, %1, %2, 1" : "=v"(x) : "v"(y), "v"(z));
#define I_bfe(x, y, z) asm volatile("v_bfe_u32 %0, %1, 3, 8" : "=v"(x) : "v"(y));
#define I_bfi(x, y, z) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
#define I_andor(x, y, z) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
#define I_or3(x, y, z) asm volatile("v_or3_b32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
#define I_lshlor(x, y, z) asm volatile("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(x) : "v"(y), "v"(z));
#define I_lshladd(x, y, z) asm volatile("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(x) : "v"(y), "v"(z));
#define I_add3(x, y, z) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
#define I_xad(x, y, z) asm volatile("v_xad_u32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
#define I_bitop3(x, y, z) asm volatile("v_bitop3_b32 %0, %1, %2, %0 bitop3:0x96" : "+v"(x) : "v"(y), "v"(z));
#define I_perm(x, y, z) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
#define I_bcnt(x, y, z) asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));
#define I_ffbl(x, y, z) asm volatile("v_ffbl_b32 %0, %1" : "=v"(x) : "v"(y));
#define I_cndmask(x, y, z) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x) : "v"(y), "v"(z));
#define I_cmp(x, y, z) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(y), "v"(z) : "vcc");
#define I_cmpsg(x, y, z) asm volatile("v_cmp_lt_u32 s[20:21], %0, %1" : : "v"(y), "v"(z) : "s20", "s21");
#define I_dot4(x, y, z) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
#define I_sad(x, y, z) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
#define I_mul24(x, y, z) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));
#define I_mad24(x, y, z) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
#define I_mullo(x, y, z) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));
#define I_min3(x, y, z) asm volatile("v_min3_u32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
#define I_min(x, y, z) asm volatile("v_min_u32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));
#define I_pkadd(x, y, z) asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));
#define I_fma(x, y, z) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
#define I_fmac(x, y, z) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
#define I_addf(x, y, z) asm volatile("v_add_f32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));
#define I_movdpp(x, y, z) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y));
#define I_xorsdwa(x, y, z) asm volatile("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(x) : "v"(y), "v"(z));
#define I_andsgpr(x, y, z) asm volatile("v_and_b32 %0, s8, %1" : "=v"(x) : "v"(y));
#define I_sand(x, y, z) asm volatile("s_and_b32 s20, s20, s21" : : : "s20", "scc");
#define I_snop(x, y, z) asm volatile("s_nop 0");
// pairs
#define I_and_s(x, y, z) asm volatile("v_and_b32 %0, %1, %2\n s_and_b32 s20, s20, s21" : "+v"(x) : "v"(y), "v"(z) : "s20", "scc");
#define I_bitop3_s(x, y, z) asm volatile("v_bitop3_b32 %0, %1, %2, %0 bitop3:0x96\n s_and_b32 s20, s20, s21" : "+v"(x) : "v"(y), "v"(z) : "s20", "scc");
#define I_shl64(x, y, z) asm volatile("v_lshlrev_b64 v[100:101], 3, v[102:103]" : : : "v100", "v101");
#define I_readlane(x, y, z) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(y) : "s20");
#define I_readfirst(x, y, z) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(y) : "s20");

#define ALL(X) X(and) X(xor) X(or) X(xorlit) X(andinl) X(not) X(mov) X(add) X(sub) X(shl) X(shr) X(shrv) X(alignbit) X(alignbyte) \
  X(bfe) X(bfi) X(andor) X(or3) X(lshlor) X(lshladd) X(add3) X(xad) X(bitop3) X(perm) X(bcnt) X(ffbl) X(cndmask) X(cmp) X(cmpsg) \
  X(dot4) X(sad) X(mul24) X(mad24) X(mullo) X(min3) X(min) X(pkadd) X(fma) X(fmac) X(addf) X(movdpp) X(xorsdwa) X(andsgpr) \
  X(sand) X(snop) X(and_s) X(bitop3_s) X(shl64) X(readlane) X(readfirst)

#define DEF_(N) DEF(N, "")
ALL(DEF_)

typedef void (*kern_t)(uint32_t*, uint32_t, int);
struct Test { const char* name; kern_t fn; int per; };
#define ENT_(N) {#N, k_##N, 1},
static Test tests[] = {ALL(ENT_)};

__global__ void clk_probe(unsigned long long* out) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long c0 = __builtin_readcyclecounter();
  unsigned long long r1 = r0;
  uint32_t x = threadIdx.x;
  while (r1 - r0 < 100000ull) {
    for (int i = 0; i < 256; ++i) x = x * 1664525u + 1013904223u;
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = x; }
}

int main() {
  uint32_t* out;
  CHECK(hipMalloc(&out, 256 * 8 * 256 * 4 * 4));
  unsigned long long* d;
  CHECK(hipMalloc(&d, 24));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 2000;
  // warm the clock
  for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k_fma, dim3(1024), dim3(256), 0, 0, out, 1u, iters);
  CHECK(hipDeviceSynchronize());
  hipLaunchKernelGGL(clk_probe, dim3(1), dim3(64), 0, 0, d);
  unsigned long long h[3];
  CHECK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
  const double ghz = 0.1 * (double)h[0] / (double)h[1];
  printf("shader clock %.3f GHz\n", ghz);
  printf("%-10s %8s %8s %8s %8s   (cycles per wave-instruction per SIMD at 1/2/3/4 waves per SIMD)\n", "op", "w1", "w2", "w3", "w4");
  for (const Test& t : tests) {
    printf("%-10s", t.name);
    for (int wps = 1; wps <= 4; ++wps) {
      dim3 grid(256 * wps), block(256);
      hipLaunchKernelGGL(t.fn, grid, block, 0, 0, out, 1u, 10);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(t.fn, grid, block, 0, 0, out, 2u, iters);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double instr_per_simd = (double)wps * iters * 64;
      printf(" %8.2f", ms * 1e6 / instr_per_simd * ghz);
    }
    printf("\n");
  }
  return 0;
}
