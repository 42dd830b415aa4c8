// bc_intrin.h -- the handful of gfx950 VALU instructions the lane code leans on.
// Under hipcc these are the hardware instructions; the plain-C++ forms exist only so that
// tests/emu can run the SAME lane logic on the host against the oracle (never a product path).
#pragma once
#if defined(__HIPCC_RTC__)
// run-time compilation (hiprtc): the HIP device builtins are predeclared, the C headers are not
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef unsigned int uint32_t;
typedef unsigned long long uint64_t;
#define BC_HD __device__ __forceinline__
#else
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BC_HD __host__ __device__ __forceinline__
#else
#define BC_HD inline
#endif
#endif

// Device tables are reached through explicit global-address-space pointers: a pointer rebuilt from an
// integer (or loaded from memory) is otherwise "generic", and the compiler then emits FLAT loads,
// which tie up the LDS counter as well as the vector-memory one.
#if defined(__HIP_DEVICE_COMPILE__)
#define BC_GLOBAL __attribute__((address_space(1)))
#else
#define BC_GLOBAL
#endif

namespace bc {

#if defined(__HIP_DEVICE_COMPILE__)
// v_alignbit_b32: ({hi,lo} >> (s & 31)) low 32 bits
BC_HD uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t s) { return __builtin_amdgcn_alignbit(hi, lo, s); }
// v_alignbyte_b32: ({hi,lo} >> 8*(s & 3)) low 32 bits
BC_HD uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t s) { return __builtin_amdgcn_alignbyte(hi, lo, s); }
// v_dot4_u32_u8: sum of the four byte products + c
BC_HD uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }
// v_perm_b32: selector bytes 0-3 pick bytes of lo, 4-7 bytes of hi
BC_HD uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// v_sad_u8: sum of absolute byte differences + c
BC_HD uint32_t sad_u8(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_sad_u8(a, b, c); }
BC_HD uint32_t popc(uint32_t x) { return __builtin_popcount(x); }
BC_HD uint32_t ctz(uint32_t x) { return __builtin_ctz(x); }
// v_bitop3_b32: any boolean function of three operands; bit (a*4 + b*2 + c) of TT is f(a,b,c)
template <int TT>
BC_HD uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, TT); }
#else
BC_HD uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t s) {
  return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (s & 31));
}
BC_HD uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t s) {
  return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8 * (s & 3)));
}
BC_HD uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) {
  for (int i = 0; i < 4; ++i) c += ((a >> (8 * i)) & 0xFF) * ((b >> (8 * i)) & 0xFF);
  return c;
}
BC_HD uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) {
  uint64_t src = (((uint64_t)hi) << 32) | lo;
  uint32_t r = 0;
  for (int i = 0; i < 4; ++i) {
    uint32_t s = (sel >> (8 * i)) & 0xFF;
    uint32_t b = s < 8 ? (uint32_t)((src >> (8 * s)) & 0xFF) : (s >= 0x0D ? 0xFFu : 0u);
    r |= b << (8 * i);
  }
  return r;
}
BC_HD uint32_t sad_u8(uint32_t a, uint32_t b, uint32_t c) {
  for (int i = 0; i < 4; ++i) {
    int x = (a >> (8 * i)) & 0xFF, y = (b >> (8 * i)) & 0xFF;
    c += (uint32_t)(x > y ? x - y : y - x);
  }
  return c;
}
BC_HD uint32_t popc(uint32_t x) { return (uint32_t)__builtin_popcount(x); }
BC_HD uint32_t ctz(uint32_t x) { return (uint32_t)__builtin_ctz(x); }
template <int TT>
BC_HD uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r = 0;
  for (int i = 0; i < 8; ++i) {
    if (!((TT >> i) & 1)) continue;
    r |= ((i & 4) ? a : ~a) & ((i & 2) ? b : ~b) & ((i & 1) ? c : ~c);
  }
  return r;
}
#endif

// truth tables for bitop3 (operand order a, b, c)
constexpr int kTT_Xor3 = 0x96, kTT_Xnor3 = 0x69, kTT_Maj = 0xE8, kTT_NotMaj = 0x17;
constexpr int kTT_AorBandC = 0xF8;  // a | (b & c)
constexpr int kTT_Xnor2ab = 0xC3;   // ~(a ^ b)      (c ignored)
constexpr int kTT_Nor2ab = 0x03;    // ~(a | b)      (c ignored)

// Keeps a wave-uniform `if` a real scalar branch: without it the compiler turns the branch into
// one v_cndmask per candidate value, which costs VALU issue slots in every lane.
#if defined(__HIP_DEVICE_COMPILE__)
#define BC_KEEP_BRANCH3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
#else
#define BC_KEEP_BRANCH3(a, b, c) (void)0
#endif

// bytes that are zero -> 0x80 in that byte
BC_HD uint32_t zero_bytes(uint32_t v) { return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u; }
// low `n` bits set, n in [0, 32]
// low 32 bits of a 24-bit x 24-bit product: v_mul_u32_u24, full rate (v_mul_lo_u32 is quarter rate)
// (the masks let the compiler prove the operand widths and pick that instruction)
BC_HD uint32_t mul24(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
// the compiler folds the nested form into v_min3_u32
BC_HD uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t m = a < b ? a : b;
  return m < c ? m : c;
}
BC_HD uint32_t lowmask(uint32_t n) { return n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u); }

}  // namespace bc
