// Shared device helpers of the split-bf16 MFMA kernels (bilinear.hip, edgez.hip): the exact 3-way bf16 split,
// LDS-DMA issue from inline asm, vector typedefs.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3_bf16(float a, __bf16& x1, __bf16& x2, __bf16& x3) {
  x1 = (__bf16)a;
  const float r1 = a - (float)x1;
  x2 = (__bf16)r1;
  x3 = (__bf16)(r1 - (float)x2);
}

// The same exact split for two values at once: each plane's pair is ONE v_cvt_pk_bf16_f32 and lands in its final
// 32-bit word (these files are built without the SLP vectoriser, which would otherwise do this pairing; element-wise
// code costs 11 VALU instructions per value instead of 6.5 -- and the kernels that split in their loops are bound by
// the SIMD's issue port).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& w1, unsigned& w2, unsigned& w3) {
  const bf16x2 x1 = __builtin_convertvector((f32x2){a, b}, bf16x2);
  const float ra = a - (float)x1[0], rb = b - (float)x1[1];
  const bf16x2 x2 = __builtin_convertvector((f32x2){ra, rb}, bf16x2);
  const float sa = ra - (float)x2[0], sb = rb - (float)x2[1];
  const bf16x2 x3 = __builtin_convertvector((f32x2){sa, sb}, bf16x2);
  w1 = __builtin_bit_cast(unsigned, x1);
  w2 = __builtin_bit_cast(unsigned, x2);
  w3 = __builtin_bit_cast(unsigned, x3);
}
// eight values -> the three bf16x8 fragment planes
__device__ __forceinline__ void split3_x8(const float (&v)[8], bf16x8& q1, bf16x8& q2, bf16x8& q3) {
  uint4 w1, w2, w3;
  split3_pair(v[0], v[1], w1.x, w2.x, w3.x);
  split3_pair(v[2], v[3], w1.y, w2.y, w3.y);
  split3_pair(v[4], v[5], w1.z, w2.z, w3.z);
  split3_pair(v[6], v[7], w1.w, w2.w, w3.w);
  q1 = __builtin_bit_cast(bf16x8, w1);
  q2 = __builtin_bit_cast(bf16x8, w2);
  q3 = __builtin_bit_cast(bf16x8, w3);
}

// global address = scalar base + per-lane 32-bit byte offset; LDS address = lds_addr + 16 (4) * lane
__device__ __forceinline__ void glds_b128(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void glds_b32(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

