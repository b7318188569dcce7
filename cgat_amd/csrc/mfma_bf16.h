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

