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

// ---- two-plane fp16 split ("f16x3": x*2^k = h + l with fp16 pieces, 22 significant bits; a product is
// a1b2 + a2b1 + a1b1 = three fp16 MFMA passes, all partial products exact in fp32, dropped term <= 2^-22).
// fp16 has 5 exponent bits, so every operand is scaled by a power of two that puts its largest magnitude (per row,
// or per tensor for operands indexed by the reduction index) into [2^13, 2^14): the low plane of any element within
// 2^-16 of that maximum is still a normal fp16 number, and what is lost on smaller elements is below 2^-37 of the
// maximum.  The inverse scales are folded into the fp32 multiplier the kernels already apply to the sums.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// power-of-two scale for a block whose largest magnitude is m (m >= 0): s = 2^(140 - E(m)), inv = 1/s (both exact)
__device__ __forceinline__ void pow2_scale(float m, float& s, float& inv) {
  int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xff);
  e = e < 30 ? 30 : (e > 230 ? 230 : e);   // all-zero / denormal / huge blocks: any finite scale will do
  s = __builtin_bit_cast(float, (unsigned)(267 - e) << 23);
  inv = __builtin_bit_cast(float, (unsigned)(e - 13) << 23);
}
__device__ __forceinline__ void split2_pair_f16(float a, float b, unsigned& w1, unsigned& w2) {
  const f16x2 h = __builtin_convertvector((f32x2){a, b}, f16x2);
  const float ra = a - (float)h[0], rb = b - (float)h[1];
  const f16x2 l = __builtin_convertvector((f32x2){ra, rb}, f16x2);
  w1 = __builtin_bit_cast(unsigned, h);
  w2 = __builtin_bit_cast(unsigned, l);
}
// eight (already scaled) values -> the two fragment planes (stored in the 16-byte bf16x8 carrier type)
__device__ __forceinline__ void split2_x8_f16(const float (&v)[8], bf16x8& q1, bf16x8& q2) {
  uint4 w1, w2;
  split2_pair_f16(v[0], v[1], w1.x, w2.x);
  split2_pair_f16(v[2], v[3], w1.y, w2.y);
  split2_pair_f16(v[4], v[5], w1.z, w2.z);
  split2_pair_f16(v[6], v[7], w1.w, w2.w);
  q1 = __builtin_bit_cast(bf16x8, w1);
  q2 = __builtin_bit_cast(bf16x8, w2);
}
// sign flip of the eight 16-bit floats (bf16 or fp16) of a fragment: exact
__device__ __forceinline__ bf16x8 neg_x8(bf16x8 q) {
  uint4 w = __builtin_bit_cast(uint4, q);
  w.x ^= 0x80008000u; w.y ^= 0x80008000u; w.z ^= 0x80008000u; w.w ^= 0x80008000u;
  return __builtin_bit_cast(bf16x8, w);
}
// one 16x16x32 matrix-core pass on 16-byte fragments held in the bf16x8 carrier type
template <bool F16>
__device__ __forceinline__ f32x4 mma16(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ---- "f16x3c": the f16x3 product completed to 24-bit operands (round 4) ----
// With x (already scaled as above) split EXACTLY into x = h + l + t, h = fp16(x), l = fp16(x - h), t = x - h - l
// (|l| <= 2^-11 |x|; t is 0 or +-ulp32(x), i.e. the 24th bit), a product is
//      a b = [hh + hl + lh] + [ll + ht + th] + O(2^-34):
// the first bracket is what the three fp16 passes compute exactly; the second has weight <= 2^-22, so three or four
// significant bits of each factor evaluate it to 2^-26 -- it runs as three block-scaled v_mfma_scale_f32_16x16x128_f8f6f4
// instructions on 6-bit pieces (K = 128 per instruction at the cycles of ONE 16x16x32 fp16 pass: +25 % matrix time
// over f16x3 instead of +100 % for six bf16 passes), accumulating into the same fp32 accumulator (the E8M0 scale
// operands undo the pieces' power-of-two scales).  Piece formats: l as fp6 e2m3 (|l| <= 4: four bits), h and t as bf6
// e3m2 (three bits, nine binades; scales below).
// Measured against fp64 (tools/f16x3c_probe.hip): at or below the error of the six-pass bf16 split and of the f32-input
// MFMA on every operand distribution tried.
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef float f32x16v __attribute__((ext_vector_type(16)));
struct frag6 { unsigned w[6]; };          // 32 six-bit values of one lane: one operand of the 16x16x128 instruction
__device__ __forceinline__ i32x8 frag6_as8(const frag6& f) {   // the builtin takes 8 dwords; the 6-bit formats read 6
  i32x8 r;
  r[0] = f.w[0]; r[1] = f.w[1]; r[2] = f.w[2]; r[3] = f.w[3]; r[4] = f.w[4]; r[5] = f.w[5];
  return r;
}
// piece scales: l6 = fp6(l) (|l| <= 4), h6 = bf6(h * 2^-12) (< 4), t6 = bf6(t * 2^12) (|t| <= 2^-10 -> <= 4): every product
// of two pieces carries the scale 1, so the instruction needs no E8M0 scale operands (zero scale arguments select the
// unscaled v_mfma_f32_16x16x128_f8f6f4: no scale VGPRs).  bf6 spans 0.0625 .. 28: t (a power of two) is exact for every
// element within 2^-7 of its block's maximum, h keeps three bits within 2^-4 of it.
// prepared-T image of the contraction kernels in this form (bilinear.hip, prepare_T_f16c_kernel)
#define F16C_CHUNK16 1600                       // 16-byte pieces per chunk (a, column half, column-block pair): 25 KB
#define F16C_A_FLOATS (4 * F16C_CHUNK16 * 4)    // floats per `a` (four chunks)
// the three 6-bit images of 32 scaled values (the lane's K elements in ANY order: both operands of a product use this
// same function on identically ordered values, so the instruction's interleaving of its two sources does not matter)
__device__ __forceinline__ void f16c_pack32(const float (&x)[32], frag6& l6, frag6& h6, frag6& t6) {
  f32x16v l0, l1, h0, h1, t0, t1;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const float h = (float)(_Float16)x[j];
    const float r = x[j] - h;
    const float l = (float)(_Float16)r;
    const float t = r - l;
    if (j < 16) { l0[j] = l; h0[j] = h; t0[j] = t; } else { l1[j - 16] = l; h1[j - 16] = h; t1[j - 16] = t; }
  }
  const u32x6 a = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(l0, l1, 1.0f);          // the instruction divides by its scale
  const u32x6 b = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(h0, h1, 4096.0f);
  const u32x6 c = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(t0, t1, 0x1p-12f);
#pragma unroll
  for (int i = 0; i < 6; ++i) { l6.w[i] = a[i]; h6.w[i] = b[i]; t6.w[i] = c[i]; }
}
// acc += A_t B_h + A_h B_t + A_l B_l  (smallest first); A = the instruction's A operand (rows), B its B operand
__device__ __forceinline__ f32x4 f16c_mma_th(const frag6& at, const frag6& bh, f32x4 c) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(frag6_as8(at), frag6_as8(bh), c, 3, 3, 0, 0, 0, 0);
}
__device__ __forceinline__ f32x4 f16c_mma_ht(const frag6& ah, const frag6& bt, f32x4 c) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(frag6_as8(ah), frag6_as8(bt), c, 3, 3, 0, 0, 0, 0);
}
__device__ __forceinline__ f32x4 f16c_mma_ll(const frag6& al, const frag6& bl, f32x4 c) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(frag6_as8(al), frag6_as8(bl), c, 2, 2, 0, 0, 0, 0);
}

// ---- bf16 STORAGE of activations (the edge phase's Z / gZ in the "bf16" edge-storage mode, BASELINE configs[4]) ----
// four fp32 -> four bf16 (round to nearest even, two v_cvt_pk_bf16_f32) as 8 bytes, and back (exact)
__device__ __forceinline__ uint2 pack4_bf16(float4 v) {
  const bf16x2 a = __builtin_convertvector((f32x2){v.x, v.y}, bf16x2);
  const bf16x2 b = __builtin_convertvector((f32x2){v.z, v.w}, bf16x2);
  return make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
}
__device__ __forceinline__ float4 unpack4_bf16(uint2 u) {
  return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                     __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
}
__device__ __forceinline__ float4 load4_bf16(const __bf16* p) { return unpack4_bf16(*reinterpret_cast<const uint2*>(p)); }
__device__ __forceinline__ void store4_bf16(__bf16* p, float4 v) { *reinterpret_cast<uint2*>(p) = pack4_bf16(v); }

// at most one atomic per workgroup (thousands of same-address atomics cost more than the pass itself)
__device__ __forceinline__ void block_absmax_commit(float m, float* out) {
  __shared__ float wm[16];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, wm[w]);
    // the slot only grows while a kernel runs (it is zeroed by a kernel before the launch: rowops.hip, fill_launch), so a
    // read that is already >= m makes the atomic unnecessary.  The read is an agent-scope atomic load, so that it is
    // served where the other XCDs' atomics land and not by a line this XCD's L2 fetched earlier in the launch.
    if (m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(reinterpret_cast<unsigned*>(out), __builtin_bit_cast(unsigned, m));
  }
}

// counted wait on the vector-memory counter.  vmcnt retires IN ORDER and counts stores as well as loads: vmcnt(N)
// returns when everything except the N youngest operations has completed, so N must cover every operation issued
// AFTER the one being waited for, and an operation that need not be complete yet must be younger than it.
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

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

