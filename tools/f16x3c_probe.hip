// Dev tool (round 4): is a 24-bit product cheaper than six bf16 passes?
//
// Every fp32 operand x (scaled by a power of two so that its block maximum is in [2^13, 2^14)) is split EXACTLY into
//   x = h + l + t,   h = fp16(x), l = fp16(x - h), t = x - h - l   (|l| <= 2^-11 |x|, |t| <= 2^-23 |x|)
// and a product a*b = hh + (hl + lh) + (ll + ht + th) + O(2^-34).  The first three cross terms need fp16 pieces (they are
// what the f16x3 mode computes: three v_mfma_f32_16x16x32_f16 passes, exact products).  The last three have weight
// <= 2^-22, so four significant bits of each factor are enough to evaluate them to 2^-26: they fit the block-scaled
// v_mfma_scale_f32_16x16x128_f8f6f4 with fp8 (e4m3: twice the bf16 rate) or fp6 (e2m3: four times) pieces, K = 128 per
// instruction, accumulating into the SAME fp32 accumulator (the E8M0 block scales undo the pieces' power-of-two scales).
//
// This probe (1) pins the operand lane maps of the scaled instruction with exact data, (2) checks the device conversion
// instructions against host encoders, (3) measures the error of [256 x 128] x [128 x 128] products against fp64 for:
// f32-input MFMA, bf16x6, f16x3, f16x3 + fp8 correction, f16x3 + fp6 correction, on several operand distributions,
// (4) times register-resident loops of the instruction mixes.
//   hipcc --offload-arch=gfx950 -O3 -o tools/f16x3c_probe tools/f16x3c_probe.hip && tools/f16x3c_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <random>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// the builtin takes 8 dwords, fp6 uses the first 6; a 6-element ext_vector_type is padded to 32 bytes in memory, so the
// 24-byte fragments are read dword by dword
__device__ __forceinline__ i32x8 load6(const i32x6* base, int idx) {
  const int* p = reinterpret_cast<const int*>(base) + 6 * idx;
  return i32x8{p[0], p[1], p[2], p[3], p[4], p[5], 0, 0};
}
__device__ __forceinline__ i32x8 widen6(i32x6 v) { return i32x8{v[0], v[1], v[2], v[3], v[4], v[5], 0, 0}; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---------------- host encoders (round to nearest even, saturating) ----------------
static float rne_to_grid(float v, float step) { return nearbyintf(v / step) * step; }
// e2m3: +-{0, .125 .. .875 (subnormal), 1 .. 1.875, 2 .. 3.75, 4 .. 7.5}
static uint8_t enc_fp6(float v) {
  uint8_t s = v < 0 ? 0x20 : 0; float a = fabsf(v);
  if (!(a == a)) return 0;
  if (a > 7.5f) a = 7.5f;
  int E; float step;
  if (a < 1.f) { E = 0; step = 0.125f; } else if (a < 2.f) { E = 1; step = 0.125f; } else if (a < 4.f) { E = 2; step = 0.25f; } else { E = 3; step = 0.5f; }
  float q = rne_to_grid(a, step);
  if (q > 7.5f) q = 7.5f;
  if (q >= 4.f) E = 3; else if (q >= 2.f) E = 2; else if (q >= 1.f) E = 1; else E = 0;
  int M = E == 0 ? (int)(q * 8.f) : (int)((q / ldexpf(1.f, E - 1) - 1.f) * 8.f);
  return s | (E << 3) | M;
}
static float dec_fp6(uint8_t c) {
  int s = c & 0x20, E = (c >> 3) & 3, M = c & 7;
  float v = E == 0 ? M / 8.f : ldexpf(1.f + M / 8.f, E - 1);
  return s ? -v : v;
}
// OCP e4m3fn: bias 7, subnormals 2^-6 * M/8, max 448, no inf
static uint8_t enc_fp8(float v) {
  uint8_t s = v < 0 ? 0x80 : 0; float a = fabsf(v);
  if (a > 448.f) a = 448.f;
  if (a < ldexpf(1.f, -10)) return s;   // below half of the smallest subnormal (2^-9)
  int e; frexpf(a, &e); e -= 1;          // a = 1.xxx * 2^e
  if (e < -6) e = -6;
  float step = ldexpf(1.f, e - 3);
  float q = rne_to_grid(a, step);
  if (q > 448.f) q = 448.f;
  frexpf(q, &e); e -= 1;
  if (q < ldexpf(1.f, -6)) { int M = (int)(q / ldexpf(1.f, -9)); return s | M; }
  int E = e + 7; int M = (int)((q / ldexpf(1.f, e) - 1.f) * 8.f);
  return s | (E << 3) | M;
}
static float dec_fp8(uint8_t c) {
  int s = c & 0x80, E = (c >> 3) & 15, M = c & 7;
  float v = E == 0 ? ldexpf(M / 8.f, -6) : ldexpf(1.f + M / 8.f, E - 7);
  return s ? -v : v;
}
// 32 six-bit codes -> 6 dwords, element j at bits [6j, 6j + 6) of the little-endian 192-bit group
static void pack_fp6(const uint8_t* c, uint32_t* w) {
  memset(w, 0, 24);
  for (int j = 0; j < 32; ++j) {
    int bit = 6 * j;
    uint64_t v = (uint64_t)(c[j] & 63) << (bit & 31);
    w[bit >> 5] |= (uint32_t)v;
    if ((bit & 31) > 26) w[(bit >> 5) + 1] |= (uint32_t)(v >> 32);
  }
}

// ---------------- (1) lane maps of the scaled instruction ----------------
// A [16][128], B [128][16] as prepared per-lane fragments: lane l holds A[l & 15][32 (l >> 4) + j], B[32 (l >> 4) + j][l & 15]
__global__ void k_scaled_fp8(const i32x8* A, const i32x8* B, float* C, int sa, int sb) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A[l], B[l], acc, 0, 0, 0, sa, 0, sb);
  for (int t = 0; t < 4; ++t) C[(4 * (l >> 4) + t) * 16 + (l & 15)] = acc[t];   // row = 4 (l >> 4) + t, col = l & 15
}
__global__ void k_scaled_fp6(const i32x6* A, const i32x6* B, float* C, int sa, int sb) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(load6(A, l), load6(B, l), acc, 2, 2, 0, sa, 0, sb);
  for (int t = 0; t < 4; ++t) C[(4 * (l >> 4) + t) * 16 + (l & 15)] = acc[t];
}
// ---------------- (2) device conversions ----------------
__global__ void k_cvt(const float* x, uint32_t* o8, uint32_t* o6, float scale) {
  const int l = threadIdx.x;
  const float* v = x + 32 * l;
  for (int j = 0; j < 8; ++j) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * j + 0], v[4 * j + 1], w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * j + 2], v[4 * j + 3], w, true);
    o8[8 * l + j] = w;
  }
  f32x16 v0, v1;
  for (int j = 0; j < 16; ++j) { v0[j] = v[j]; v1[j] = v[16 + j]; }
  u32x6 r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(v0, v1, scale);
  for (int j = 0; j < 6; ++j) o6[6 * l + j] = r[j];
}

// ---------------- (3) accuracy: one wave per 16 x 16 output tile, K = 128 ----------------
// prepared on the host: per (row block, lane) fragments of every piece
struct Frags {
  const f16x8* ah; const f16x8* al;          // [rb][s][lane]  16x16x32 A fragments (row l&15, k = 32 s + 8 (l>>4) + j)
  const f16x8* bh; const f16x8* bl;          // [cb][s][lane]
  const bf16x8 *a1, *a2, *a3, *b1, *b2, *b3; // bf16 three-way split, same indexing
  const i32x8 *al8, *ah8, *at8, *bl8, *bt8, *bh8;   // [rb or cb][lane]  fp8 pieces, k = 32 (l>>4) + j
  const i32x6 *al6, *ah6, *at6, *bl6, *bt6, *bh6;   // fp6 pieces
  const float* arow;  // [rows] inverse row scale of A  (f16 modes)
  float binv;         // inverse tensor scale of B
  int sc8_l, sc8_h, sc8_t, sc6_l, sc6_h, sc6_t;   // E8M0 codes undoing the pieces' scales
};
// mode: 0 f32 MFMA, 1 bf16x6, 2 f16x3, 3 f16x3 + fp8, 4 f16x3 + fp6
__global__ void k_prod(Frags f, const float* A, const float* B, float* C, int mode) {
  const int l = threadIdx.x, rb = blockIdx.x, cb = blockIdx.y;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // computed transposed like the product kernels: D[c, n] = sum_k B^T[c, k] A^T[k, n]; here simply A as the MFMA's A
  if (mode == 0) {
    for (int k = 0; k < 128; k += 4)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(16 * rb + (l & 15)) * 128 + k + (l >> 4)], B[(k + (l >> 4)) * 128 + 16 * cb + (l & 15)], acc, 0, 0, 0);
  } else if (mode == 1) {
    for (int s = 0; s < 4; ++s) {
      const int ia = (rb * 4 + s) * 64 + l, ib = (cb * 4 + s) * 64 + l;
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a3[ia], f.b1[ib], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a1[ia], f.b3[ib], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a2[ia], f.b2[ib], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a2[ia], f.b1[ib], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a1[ia], f.b2[ib], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a1[ia], f.b1[ib], acc, 0, 0, 0);
    }
  } else {
    const int ia8 = rb * 64 + l, ib8 = cb * 64 + l;
    if (mode == 3) {   // smallest terms first
      acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(f.at8[ia8], f.bh8[ib8], acc, 0, 0, 0, f.sc8_t, 0, f.sc8_h);
      acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(f.ah8[ia8], f.bt8[ib8], acc, 0, 0, 0, f.sc8_h, 0, f.sc8_t);
      acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(f.al8[ia8], f.bl8[ib8], acc, 0, 0, 0, f.sc8_l, 0, f.sc8_l);
    } else if (mode == 4) {
      acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(load6(f.at6, ia8), load6(f.bh6, ib8), acc, 2, 2, 0, f.sc6_t, 0, f.sc6_h);
      acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(load6(f.ah6, ia8), load6(f.bt6, ib8), acc, 2, 2, 0, f.sc6_h, 0, f.sc6_t);
      acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(load6(f.al6, ia8), load6(f.bl6, ib8), acc, 2, 2, 0, f.sc6_l, 0, f.sc6_l);
    }
    for (int s = 0; s < 4; ++s) {
      const int ia = (rb * 4 + s) * 64 + l, ib = (cb * 4 + s) * 64 + l;
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.al[ia], f.bh[ib], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.ah[ia], f.bl[ib], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.ah[ia], f.bh[ib], acc, 0, 0, 0);
    }
  }
  for (int t = 0; t < 4; ++t) {
    const int row = 16 * rb + 4 * (l >> 4) + t, col = 16 * cb + (l & 15);
    float v = acc[t];
    if (mode >= 2) v *= f.arow[row] * f.binv;
    C[row * 128 + col] = v;
  }
}

// ---------------- (4) throughput of the instruction mixes (operands in registers) ----------------
// per iteration: NF16 x (16x16x32 f16) on 4 accumulators + NC x (16x16x128 scaled, fmt) ; 2 waves per SIMD, every CU
template <int NF16, int NC, int FMT>
__global__ __launch_bounds__(512, 2) void k_rate(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x16 acc32[2];
  for (int i = 0; i < 2; ++i) for (int t = 0; t < 16; ++t) acc32[i][t] = 0.f;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j - 3.f); }
  i32x8 ca, cb;
  for (int j = 0; j < 8; ++j) { ca[j] = 0x38383838 + threadIdx.x + j; cb[j] = 0x30303030 + j; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NF16; ++i) {
      if constexpr (FMT >= 3) acc32[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc32[i & 1], 0, 0, 0);
      else acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i & 7], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      if constexpr (FMT == 0) acc[i & 7] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ca, cb, acc[i & 7], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      else if constexpr (FMT == 2) {   // mixed: A fp8 (8 dwords), B fp6 (6 dwords)
        i32x6 b6 = {cb[0], cb[1], cb[2], cb[3], cb[4], cb[5]};
        acc[i & 7] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ca, widen6(b6), acc[i & 7], 0, 2, 0, 0, 0, 0);
      } else if constexpr (FMT == 3) {   // 32x32x64 fp6 (unscaled)
        i32x6 a6 = {ca[0], ca[1], ca[2], ca[3], ca[4], ca[5]}, b6 = {cb[0], cb[1], cb[2], cb[3], cb[4], cb[5]};
        acc32[i & 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(widen6(a6), widen6(b6), acc32[i & 1], 2, 2, 0, 0, 0, 0);
      } else if constexpr (FMT == 4) {   // 32x32x64 mixed A fp8 / B fp6
        i32x6 b6 = {cb[0], cb[1], cb[2], cb[3], cb[4], cb[5]};
        acc32[i & 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ca, widen6(b6), acc32[i & 1], 0, 2, 0, 0, 0, 0);
      } else {
        i32x6 a6 = {ca[0], ca[1], ca[2], ca[3], ca[4], ca[5]}, b6 = {cb[0], cb[1], cb[2], cb[3], cb[4], cb[5]};
        acc[i & 7] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(widen6(a6), widen6(b6), acc[i & 7], 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  s += acc32[0][0] + acc32[1][5];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NF16, int NC, int FMT>
static float time_rate(const char* name, float* dout) {
  const int iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_rate<NF16, NC, FMT>), dim3(256), dim3(512), 0, 0, dout, 100);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_rate<NF16, NC, FMT>), dim3(256), dim3(512), 0, 0, dout, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  // cycles per iteration per SIMD at a nominal 2.4 GHz: two waves per SIMD share the pipe
  const double per_iter_us = ms * 1e3 / iters;
  printf("  %-34s %8.3f ms  %7.3f us/iter (2 waves/SIMD)  = %6.1f cycles@2.4GHz per wave-iteration-pair\n", name, ms, per_iter_us, per_iter_us * 2400.0);
  return ms;
}

template <class T> static T* to_dev(const std::vector<T>& v) { T* d; CK(hipMalloc(&d, v.size() * sizeof(T) + 16)); CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }

static void pow2_scale(float m, float& s, float& inv) {   // largest magnitude into [2^13, 2^14)
  int e; frexpf(m > 0 ? m : 1.f, &e);                     // m = f * 2^e, f in [0.5, 1)
  s = ldexpf(1.f, 14 - e); inv = ldexpf(1.f, e - 14);
}

int main() {
  // ---------- (1) lane maps ----------
  {
    std::mt19937 g(7);
    const float vals[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};   // exact in e2m3 and e4m3
    std::vector<float> A(16 * 128), B(128 * 16);
    for (auto& v : A) v = vals[g() & 7] * ((g() & 1) ? -1.f : 1.f);
    for (auto& v : B) v = vals[g() & 7] * ((g() & 1) ? -1.f : 1.f);
    std::vector<float> ref(256, 0.f);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 128; ++k) s += (double)A[i * 128 + k] * B[k * 16 + j]; ref[i * 16 + j] = (float)s; }
    std::vector<uint32_t> a8(64 * 8), b8(64 * 8), a6(64 * 6), b6(64 * 6);
    for (int l = 0; l < 64; ++l) {
      uint8_t ca[32], cb[32], da[32], db[32];
      for (int j = 0; j < 32; ++j) {
        const int k = 32 * (l >> 4) + j;
        ca[j] = enc_fp8(A[(l & 15) * 128 + k]); cb[j] = enc_fp8(B[k * 16 + (l & 15)]);
        da[j] = enc_fp6(A[(l & 15) * 128 + k]); db[j] = enc_fp6(B[k * 16 + (l & 15)]);
      }
      memcpy(&a8[8 * l], ca, 32); memcpy(&b8[8 * l], cb, 32);
      pack_fp6(da, &a6[6 * l]); pack_fp6(db, &b6[6 * l]);
    }
    float* dC; CK(hipMalloc(&dC, 1024));
    std::vector<float> C(256);
    auto report = [&](const char* name, float expect_mult) {
      CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
      double e = 0; for (int i = 0; i < 256; ++i) e = fmax(e, fabs(C[i] - expect_mult * ref[i]));
      printf("lane map %-28s max |C - ref| = %g   (ref max %g)\n", name, e, *std::max_element(ref.begin(), ref.end()));
    };
    uint32_t *da8 = to_dev(a8), *db8 = to_dev(b8), *da6 = to_dev(a6), *db6 = to_dev(b6);
    hipLaunchKernelGGL(k_scaled_fp8, dim3(1), dim3(64), 0, 0, (const i32x8*)da8, (const i32x8*)db8, dC, 0x7f7f7f7f, 0x7f7f7f7f);
    report("fp8 e4m3, scales 1", 1.f);
    hipLaunchKernelGGL(k_scaled_fp8, dim3(1), dim3(64), 0, 0, (const i32x8*)da8, (const i32x8*)db8, dC, 0x79797979, 0x81818181);
    report("fp8, scales 2^-6 * 2^2", 1.f / 16.f);
    hipLaunchKernelGGL(k_scaled_fp6, dim3(1), dim3(64), 0, 0, (const i32x6*)da6, (const i32x6*)db6, dC, 0x7f7f7f7f, 0x7f7f7f7f);
    report("fp6 e2m3, scales 1", 1.f);
    hipLaunchKernelGGL(k_scaled_fp6, dim3(1), dim3(64), 0, 0, (const i32x6*)da6, (const i32x6*)db6, dC, 0x7c7c7c7c, 0x7f7f7f7f);
    report("fp6, scale_a 2^-3", 1.f / 8.f);
  }
  // ---------- (2) device conversions ----------
  {
    std::mt19937 g(11); std::normal_distribution<float> nd(0.f, 2.f);
    std::vector<float> x(64 * 32);
    for (auto& v : x) v = nd(g);
    x[0] = 7.4f; x[1] = 7.6f; x[2] = 7.9f; x[3] = 9.f; x[4] = 0.06f; x[5] = 0.0625f; x[6] = 0.07f; x[7] = 0.1875f; x[8] = 500.f; x[9] = 1e-3f; x[10] = 0.0015f;
    float* dx = to_dev(x); uint32_t *d8, *d6; CK(hipMalloc(&d8, 64 * 32)); CK(hipMalloc(&d6, 64 * 24));
    for (float scale : {1.f, 4.f}) {
      hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, dx, d8, d6, scale);
      std::vector<uint32_t> o8(64 * 8), o6(64 * 6);
      CK(hipMemcpy(o8.data(), d8, 64 * 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(o6.data(), d6, 64 * 24, hipMemcpyDeviceToHost));
      int bad8 = 0, bad6_seq = 0, bad6_div = 0, bad6_mul = 0, bad6_il = 0;
      for (int l = 0; l < 64; ++l) {
        const uint8_t* c8 = reinterpret_cast<const uint8_t*>(&o8[8 * l]);
        for (int j = 0; j < 32; ++j) {
          if (c8[j] != enc_fp8(x[32 * l + j])) { if (bad8 < 6 && scale == 1.f) printf("  fp8 mismatch x=%g dev=%02x (%g) host=%02x (%g)\n", x[32 * l + j], c8[j], dec_fp8(c8[j]), enc_fp8(x[32 * l + j]), dec_fp8(enc_fp8(x[32 * l + j]))); ++bad8; }
          const int bit = 6 * j; const uint64_t w = o6[6 * l + (bit >> 5)] | ((uint64_t)((bit >> 5) < 5 ? o6[6 * l + (bit >> 5) + 1] : 0) << 32);
          const uint8_t c6 = (w >> (bit & 31)) & 63;
          const float xs = x[32 * l + j];
          if (c6 != enc_fp6(xs / scale)) ++bad6_div;
          if (c6 != enc_fp6(xs * scale)) ++bad6_mul;
          const int jj = (j & 1) * 16 + (j >> 1);   // interleaved hypothesis: element j from (src j&1)[j>>1]
          if (c6 != enc_fp6(x[32 * l + jj] / scale)) ++bad6_il;
          if (scale == 1.f && c6 != enc_fp6(xs)) { if (bad6_seq < 6) printf("  fp6 mismatch x=%g dev=%02x (%g) host=%02x (%g)\n", xs, c6, dec_fp6(c6), enc_fp6(xs), dec_fp6(enc_fp6(xs))); ++bad6_seq; }
        }
      }
      printf("cvt scale=%g: fp8 mismatches %d/2048 ; fp6 sequential: divide-by-scale %d, multiply-by-scale %d ; interleaved+divide %d\n", scale, bad8, bad6_div, bad6_mul, bad6_il);
    }
  }
  // ---------- (3) accuracy ----------
  const int R = 256;
  for (int dist = 0; dist < 5; ++dist) {
    std::mt19937 g(100 + dist); std::normal_distribution<float> nd(0.f, 1.f); std::uniform_real_distribution<float> ud(0.f, 1.f);
    std::vector<float> A(R * 128), B(128 * 128);
    const char* dname[] = {"gaussian x gaussian", "tanh(2g) x 0.05 g", "wide range: g * 10^U(-3,0) x g", "positive x positive", "one huge element per row x g"};
    for (int i = 0; i < R * 128; ++i) {
      float v = nd(g);
      if (dist == 1) v = tanhf(2.f * v);
      if (dist == 2) v *= powf(10.f, -3.f * ud(g));
      if (dist == 3) v = 0.25f + ud(g);
      if (dist == 4 && (i & 127) == 5) v *= 1000.f;
      A[i] = v;
    }
    for (auto& v : B) { v = nd(g); if (dist == 1) v *= 0.05f; if (dist == 3) v = 0.25f + ud(g); }
    std::vector<double> ref(R * 128), aref(R * 128);
    for (int i = 0; i < R; ++i) for (int j = 0; j < 128; ++j) { double s = 0, sa = 0; for (int k = 0; k < 128; ++k) { double p = (double)A[i * 128 + k] * B[k * 128 + j]; s += p; sa += fabs(p); } ref[i * 128 + j] = s; aref[i * 128 + j] = sa; }
    // pieces
    float bmax = 0; for (auto v : B) bmax = fmaxf(bmax, fabsf(v));
    float sb, ib; pow2_scale(bmax, sb, ib);
    std::vector<float> arow(R);
    const int RB = R / 16, CB = 8;
    std::vector<_Float16> ah(RB * 4 * 64 * 8), al(ah.size()), bh(CB * 4 * 64 * 8), bl(bh.size());
    std::vector<__bf16> a1(ah.size()), a2(ah.size()), a3(ah.size()), b1(bh.size()), b2(bh.size()), b3(bh.size());
    std::vector<uint32_t> al8(RB * 64 * 8), ah8(al8.size()), at8(al8.size()), bl8(CB * 64 * 8), bh8(bl8.size()), bt8(bl8.size());
    std::vector<uint32_t> al6(RB * 64 * 6), ah6(al6.size()), at6(al6.size()), bl6(CB * 64 * 6), bh6(bl6.size()), bt6(bl6.size());
    // piece scales: fp8 (max 448): h * 2^-6, l * 2^6, t * 2^18 ; fp6 (max 7.5): h * 2^-11, l * 2^0 (|l| <= 4), t * 2^12 (|t| <= 2^-10 -> 4)
    auto split = [&](float xs, float& h, float& l, float& t) { h = (float)(_Float16)xs; l = (float)(_Float16)(xs - h); t = xs - h - l; };
    auto prep = [&](bool isA, int blk, std::vector<_Float16>& vh, std::vector<_Float16>& vl, std::vector<__bf16>& v1, std::vector<__bf16>& v2, std::vector<__bf16>& v3,
                    std::vector<uint32_t>& l8, std::vector<uint32_t>& h8, std::vector<uint32_t>& t8, std::vector<uint32_t>& l6, std::vector<uint32_t>& h6, std::vector<uint32_t>& t6) {
      for (int l = 0; l < 64; ++l) {
        const int rc = 16 * blk + (l & 15);
        float sc = sb;
        if (isA) { float m = 0; for (int k = 0; k < 128; ++k) m = fmaxf(m, fabsf(A[rc * 128 + k])); float inv; pow2_scale(m, sc, inv); arow[rc] = inv; }
        uint8_t cl8[32], ch8[32], ct8[32], cl6[32], ch6[32], ct6[32];
        for (int j = 0; j < 32; ++j) {
          const int k = 32 * (l >> 4) + j;
          const float x = isA ? A[rc * 128 + k] : B[k * 128 + rc];
          float h, lo, t; split(x * sc, h, lo, t);
          cl8[j] = enc_fp8(ldexpf(lo, 6)); ch8[j] = enc_fp8(ldexpf(h, -6)); ct8[j] = enc_fp8(ldexpf(t, 18));
          cl6[j] = enc_fp6(lo); ch6[j] = enc_fp6(ldexpf(h, -11)); ct6[j] = enc_fp6(ldexpf(t, 12));
        }
        memcpy(&l8[(blk * 64 + l) * 8], cl8, 32); memcpy(&h8[(blk * 64 + l) * 8], ch8, 32); memcpy(&t8[(blk * 64 + l) * 8], ct8, 32);
        pack_fp6(cl6, &l6[(blk * 64 + l) * 6]); pack_fp6(ch6, &h6[(blk * 64 + l) * 6]); pack_fp6(ct6, &t6[(blk * 64 + l) * 6]);
        for (int s = 0; s < 4; ++s) for (int j = 0; j < 8; ++j) {
          const int k = 32 * s + 8 * (l >> 4) + j;
          const float x = isA ? A[rc * 128 + k] : B[k * 128 + rc];
          float h, lo, t; split(x * sc, h, lo, t);
          const size_t o = ((size_t)(blk * 4 + s) * 64 + l) * 8 + j;
          vh[o] = (_Float16)h; vl[o] = (_Float16)lo;
          __bf16 p1 = (__bf16)x; float r1 = x - (float)p1; __bf16 p2 = (__bf16)r1; __bf16 p3 = (__bf16)(r1 - (float)p2);
          v1[o] = p1; v2[o] = p2; v3[o] = p3;
        }
      }
    };
    for (int rb = 0; rb < RB; ++rb) prep(true, rb, ah, al, a1, a2, a3, al8, ah8, at8, al6, ah6, at6);
    for (int cb = 0; cb < CB; ++cb) prep(false, cb, bh, bl, b1, b2, b3, bl8, bh8, bt8, bl6, bh6, bt6);
    Frags f;
    f.ah = (const f16x8*)to_dev(ah); f.al = (const f16x8*)to_dev(al); f.bh = (const f16x8*)to_dev(bh); f.bl = (const f16x8*)to_dev(bl);
    f.a1 = (const bf16x8*)to_dev(a1); f.a2 = (const bf16x8*)to_dev(a2); f.a3 = (const bf16x8*)to_dev(a3);
    f.b1 = (const bf16x8*)to_dev(b1); f.b2 = (const bf16x8*)to_dev(b2); f.b3 = (const bf16x8*)to_dev(b3);
    f.al8 = (const i32x8*)to_dev(al8); f.ah8 = (const i32x8*)to_dev(ah8); f.at8 = (const i32x8*)to_dev(at8);
    f.bl8 = (const i32x8*)to_dev(bl8); f.bh8 = (const i32x8*)to_dev(bh8); f.bt8 = (const i32x8*)to_dev(bt8);
    f.al6 = (const i32x6*)to_dev(al6); f.ah6 = (const i32x6*)to_dev(ah6); f.at6 = (const i32x6*)to_dev(at6);
    f.bl6 = (const i32x6*)to_dev(bl6); f.bh6 = (const i32x6*)to_dev(bh6); f.bt6 = (const i32x6*)to_dev(bt6);
    f.arow = to_dev(arow); f.binv = ib;
    auto e8 = [](int k) { const unsigned c = 127 + k; return (int)(c | c << 8 | c << 16 | c << 24); };
    f.sc8_l = e8(-6); f.sc8_h = e8(6); f.sc8_t = e8(-18);
    f.sc6_l = e8(0); f.sc6_h = e8(11); f.sc6_t = e8(-12);
    float *dA = to_dev(A), *dB = to_dev(B), *dC; CK(hipMalloc(&dC, R * 128 * 4));
    printf("distribution %d: %s\n", dist, dname[dist]);
    const char* mname[] = {"f32-input MFMA", "bf16x6", "f16x3 (22-bit)", "f16x3 + fp8 correction", "f16x3 + fp6 correction"};
    for (int mode = 0; mode < 5; ++mode) {
      hipLaunchKernelGGL(k_prod, dim3(RB, CB), dim3(64), 0, 0, f, dA, dB, dC, mode);
      std::vector<float> C(R * 128); CK(hipMemcpy(C.data(), dC, R * 128 * 4, hipMemcpyDeviceToHost));
      double emax = 0, rmax = 0, sum2 = 0, mean = 0, worst_rel_abs = 0;
      for (int i = 0; i < R * 128; ++i) { const double e = C[i] - ref[i]; emax = fmax(emax, fabs(e)); rmax = fmax(rmax, fabs(ref[i])); sum2 += (e / aref[i]) * (e / aref[i]); mean += e / aref[i]; worst_rel_abs = fmax(worst_rel_abs, fabs(e) / aref[i]); }
      printf("  %-26s max-norm rel %.3e   err/sum|ab|: rms %.3e  max %.3e  mean %+.2e\n", mname[mode], emax / rmax, sqrt(sum2 / (R * 128)), worst_rel_abs, mean / (R * 128));
    }
  }
  // ---------- (4) throughput ----------
  {
    float* dout; CK(hipMalloc(&dout, 256 * 512 * 4));
    printf("instruction mixes (256 workgroups x 8 waves, operands in registers):\n");
    time_rate<12, 0, 0>("12 x f16 16x16x32 (f16x3)", dout);
    time_rate<24, 0, 0>("24 x f16 16x16x32 (six passes)", dout);
    time_rate<12, 3, 0>("12 x f16 + 3 x fp8 16x16x128", dout);
    time_rate<12, 3, 1>("12 x f16 + 3 x fp6 16x16x128", dout);
    time_rate<0, 12, 0>("12 x fp8 16x16x128", dout);
    time_rate<0, 12, 1>("12 x fp6 16x16x128", dout);
    time_rate<12, 3, 2>("12 x f16 + 3 x (A fp8, B fp6) 16x16x128", dout);
    time_rate<0, 12, 2>("12 x (A fp8, B fp6) 16x16x128", dout);
    time_rate<12, 0, 3>("12 x f16 32x32x16", dout);
    time_rate<12, 3, 3>("12 x f16 32x32x16 + 3 x fp6 32x32x64", dout);
    time_rate<12, 3, 4>("12 x f16 32x32x16 + 3 x (A fp8, B fp6) 32x32x64", dout);
  }
  return 0;
}
