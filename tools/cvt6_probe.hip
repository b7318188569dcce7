// Dev tool: cycles per 6-bit conversion instruction on gfx950 (one wave per SIMD, back-to-back, independent and dependent):
//   v_cvt_scalef32_pk32_bf6_f16 / _fp6_f16 / _bf6_bf16 (32 packed 16-bit values -> 6 dwords)
//   v_cvt_scalef32_2xpk16_bf6_f32 (2 x 16 floats -> 6 dwords)
// and beside a dependent chain of v_mfma_f32_32x32x16_f16 issued by a SECOND wave on the same SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/cvt6_probe tools/cvt6_probe.hip ; run: tools/cvt6_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef __bf16 bf16x32 __attribute__((ext_vector_type(32)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));

template <int KIND>   // 0 pk32_bf6_f16, 1 pk32_fp6_f16, 2 pk32_bf6_bf16, 3 2xpk16_bf6_f32, 4 v_add_f32 (reference)
__global__ __launch_bounds__(512) void probe(unsigned* out, long long* cyc, int iters, int mfma_waves) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f16x32 h;
  bf16x32 b;
  f32x16 f0, f1;
  for (int i = 0; i < 32; ++i) { h[i] = (_Float16)(lane + i); b[i] = (__bf16)(lane * 0.5f + i); }
  for (int i = 0; i < 16; ++i) { f0[i] = lane + i; f1[i] = lane - i; }
  u32x6 acc = {0, 0, 0, 0, 0, 0};
  float fa = lane;
  f32x16 macc = {};
  f16x8 ma, mb;
  for (int i = 0; i < 8; ++i) { ma[i] = (_Float16)(lane * 0.01f); mb[i] = (_Float16)(i * 0.1f); }
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (wave >= 4 && wave < 4 + mfma_waves) {          // waves 4..7 share SIMDs with waves 0..3
    for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) macc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ma, mb, macc, 0, 0, 0);
    }
  } else if (wave < 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        u32x6 r;
        if (KIND == 0) r = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(h, 4096.0f);
        else if (KIND == 1) r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(h, 1.0f);
        else if (KIND == 2) r = __builtin_amdgcn_cvt_scalef32_pk32_bf6_bf16(b, 0x1p-12f);
        else if (KIND == 3) r = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(f0, f1, 4096.0f);
        else { fa = fa * 1.0001f + 1.f; r = acc; }
        acc[0] ^= r[0]; acc[5] += r[5];
        h[k] = (_Float16)((float)h[k] + 1.f);         // make every conversion's input differ
        b[k] = (__bf16)((float)b[k] + 1.f);
        f0[k] += 1.f;
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[5] + (unsigned)fa + (unsigned)macc[0];
}

template <int KIND>
static void run(const char* name, int mfma_waves) {
  unsigned* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 200;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, out, cyc, iters, mfma_waves);
  hipDeviceSynchronize();
  long long h[256 * 8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double c = 0, m = 0;
  for (int b = 0; b < 256; ++b) { for (int w = 0; w < 4; ++w) c += h[b * 8 + w]; for (int w = 4; w < 8; ++w) m += h[b * 8 + w]; }
  printf("%-28s mfma partner waves %d: %7.1f cycles per conversion (wave 0-3);  partner: %6.1f cycles per MFMA\n", name, mfma_waves,
         c / (256 * 4) / (iters * 8.0), mfma_waves ? m / (256 * 4) / (iters * 32.0) : 0.0);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int mw = 0; mw <= 4; mw += 4) {
    run<4>("v_fma_f32 (reference)", mw);
    run<0>("cvt_scalef32_pk32_bf6_f16", mw);
    run<1>("cvt_scalef32_pk32_fp6_f16", mw);
    run<2>("cvt_scalef32_pk32_bf6_bf16", mw);
    run<3>("cvt_scalef32_2xpk16_bf6_f32", mw);
  }
  return 0;
}
