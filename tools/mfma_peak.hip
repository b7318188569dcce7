// Dev tool: sustained rate of back-to-back v_mfma_f32_32x32x2_f32 on every CU (random operands).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c) for (int t = 0; t < 16; ++t) acc[c][t] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 2e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[3], 0, 0, 0);
    }
    a = a * 0.999f + 1e-4f;
  }
  float s = 0.f;
  for (int c = 0; c < 4; ++c) for (int t = 0; t < 16; ++t) s += acc[c][t];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wg : {256, 512, 1024}) {
    int iters = 20000;
    k<<<wg, 256>>>(out, 1000, 0.3f, 0.7f); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<wg, 256>>>(out, iters, 0.3f, 0.7f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)wg * 4 /*waves*/ * iters * 64.0 * 4096.0;
    printf("wgs=%d  %.2f ms  %.1f TFLOP/s  (implied clock if 256 flop/clk/CU: %.2f GHz)\n", wg, ms, flop / ms / 1e9,
           flop / ms / 1e6 / (256.0 * 256.0) * (wg > 256 ? 256.0 / 256.0 : 1.0));
  }
  return 0;
}
