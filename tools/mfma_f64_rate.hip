// Dev tool (GPU box): issue rate of v_mfma_f64_16x16x4_f64 against v_mfma_f32_16x16x4_f32 on gfx950 -- one wave per SIMD,
// NACC independent accumulators, operands in registers.    hipcc --offload-arch=gfx950 -O3 -w mfma_f64_rate.hip -o build/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k64(double* out, int iters, double a, double b) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
  double av = a + threadIdx.x, bv = b - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
  v4f acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = v4f{0, 0, 0, 0};
  float av = a + threadIdx.x, bv = b - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
static float run(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  double* o; hipMalloc(&o, 1 << 24);
  const int iters = 20000, wgs = 256;
  const double ghz = 2.4;
#define GO(KN, NACC, T, FL)                                                                                              \
  {                                                                                                                      \
    float ms = run([&] { hipLaunchKernelGGL((KN<NACC>), dim3(wgs), dim3(256), 0, 0, (T*)o, iters, (T)1.0, (T)2.0); });     \
    double n = (double)iters * NACC;                                                                                     \
    printf(#KN " NACC=%d: %.3f ms, %.1f cycles per instruction and SIMD at %.1f GHz, %.1f TFLOP/s chip\n", NACC, ms,   \
           ms * 1e-3 * ghz * 1e9 / n, ghz, n * FL * wgs * 4 / (ms * 1e-3) / 1e12);                                       \
  }
  GO(k64, 1, double, 2048.0) GO(k64, 2, double, 2048.0) GO(k64, 4, double, 2048.0)
  GO(k32, 1, float, 2048.0) GO(k32, 4, float, 2048.0)
  return 0;
}
