// Dev tool: accuracy of a 32x32xK product computed (a) with v_mfma_f32_32x32x2_f32 and (b) as six
// v_mfma_f32_32x32x16_bf16 passes over a 3-way bf16 split of both operands, both vs fp64.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split3(float a, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)a;
  float r = a - (float)h;
  m = (__bf16)r;
  float r2 = r - (float)m;
  l = (__bf16)r2;
}

// A [32][K] row-major, B [K][32] row-major, one wave
__global__ void k_f32(const float* A, const float* B, float* C, int K) {
  int l = threadIdx.x, r = l & 31, hi = l >> 5;
  f32x16 acc;
  for (int t = 0; t < 16; ++t) acc[t] = 0.f;
  for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + hi], B[(k + hi) * 32 + r], acc, 0, 0, 0);
  for (int t = 0; t < 16; ++t) C[((t & 3) + 8 * (t >> 2) + 4 * hi) * 32 + r] = acc[t];
}
__global__ void k_bf16x3(const float* A, const float* B, float* C, int K, int npass) {
  int l = threadIdx.x, r = l & 31, hi = l >> 5;
  f32x16 acc;
  for (int t = 0; t < 16; ++t) acc[t] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    bf16x8 a1, a2, a3, b1, b2, b3;
    for (int j = 0; j < 8; ++j) {
      __bf16 h, m, lo;
      split3(A[r * K + k0 + 8 * hi + j], h, m, lo); a1[j] = h; a2[j] = m; a3[j] = lo;
      split3(B[(k0 + 8 * hi + j) * 32 + r], h, m, lo); b1[j] = h; b2[j] = m; b3[j] = lo;
    }
    // smallest terms first
    if (npass >= 6) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
    }
    if (npass >= 3) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
  }
  for (int t = 0; t < 16; ++t) C[((t & 3) + 8 * (t >> 2) + 4 * hi) * 32 + r] = acc[t];
}
int main() {
  for (int K : {128, 2048, 16384}) {
    std::vector<float> A(32 * K), B(K * 32);
    srand(1);
    const bool positive = getenv("PROBE_POSITIVE") != nullptr;
    for (auto& v : A) v = positive ? 0.25f + rand() / (float)RAND_MAX : (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    for (auto& v : B) v = positive ? (getenv("PROBE_NEGATIVE") ? -1.f : 1.f) * (0.25f + rand() / (float)RAND_MAX) : (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    std::vector<double> ref(1024, 0.0), absref(1024, 0.0);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < K; ++k) {
      ref[i * 32 + j] += (double)A[i * K + k] * B[k * 32 + j];
      absref[i * 32 + j] += fabs((double)A[i * K + k] * B[k * 32 + j]);
    }
    float *dA, *dB, *dC;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dC, 4096);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> C(1024);
    auto report = [&](const char* name) {
      (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
      double maxabs = 0, maxerr = 0, maxrel_sumabs = 0, meanrel = 0;
      for (int i = 0; i < 1024; ++i) {
        maxabs = fmax(maxabs, fabs(ref[i])); maxerr = fmax(maxerr, fabs(C[i] - ref[i]));
        maxrel_sumabs = fmax(maxrel_sumabs, fabs(C[i] - ref[i]) / absref[i]);
        meanrel += (C[i] - ref[i]) / absref[i] / 1024.0;
      }
      printf("K=%6d %-22s max-norm rel err %.3e   err / sum|a*b| %.3e   MEAN signed err / sum|a*b| %+.3e\n", K, name,
             maxerr / maxabs, maxrel_sumabs, meanrel);
    };
    k_f32<<<1, 64>>>(dA, dB, dC, K); (void)hipDeviceSynchronize(); report("fp32 mfma 32x32x2");
    for (int np : {1, 3, 6}) {
      k_bf16x3<<<1, 64>>>(dA, dB, dC, K, np); (void)hipDeviceSynchronize();
      char nm[64]; snprintf(nm, 64, "bf16 split, %d passes", np); report(nm);
    }
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
  }
  return 0;
}
