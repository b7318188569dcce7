// Fused multi-tensor optimiser steps and the robust losses: the per-step work after the hot path.
//   * AdamW   -- the reference's default optimiser: torch.optim.AdamW(lr, weight_decay) at
//                CGAT/lightning_module.py:328-331 (decoupled decay, bias-corrected moments, eps outside the sqrt)
//   * LAMB    -- CGAT/lambs.py:155-181 lamb_kernel as driven by JITLamb.step (226-262): no bias correction,
//                weight norm clamped to [0, 10], trust ratio = |w| / (|adam_step| + eps) with zero guards
//   * RobustL1 / RobustL2 -- CGAT/utils.py:30-47 (Lorentzian / Gaussian aleatoric losses), value and gradients
// One launch covers every parameter tensor: the host uploads a table of (param, grad, exp_avg, exp_avg_sq, n) and a
// list of (tensor, offset) chunks; a workgroup owns one chunk.  HBM-bound: 16 B read + 12 B written per parameter.
// LAMB needs two norms per tensor before the update: phase 1 updates the moments and writes per-chunk partial sums,
// a per-tensor reduction in chunk order (fixed -> deterministic) forms the trust ratio, phase 2 applies it.
#include "../../include/cgat_hip.h"
#include "common.h"
#include "kernels.h"

#define MT_CHUNK 16384   // elements per workgroup

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  const float t = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(256) void adamw_mt_kernel(const cgat_mt_tensor* __restrict__ tab,
                                                       const int32_t* __restrict__ ch_tensor,
                                                       const int64_t* __restrict__ ch_off, float lr, float beta1,
                                                       float beta2, float eps, float wd, float bc1, float bc2_sqrt) {
  const cgat_mt_tensor t = tab[ch_tensor[blockIdx.x]];
  const int64_t o = ch_off[blockIdx.x];
  const int64_t end = o + MT_CHUNK < t.n ? o + MT_CHUNK : t.n;
  const float step_size = lr / bc1;
  for (int64_t i = o + threadIdx.x; i < end; i += 256) {
    const float g = t.g[i];
    float p = t.p[i] * (1.f - lr * wd);                 // decoupled weight decay
    float m = t.m[i];
    m = m + (g - m) * (1.f - beta1);                    // exp_avg.lerp_(grad, 1 - beta1)
    const float v = t.v[i] * beta2 + (1.f - beta2) * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p -= step_size * (m / denom);
    t.p[i] = p; t.m[i] = m; t.v[i] = v;
  }
}

// phase 1: moments, partial sums of |param|^2 and |adam_step|^2 per chunk
__global__ __launch_bounds__(256) void lamb_phase1_kernel(const cgat_mt_tensor* __restrict__ tab,
                                                          const int32_t* __restrict__ ch_tensor,
                                                          const int64_t* __restrict__ ch_off, float beta1, float beta2,
                                                          float eps, float wd, float* __restrict__ partial) {
  __shared__ float sh[4];
  const cgat_mt_tensor t = tab[ch_tensor[blockIdx.x]];
  const int64_t o = ch_off[blockIdx.x];
  const int64_t end = o + MT_CHUNK < t.n ? o + MT_CHUNK : t.n;
  float wn = 0.f, an = 0.f;
  for (int64_t i = o + threadIdx.x; i < end; i += 256) {
    const float g = t.g[i], p = t.p[i];
    const float m = t.m[i] * beta1 + (1.f - beta1) * g;
    const float v = t.v[i] * beta2 + (1.f - beta2) * (g * g);
    t.m[i] = m; t.v[i] = v;
    const float s = m / (sqrtf(v) + eps) + wd * p;
    wn += p * p;
    an += s * s;
  }
  wn = block_sum_256(wn, sh);
  an = block_sum_256(an, sh);
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = wn; partial[2 * blockIdx.x + 1] = an; }
}

// one thread per tensor: chunks of a tensor are consecutive in the chunk list
__global__ void lamb_ratio_kernel(const int32_t* __restrict__ first_chunk, int n_tensors,
                                  const float* __restrict__ partial, float eps, float* __restrict__ ratio) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tensors) return;
  float wn = 0.f, an = 0.f;
  for (int c = first_chunk[t]; c < first_chunk[t + 1]; ++c) { wn += partial[2 * c]; an += partial[2 * c + 1]; }
  float weight_norm = fminf(fmaxf(sqrtf(wn), 0.f), 10.f);
  const float adam_norm = sqrtf(an);
  float r = weight_norm / (adam_norm + eps);
  if (weight_norm == 0.f) r = 1.f;
  if (adam_norm == 0.f) r = 1.f;
  ratio[t] = r;
}

__global__ __launch_bounds__(256) void lamb_phase2_kernel(const cgat_mt_tensor* __restrict__ tab,
                                                          const int32_t* __restrict__ ch_tensor,
                                                          const int64_t* __restrict__ ch_off, float lr, float eps,
                                                          float wd, const float* __restrict__ ratio) {
  const int ti = ch_tensor[blockIdx.x];
  const cgat_mt_tensor t = tab[ti];
  const int64_t o = ch_off[blockIdx.x];
  const int64_t end = o + MT_CHUNK < t.n ? o + MT_CHUNK : t.n;
  const float scale = lr * ratio[ti];
  for (int64_t i = o + threadIdx.x; i < end; i += 256) {
    const float p = t.p[i];
    const float s = t.m[i] / (sqrtf(t.v[i]) + eps) + wd * p;
    t.p[i] = p - scale * s;
  }
}

extern "C" int32_t cgat_mt_chunk_elems(void) { return MT_CHUNK; }

extern "C" int cgat_adamw_step(const cgat_mt_tensor* table, const int32_t* chunk_tensor, const int64_t* chunk_off,
                               int32_t n_chunks, float lr, float beta1, float beta2, float eps, float weight_decay,
                               int64_t step, void* stream) {
  CGAT_CHECK_ARG(n_chunks >= 0 && step >= 1, "adamw_step: bad arguments");
  if (n_chunks == 0) return CGAT_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  CGAT_PROF("adamw", (hipStream_t)stream);
  hipLaunchKernelGGL(adamw_mt_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table, chunk_tensor, chunk_off,
                     lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2));
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

extern "C" int cgat_lamb_step(const cgat_mt_tensor* table, const int32_t* chunk_tensor, const int64_t* chunk_off,
                              int32_t n_chunks, const int32_t* first_chunk, int32_t n_tensors, float lr, float beta1,
                              float beta2, float eps, float weight_decay, float* ws /* 2*n_chunks + n_tensors floats */,
                              void* stream) {
  CGAT_CHECK_ARG(n_chunks >= 0 && n_tensors >= 0, "lamb_step: bad arguments");
  if (n_chunks == 0) return CGAT_OK;
  float* partial = ws;
  float* ratio = ws + 2 * (size_t)n_chunks;
  hipStream_t s = (hipStream_t)stream;
  CGAT_PROF("lamb", s);
  hipLaunchKernelGGL(lamb_phase1_kernel, dim3(n_chunks), dim3(256), 0, s, table, chunk_tensor, chunk_off, beta1, beta2, eps,
                     weight_decay, partial);
  CGAT_LAUNCH_CHECK();
  hipLaunchKernelGGL(lamb_ratio_kernel, dim3(cdiv(n_tensors, 64)), dim3(64), 0, s, first_chunk, n_tensors, partial, eps,
                     ratio);
  CGAT_LAUNCH_CHECK();
  hipLaunchKernelGGL(lamb_phase2_kernel, dim3(n_chunks), dim3(256), 0, s, table, chunk_tensor, chunk_off, lr, eps,
                     weight_decay, ratio);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// ---- robust losses: per-row loss terms and gradients (the mean and its 1/n factor are applied by the caller) ----
// kind 1: sqrt(2) |o - t| exp(-s) + s       kind 2: 0.5 (o - t)^2 exp(-2 s) + s
__global__ void robust_loss_kernel(const float* __restrict__ o, const float* __restrict__ s, const float* __restrict__ t,
                                   int n, int kind, float* __restrict__ loss, float* __restrict__ go,
                                   float* __restrict__ gs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float d = o[i] - t[i], ls = s[i];
  if (kind == 1) {
    const float e = expf(-ls), a = 1.41421356237309515f * fabsf(d) * e;   // np.sqrt(2.0) rounded to fp32 by the product
    loss[i] = a + ls;
    go[i] = 1.41421356237309515f * e * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    gs[i] = 1.f - a;
  } else {
    const float e = expf(-2.f * ls), a = 0.5f * d * d * e;
    loss[i] = a + ls;
    go[i] = d * e;
    gs[i] = 1.f - 2.f * a;
  }
}

extern "C" int cgat_robust_loss(const float* output, const float* log_std, const float* target, int32_t n, int32_t kind,
                                float* loss_terms, float* g_output, float* g_log_std, void* stream) {
  CGAT_CHECK_ARG(n >= 0 && (kind == 1 || kind == 2), "robust_loss: kind must be 1 (L1) or 2 (L2)");
  if (n == 0) return CGAT_OK;
  hipLaunchKernelGGL(robust_loss_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, output, log_std, target,
                     n, kind, loss_terms, g_output, g_log_std);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}
