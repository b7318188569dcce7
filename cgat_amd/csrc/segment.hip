// Segment kernels over rows that are already grouped by segment (CSR rowptr): the
// softmax-over-incoming-edges of the attention (third-party torch_geometric.utils.softmax at
// reference CGAT.py:323,59: exp(a - segmax) / (segsum + 1e-16)), Roost's weighted variant
// (roost_message.py:307-311: (w**pow) * exp(a - segmax) / (segsum + 1e-13)), weighted segment
// sums (scatter_add, CGAT.py:60 / PyG aggregate) and per-row per-head dot products.
// No atomics anywhere: every output element has exactly one writer and a fixed summation
// order, so results are bitwise reproducible.
#include "common.h"
#include "kernels.h"
#include "mfma_bf16.h"

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float act_f(float v, int act) {
  switch (act) {
    case CGAT_ACT_TANH: return tanhf(v);
    case CGAT_ACT_LEAKY: return v > 0.f ? v : 0.01f * v;
    case CGAT_ACT_RELU: return v > 0.f ? v : 0.f;
    default: return v;
  }
}

// Segments longer than SEG_LONG rows (a hub atom: 20 000 incoming edges in the test) leave the thread-per-output /
// workgroup-per-segment kernels below and are handled by the *_long kernels: all threads of a workgroup stride over
// the rows of ONE segment and combine through wavefront (DPP / shuffle) and LDS reductions in a fixed order, so the
// results stay bitwise reproducible.  Which segments are long is found on the device (every workgroup of a long
// kernel scans the row pointers of its share of segments and collects the long ones in LDS): no host round trip, no
// workspace, nothing to capture around.
#define SEG_LONG 256

__device__ __forceinline__ float wave_max64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sum / max over the 256 threads of a workgroup, result in every thread; `red` = 4 floats of LDS
__device__ __forceinline__ float block256_sum(float v, float* red) {
  v = wave_sum64(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block256_max(float v, float* red) {
  v = wave_max64(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
// the long segments among segments [256 b, 256 b + 256): ids into list[] (LDS), count returned; order irrelevant (each
// is processed independently)
__device__ __forceinline__ int collect_long256(const int* __restrict__ rowptr, int S, int* list, int* count, int b) {
  if (threadIdx.x == 0) *count = 0;
  __syncthreads();
  const int s = b * 256 + threadIdx.x;
  if (s < S && rowptr[s + 1] - rowptr[s] > SEG_LONG) list[atomicAdd(count, 1)] = s;
  __syncthreads();
  return *count;
}

// one thread per (segment, feature).  The workgroups behind the first `main_blocks` handle the long segments
// (seg_softmax_fwd_long below): ONE launch for both (a separate launch that finds nothing to do is what the normal case
// paid before, and at the harness' 64-crystal batch the step is launch-bound)
__device__ void seg_softmax_fwd_long(const float* __restrict__ a, const float* __restrict__ mult,
                                     const int* __restrict__ rowptr, int S, int F, float eps, float* __restrict__ alpha,
                                     float* __restrict__ ssum, int b);
__global__ __launch_bounds__(256) void seg_softmax_fwd_kernel(const float* __restrict__ a, const float* __restrict__ mult,
                                       const int* __restrict__ rowptr, int S, int F, float eps,
                                       float* __restrict__ alpha, float* __restrict__ ssum, int main_blocks) {
  if ((int)blockIdx.x >= main_blocks) return seg_softmax_fwd_long(a, mult, rowptr, S, F, eps, alpha, ssum, blockIdx.x - main_blocks);
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)S * F) return;
  int s = (int)(i / F), f = (int)(i % F);
  int r0 = rowptr[s], r1 = rowptr[s + 1];
  if (r1 - r0 > SEG_LONG) return;                 // seg_softmax_fwd_long
  float mx = -INFINITY;
  for (int r = r0; r < r1; ++r) mx = fmaxf(mx, a[(long)r * F + f]);
  float z = 0.f;
  for (int r = r0; r < r1; ++r) {
    float e = expf(a[(long)r * F + f] - mx);
    if (mult) e *= mult[r];
    z += e;
  }
  const float den = z + eps;                      // alpha = e / (sum + eps): one correctly rounded division, as the
  float tot = 0.f;                                // reference forms it (see seg_attnpool_fwd_kernel)
  for (int r = r0; r < r1; ++r) {
    float e = expf(a[(long)r * F + f] - mx);
    if (mult) e *= mult[r];
    float al = e / den;
    alpha[(long)r * F + f] = al;
    tot += al;
  }
  if (ssum) ssum[i] = tot;
}

__device__ void seg_softmax_fwd_long(const float* __restrict__ a, const float* __restrict__ mult,
                                     const int* __restrict__ rowptr, int S, int F, float eps, float* __restrict__ alpha,
                                     float* __restrict__ ssum, int b) {
  __shared__ int list[256];
  __shared__ int count;
  __shared__ float red[4];
  const int n = collect_long256(rowptr, S, list, &count, b);
  for (int k = 0; k < n; ++k) {
    const int s = list[k], r0 = rowptr[s], r1 = rowptr[s + 1];
    for (int f = 0; f < F; ++f) {
      float mx = -INFINITY;
      for (int r = r0 + threadIdx.x; r < r1; r += 256) mx = fmaxf(mx, a[(long)r * F + f]);
      mx = block256_max(mx, red);
      float z = 0.f;
      for (int r = r0 + threadIdx.x; r < r1; r += 256) {
        float e = expf(a[(long)r * F + f] - mx);
        if (mult) e *= mult[r];
        z += e;
      }
      z = block256_sum(z, red);
      const float den = z + eps;
      float tot = 0.f;
      for (int r = r0 + threadIdx.x; r < r1; r += 256) {
        float e = expf(a[(long)r * F + f] - mx);
        if (mult) e *= mult[r];
        const float al = e / den;
        alpha[(long)r * F + f] = al;
        tot += al;
      }
      if (ssum) {                                  // (uniform: kernel argument)
        tot = block256_sum(tot, red);
        if (threadIdx.x == 0) ssum[(long)s * F + f] = tot;
      }
    }
  }
}

// ga = alpha * (g - sum_seg alpha*g),  g = galpha + gssum[seg];   gmult = ga / mult  (F == 1 only)
__device__ void seg_softmax_bwd_long(const float* __restrict__ alpha, const float* __restrict__ galpha,
                                     const float* __restrict__ gssum, const float* __restrict__ mult,
                                     const int* __restrict__ rowptr, int S, int F, float* __restrict__ ga,
                                     float* __restrict__ gmult, int b);
__global__ __launch_bounds__(256) void seg_softmax_bwd_kernel(const float* __restrict__ alpha, const float* __restrict__ galpha,
                                       const float* __restrict__ gssum, const float* __restrict__ mult,
                                       const int* __restrict__ rowptr, int S, int F, float* __restrict__ ga,
                                       float* __restrict__ gmult, int main_blocks) {
  if ((int)blockIdx.x >= main_blocks)   // the long segments: trailing workgroups of the same launch (see the forward)
    return seg_softmax_bwd_long(alpha, galpha, gssum, mult, rowptr, S, F, ga, gmult, blockIdx.x - main_blocks);
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)S * F) return;
  int s = (int)(i / F), f = (int)(i % F);
  int r0 = rowptr[s], r1 = rowptr[s + 1];
  if (r1 - r0 > SEG_LONG) return;                 // seg_softmax_bwd_long
  float gs = gssum ? gssum[i] : 0.f;
  float dot = 0.f;
  for (int r = r0; r < r1; ++r) dot += alpha[(long)r * F + f] * (galpha[(long)r * F + f] + gs);
  for (int r = r0; r < r1; ++r) {
    float al = alpha[(long)r * F + f];
    float g = al * (galpha[(long)r * F + f] + gs - dot);
    ga[(long)r * F + f] = g;
    if (gmult) gmult[r] = (mult[r] != 0.f) ? g / mult[r] : 0.f;
  }
}

__device__ void seg_softmax_bwd_long(const float* __restrict__ alpha, const float* __restrict__ galpha,
                                     const float* __restrict__ gssum, const float* __restrict__ mult,
                                     const int* __restrict__ rowptr, int S, int F, float* __restrict__ ga,
                                     float* __restrict__ gmult, int b) {
  __shared__ int list[256];
  __shared__ int count;
  __shared__ float red[4];
  const int n = collect_long256(rowptr, S, list, &count, b);
  for (int k = 0; k < n; ++k) {
    const int s = list[k], r0 = rowptr[s], r1 = rowptr[s + 1];
    for (int f = 0; f < F; ++f) {
      const float gs = gssum ? gssum[(long)s * F + f] : 0.f;
      float dot = 0.f;
      for (int r = r0 + threadIdx.x; r < r1; r += 256) dot += alpha[(long)r * F + f] * (galpha[(long)r * F + f] + gs);
      dot = block256_sum(dot, red);
      for (int r = r0 + threadIdx.x; r < r1; r += 256) {
        const float al = alpha[(long)r * F + f];
        const float g = al * (galpha[(long)r * F + f] + gs - dot);
        ga[(long)r * F + f] = g;
        if (gmult) gmult[r] = (mult[r] != 0.f) ? g / mult[r] : 0.f;
      }
    }
  }
}

int seg_softmax_fwd_launch(const float* a, const float* mult, const int* rowptr, int S, int F, float eps, float* alpha,
                           float* ssum, hipStream_t s) {
  long n = (long)S * F;
  if (n <= 0) return CGAT_OK;
  CGAT_PROF("seg_softmax", s);
  const int main_blocks = (int)cdiv(n, 256);
  hipLaunchKernelGGL(seg_softmax_fwd_kernel, dim3(main_blocks + cdiv(S, 256)), dim3(256), 0, s, a, mult, rowptr, S, F, eps,
                     alpha, ssum, main_blocks);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

int seg_softmax_bwd_launch(const float* alpha, const float* galpha, const float* gssum, const float* mult,
                           const int* rowptr, int S, int F, float* ga, float* gmult, hipStream_t s) {
  long n = (long)S * F;
  if (n <= 0) return CGAT_OK;
  CGAT_CHECK_ARG(!gmult || F == 1, "seg_softmax_bwd: gradient of the multiplier needs F == 1");
  const int main_blocks = (int)cdiv(n, 256);
  hipLaunchKernelGGL(seg_softmax_bwd_kernel, dim3(main_blocks + cdiv(S, 256)), dim3(256), 0, s, alpha, galpha, gssum, mult,
                     rowptr, S, F, ga, gmult, main_blocks);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// one workgroup per segment, threads stride the feature dimension, rows in CSR order
__global__ void seg_wsum_kernel(const float* __restrict__ x, long ldx, const int* __restrict__ ridx,
                                const float* __restrict__ w, int wF, int fw, const int* __restrict__ rowptr, int F,
                                int act, float* __restrict__ out, long ldo, long xblock) {
  int s = blockIdx.x;
  int r0 = rowptr[s], r1 = rowptr[s + 1];
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    float acc = 0.f;
    int wf = w ? f / fw : 0;
    for (int r = r0; r < r1; ++r) {
      long row = ridx ? (long)ridx[r] : (long)r;
      float v = act_f(xblock ? x[(long)(f >> 7) * xblock + row * 128 + (f & 127)] : x[row * ldx + f], act);
      if (w) v *= w[(long)r * wF + wf];
      acc += v;
    }
    out[(long)s * ldo + f] = acc;
  }
}

// Same sum, four consecutive features per thread (16-byte loads) and the rows of a segment fetched four at a time
// before they are added in CSR order: the loads of a batch are independent, so a 12-row segment costs three memory
// round trips instead of twelve.  Needs F % 4 == 0, fw % 4 == 0, 16-byte aligned rows.
template <int SEG_U>
__global__ __launch_bounds__(256) void seg_wsum_vec_kernel(const float* __restrict__ x, long ldx,
                                                           const int* __restrict__ ridx, const float* __restrict__ w,
                                                           int wF, int fw, const int* __restrict__ rowptr, int F, int act,
                                                           float* __restrict__ out, long ldo, long xblock) {
  const int s = blockIdx.x;
  const int r0 = rowptr[s], r1 = rowptr[s + 1];
  if (r1 - r0 > SEG_LONG) return;                 // seg_wsum_long_kernel
  for (int f = 4 * threadIdx.x; f < F; f += 4 * blockDim.x) {
    const int wf = w ? f / fw : 0;
    const float* xb = xblock ? x + (long)(f >> 7) * xblock + (f & 127) : x + f;
    const long pitch = xblock ? 128 : ldx;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // SEG_U rows per batch: a workgroup's time is (number of batches) x (memory round trip), not bytes.  For rows in
    // CSR order (the forward's weighted sum of messages) 12 rows per trip took the 3 GB pass from 0.98 to 0.68 ms;
    // for gathered rows (the source-side sum of gZ) 12 in flight per thread is slower than 4 (1.30 vs 0.98 ms)
    for (int r = r0; r < r1; r += SEG_U) {
      float4 v[SEG_U];
      float wv[SEG_U];
#pragma unroll
      for (int u = 0; u < SEG_U; ++u) {
        const int rr = r + u < r1 ? r + u : r1 - 1;
        const long row = ridx ? (long)ridx[rr] : (long)rr;
        v[u] = *reinterpret_cast<const float4*>(xb + row * pitch);
        wv[u] = w ? w[(long)rr * wF + wf] : 1.f;
      }
#pragma unroll
      for (int u = 0; u < SEG_U; ++u) {
        if (r + u < r1) {
          acc.x += act_f(v[u].x, act) * wv[u];
          acc.y += act_f(v[u].y, act) * wv[u];
          acc.z += act_f(v[u].z, act) * wv[u];
          acc.w += act_f(v[u].w, act) * wv[u];
        }
      }
    }
    *reinterpret_cast<float4*>(out + (long)s * ldo + f) = acc;
  }
}

// the same with x stored as bf16 (the "bf16" edge-storage mode: ldx / xblock count bf16 elements)
template <int SEG_U>
__global__ __launch_bounds__(256) void seg_wsum_vec_bf16_kernel(const __bf16* __restrict__ x, long ldx,
                                                                const int* __restrict__ ridx, const float* __restrict__ w,
                                                                int wF, int fw, const int* __restrict__ rowptr, int F,
                                                                int act, float* __restrict__ out, long ldo, long xblock) {
  const int s = blockIdx.x;
  const int r0 = rowptr[s], r1 = rowptr[s + 1];
  if (r1 - r0 > SEG_LONG) return;                 // seg_wsum_long_kernel
  for (int f = 4 * threadIdx.x; f < F; f += 4 * blockDim.x) {
    const int wf = w ? f / fw : 0;
    const __bf16* xb = xblock ? x + (long)(f >> 7) * xblock + (f & 127) : x + f;
    const long pitch = xblock ? 128 : ldx;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = r0; r < r1; r += SEG_U) {
      float4 v[SEG_U];
      float wv[SEG_U];
#pragma unroll
      for (int u = 0; u < SEG_U; ++u) {
        const int rr = r + u < r1 ? r + u : r1 - 1;
        const long row = ridx ? (long)ridx[rr] : (long)rr;
        v[u] = load4_bf16(xb + row * pitch);
        wv[u] = w ? w[(long)rr * wF + wf] : 1.f;
      }
#pragma unroll
      for (int u = 0; u < SEG_U; ++u) {
        if (r + u < r1) {
          acc.x += act_f(v[u].x, act) * wv[u];
          acc.y += act_f(v[u].y, act) * wv[u];
          acc.z += act_f(v[u].z, act) * wv[u];
          acc.w += act_f(v[u].w, act) * wv[u];
        }
      }
    }
    *reinterpret_cast<float4*>(out + (long)s * ldo + f) = acc;
  }
}

// Long segments of the vector forms: 1024 threads = G row groups x (up to 256) column quads; group g sums rows
// r0 + g, r0 + g + G, ... (four in flight), the groups are added through LDS in group order.
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const __bf16* p) { return load4_bf16(p); }
template <typename T>
__global__ __launch_bounds__(1024) void seg_wsum_long_kernel(const T* __restrict__ x, long ldx, const int* __restrict__ ridx,
                                                             const float* __restrict__ w, int wF, int fw,
                                                             const int* __restrict__ rowptr, int S, int F, int act,
                                                             float* __restrict__ out, long ldo, long xblock) {
  __shared__ int list[1024];
  __shared__ int count;
  __shared__ float4 part[1024];
  if (threadIdx.x == 0) count = 0;
  __syncthreads();
  {
    const int s = blockIdx.x * 1024 + threadIdx.x;
    if (s < S && rowptr[s + 1] - rowptr[s] > SEG_LONG) list[atomicAdd(&count, 1)] = s;
  }
  __syncthreads();
  const int n = count;
  const int quads = F / 4;
  const int tpr = quads < 256 ? quads : 256;      // threads per row (column quads handled at a time)
  const int G = 1024 / tpr;                        // row groups
  const int g = threadIdx.x / tpr, q = threadIdx.x - g * tpr;
  for (int k = 0; k < n; ++k) {
    const int s = list[k], r0 = rowptr[s], r1 = rowptr[s + 1];
    for (int q0 = 0; q0 < quads; q0 += tpr) {      // (uniform trip count)
      const int f = 4 * (q0 + q);
      const bool live = g < G && q0 + q < quads;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live) {
        const int wf = w ? f / fw : 0;
        const T* xb = xblock ? x + (long)(f >> 7) * xblock + (f & 127) : x + f;
        const long pitch = xblock ? 128 : ldx;
        for (int r = r0 + g; r < r1; r += 4 * G) {
          float4 v[4];
          float wv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int rr = r + u * G < r1 ? r + u * G : r1 - 1;
            const long row = ridx ? (long)ridx[rr] : (long)rr;
            v[u] = ld4(xb + row * pitch);
            wv[u] = w ? w[(long)rr * wF + wf] : 1.f;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (r + u * G < r1) {
              acc.x += act_f(v[u].x, act) * wv[u];
              acc.y += act_f(v[u].y, act) * wv[u];
              acc.z += act_f(v[u].z, act) * wv[u];
              acc.w += act_f(v[u].w, act) * wv[u];
            }
        }
      }
      __syncthreads();
      part[threadIdx.x] = acc;
      __syncthreads();
      if (g == 0 && q0 + q < quads) {
        float4 t = part[q];
        for (int gg = 1; gg < G; ++gg) {
          const float4 o = part[gg * tpr + q];
          t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
        }
        *reinterpret_cast<float4*>(out + (long)s * ldo + f) = t;
      }
    }
  }
}

int seg_wsum_launch(const float* x, long ldx, const int* ridx, const float* w, int wF, int fw, const int* rowptr, int S,
                    int F, int act, float* out, long ldo, hipStream_t s, long xblock, int x_bf16) {
  if (S <= 0 || F <= 0) return CGAT_OK;
  CGAT_PROF("seg_wsum", s);
  if (x_bf16) {
    CGAT_CHECK_ARG((F % 4) == 0 && (!w || (fw % 4) == 0) && (ldx % 4) == 0 && (ldo % 4) == 0 && (xblock % 4) == 0 &&
                   (((uintptr_t)x) & 7) == 0 && (((uintptr_t)out) & 15) == 0, "seg_wsum: bf16 input needs the vector shape");
    int threads = F >= 1024 ? 256 : (F >= 512 ? 192 : 64);
    if (F / 4 < threads) threads = ((F / 4 + 63) / 64) * 64;
    const __bf16* xb = reinterpret_cast<const __bf16*>(x);
    if (ridx)
      hipLaunchKernelGGL(seg_wsum_vec_bf16_kernel<4>, dim3(S), dim3(threads), 0, s, xb, ldx, ridx, w, wF, fw, rowptr, F, act,
                         out, ldo, xblock);
    else
      hipLaunchKernelGGL(seg_wsum_vec_bf16_kernel<12>, dim3(S), dim3(threads), 0, s, xb, ldx, ridx, w, wF, fw, rowptr, F, act,
                         out, ldo, xblock);
    CGAT_LAUNCH_CHECK();
    hipLaunchKernelGGL(seg_wsum_long_kernel<__bf16>, dim3(cdiv(S, 1024)), dim3(1024), 0, s, xb, ldx, ridx, w, wF, fw, rowptr,
                       S, F, act, out, ldo, xblock);
    CGAT_LAUNCH_CHECK();
    return CGAT_OK;
  }
  const bool vec = (F % 4) == 0 && (!w || (fw % 4) == 0) && (ldx % 4) == 0 && (ldo % 4) == 0 && (xblock % 4) == 0 &&
                   ((((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0;
  if (vec) {
    int threads = F >= 1024 ? 256 : (F >= 512 ? 192 : (F >= 256 ? 64 : 64));
    if (F / 4 < threads) threads = ((F / 4 + 63) / 64) * 64;
    if (ridx)
      hipLaunchKernelGGL(seg_wsum_vec_kernel<4>, dim3(S), dim3(threads), 0, s, x, ldx, ridx, w, wF, fw, rowptr, F, act, out,
                         ldo, xblock);
    else
      hipLaunchKernelGGL(seg_wsum_vec_kernel<12>, dim3(S), dim3(threads), 0, s, x, ldx, ridx, w, wF, fw, rowptr, F, act, out,
                         ldo, xblock);
    CGAT_LAUNCH_CHECK();
    hipLaunchKernelGGL(seg_wsum_long_kernel<float>, dim3(cdiv(S, 1024)), dim3(1024), 0, s, x, ldx, ridx, w, wF, fw, rowptr, S,
                       F, act, out, ldo, xblock);
    CGAT_LAUNCH_CHECK();
    return CGAT_OK;
  }
  int threads = F >= 256 ? 256 : (F >= 128 ? 128 : 64);
  hipLaunchKernelGGL(seg_wsum_kernel, dim3(S), dim3(threads), 0, s, x, ldx, ridx, w, wF, fw, rowptr, F, act, out, ldo,
                     xblock);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// one wave per row: out[r,h] = sum_j act(x[r,h*Hd+j]) * v[vrow(r)*ldv + h*Hd + j] + bias[h] + addv[vrow(r)*H + h]
// VEC: 16-byte loads (Hd % 4 == 0, 16-byte aligned rows)
template <bool VEC>
__global__ void rowdot_kernel(const float* __restrict__ x, long ldx, int act, const float* __restrict__ v, long ldv,
                              const int* __restrict__ vrow, const float* __restrict__ bias,
                              const float* __restrict__ addv, int rows, int H, int Hd, float* __restrict__ out) {
  int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= rows) return;
  long vr = vrow ? (long)vrow[row] : 0;
  const float* xr = x + (long)row * ldx;
  const float* vv = v + vr * ldv;
  for (int h = 0; h < H; ++h) {
    float s = 0.f;
    if (VEC) {
      const float4* x4 = reinterpret_cast<const float4*>(xr + h * Hd);
      const float4* v4 = reinterpret_cast<const float4*>(vv + h * Hd);
      for (int j = lane; j < Hd / 4; j += 64) {
        float4 a = x4[j], b = v4[j];
        s += act_f(a.x, act) * b.x + act_f(a.y, act) * b.y + act_f(a.z, act) * b.z + act_f(a.w, act) * b.w;
      }
    } else {
      for (int j = lane; j < Hd; j += 64) s += act_f(xr[h * Hd + j], act) * vv[h * Hd + j];
    }
    s = wave_sum64(s);
    if (lane == 0) {
      if (bias) s += bias[h];
      if (addv) s += addv[vr * H + h];
      out[(long)row * H + h] = s;
    }
  }
}

int rowdot_launch(const float* x, long ldx, int act, const float* v, long ldv, const int* vrow, const float* bias,
                  const float* addv, int rows, int H, int Hd, float* out, hipStream_t s) {
  if (rows <= 0) return CGAT_OK;
  const bool vec = (Hd % 4 == 0) && (ldx % 4 == 0) && (ldv % 4 == 0) && ((((uintptr_t)x) | ((uintptr_t)v)) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(rowdot_kernel<true>, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ldx, act, v, ldv, vrow, bias, addv,
                       rows, H, Hd, out);
  else
    hipLaunchKernelGGL(rowdot_kernel<false>, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ldx, act, v, ldv, vrow, bias, addv,
                       rows, H, Hd, out);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// ---------------------------------------------------------------------------------------
// Softmax-weighted segment sum in ONE pass per direction ("attention pooling"):
//     out[s, f] = sum_{r in seg s} alpha[r, f / fw] * m[r, f],
//     alpha[r, c] = mult[r] * exp(a[r, c] - max_seg a[., c]) / (sum_seg mult * exp(...) + eps)
// The three places of the reference that compute it as softmax -> multiply -> scatter_add over [rows, F] tensors:
//   * GATConvNodes with vector attention (CGAT.py:323-329: one logit per head AND channel, aF = F, fw = 1),
//   * MHAttention, the per-crystal pooling (CGAT.py:59-61: aF = heads, fw = C, or aF = F for global_vector_attention),
//   * Roost's WeightedAttention (roost_message.py:305-317: aF = 1, fw = C, mult = weights ** pow, eps = 1e-13).
// Forward keeps per (segment, logit column) the maximum and 1 / (sum + eps) instead of alpha[rows, aF]; backward
// recomputes alpha from them:  g_m = alpha * g_out,  g_a[r, c] = sum_{f in c} alpha * g_out * (m - out),
// g_mult[r] = e inv sum_f g_out (m - out) (aF == 1; also where mult[r] == 0).  One workgroup per segment, four consecutive features per thread, rows
// in CSR order (row r of the segment is ridx[r] of the operands when ridx is given, so callers whose rows are not
// grouped pass the plan's permutation instead of gathering), U of them in flight; no atomics, fixed summation order.
// ---------------------------------------------------------------------------------------
#define AP_U 4
#define AP_W 8     // rows whose logits are in flight together in the maximum / normaliser passes
template <bool PER_F>   // PER_F: one logit column per feature (fw == 1); else fw % 4 == 0
__global__ __launch_bounds__(1024) void seg_attnpool_fwd_kernel(const float* __restrict__ a, int aF, int fw,
                                                                const float* __restrict__ mult,
                                                                const float* __restrict__ m, long ldm,
                                                                const int* __restrict__ rowptr,
                                                                const int* __restrict__ ridx, int F, float eps,
                                                                float* __restrict__ out, float* __restrict__ mx_out,
                                                                float* __restrict__ inv_out, float* __restrict__ out_lo) {
  const int s = blockIdx.x;
  const int r0 = rowptr[s], r1 = rowptr[s + 1];
  for (int f = 4 * threadIdx.x; f < F; f += 4 * blockDim.x) {
    const int ac = PER_F ? f : f / fw;
    // (the logits of AP_W rows are requested together in the two passes below: one row per round trip made a segment of
    // 24 neighbours 48 dependent L2 latencies -- 77 us per launch at 1 280 atoms; same operations in the same order)
    float4 mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int r = r0; r < r1; r += AP_W) {
      float4 v[AP_W];
#pragma unroll
      for (int u = 0; u < AP_W; ++u) {
        const int rs = r + u < r1 ? r + u : r1 - 1;
        const long row = ridx ? (long)ridx[rs] : (long)rs;
        if (PER_F) v[u] = *reinterpret_cast<const float4*>(a + row * aF + ac);
        else v[u].x = a[row * aF + ac];
      }
#pragma unroll
      for (int u = 0; u < AP_W; ++u) {       // (a clamped repeat of the last row leaves a maximum unchanged)
        mx.x = fmaxf(mx.x, v[u].x);
        if (PER_F) { mx.y = fmaxf(mx.y, v[u].y); mx.z = fmaxf(mx.z, v[u].z); mx.w = fmaxf(mx.w, v[u].w); }
      }
    }
    if (!PER_F) mx.y = mx.z = mx.w = mx.x;
    // Normalised coefficients exactly as the reference forms them -- alpha = (mult e) / (sum + eps), ONE correctly rounded
    // division per coefficient (torch_geometric softmax: out / (out_sum + 1e-16); roost_message.py:311) -- not e times a
    // rounded reciprocal: a segment of one row then has alpha == 1 and two equal rows 0.5 each, bit for bit, and the
    // gradient of the logits, which vanishes or is antisymmetric there, comes out so (tools/roost_gate_probe.py: with
    // e * (1 / sum) a single-row segment had alpha = 1 - 2^-24 and a logit gradient of 2e-8 instead of 0, which the gate
    // network's weight gradient -- a sum that cancels over every segment -- amplified to 2e-4 of its value).
    // The weighted sum is accumulated in fp64 and handed to backward as a float pair (out, out_lo): backward centres every
    // row on it, g_a = sum_f alpha g (m - out), and a rounding error of `out` would be common to all rows of the segment.
    // The segment's denominator sum + eps goes to `inv_out` (backward divides by it again: the same alpha).
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = r0; r < r1; r += AP_W) {
      float4 v[AP_W];
      float wv[AP_W];
#pragma unroll
      for (int u = 0; u < AP_W; ++u) {
        const int rs = r + u < r1 ? r + u : r1 - 1;
        const long rr = ridx ? (long)ridx[rs] : (long)rs;
        wv[u] = mult ? mult[rr] : 1.f;
        if (PER_F) v[u] = *reinterpret_cast<const float4*>(a + rr * aF + ac);
        else v[u].x = a[rr * aF + ac];
      }
#pragma unroll
      for (int u = 0; u < AP_W; ++u) {
        if (r + u < r1) {
          if (PER_F) {
            z.x += expf(v[u].x - mx.x) * wv[u]; z.y += expf(v[u].y - mx.y) * wv[u];
            z.z += expf(v[u].z - mx.z) * wv[u]; z.w += expf(v[u].w - mx.w) * wv[u];
          } else {
            z.x += expf(v[u].x - mx.x) * wv[u];
          }
        }
      }
    }
    if (!PER_F) z.y = z.z = z.w = z.x;
    const float4 den = make_float4(z.x + eps, z.y + eps, z.z + eps, z.w + eps);
    double accd[4] = {0.0, 0.0, 0.0, 0.0};
    double asum[4] = {0.0, 0.0, 0.0, 0.0};     // the ROUNDED coefficients' own sum (1 up to a few 2^-24)
    for (int r = r0; r < r1; r += AP_U) {
      float4 av[AP_U], mv[AP_U];
      float wv[AP_U];
#pragma unroll
      for (int u = 0; u < AP_U; ++u) {
        const int rs = r + u < r1 ? r + u : r1 - 1;
        const long rr = ridx ? (long)ridx[rs] : (long)rs;
        if (PER_F) av[u] = *reinterpret_cast<const float4*>(a + rr * aF + ac);
        else av[u].x = a[rr * aF + ac];
        mv[u] = *reinterpret_cast<const float4*>(m + rr * ldm + f);
        wv[u] = mult ? mult[rr] : 1.f;
      }
#pragma unroll
      for (int u = 0; u < AP_U; ++u) {
        if (r + u < r1) {
          float4 al;
          al.x = (expf(av[u].x - mx.x) * wv[u]) / den.x;
          if (PER_F) {
            al.y = (expf(av[u].y - mx.y) * wv[u]) / den.y; al.z = (expf(av[u].z - mx.z) * wv[u]) / den.z;
            al.w = (expf(av[u].w - mx.w) * wv[u]) / den.w;
          } else al.y = al.z = al.w = al.x;
          accd[0] += (double)al.x * (double)mv[u].x; accd[1] += (double)al.y * (double)mv[u].y;
          accd[2] += (double)al.z * (double)mv[u].z; accd[3] += (double)al.w * (double)mv[u].w;
          asum[0] += (double)al.x; asum[1] += (double)al.y; asum[2] += (double)al.z; asum[3] += (double)al.w;
        }
      }
    }
    const float4 inv = den;   // (stored under the old name: the denominators)
    const float4 oh = make_float4((float)accd[0], (float)accd[1], (float)accd[2], (float)accd[3]);
    *reinterpret_cast<float4*>(out + (long)s * F + f) = oh;
    // Backward centres on out_hi + out_lo.  The centre is the weighted sum divided by the rounded coefficients' OWN sum
    // (round 6): g_a[r] = sum_f alpha_r g (m_r - centre) then adds up to zero over the segment, as the exact gradient
    // does (softmax shift invariance), whatever way the coefficients were rounded -- with the plain sum as the centre the
    // segment's g_a add up to (g . out) (1 - sum alpha) ~ 1e-7 (g . out), which for a segment of nearly equal rows is
    // 1e-5 ... 1e-1 of g_a itself and which the gate networks' weight gradients (h nearly equal over the segment) turn
    // into an error common to all their entries: +-2e-4 of the value from one ulp of a logit (tools/rowprog_gate_probe.py).
    // One row: centre = (alpha m) / alpha = m exactly; coefficients that sum to 1 exactly: centre = the sum, as before.
    if (out_lo) {
      double c[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) c[j] = asum[j] > 0.0 ? accd[j] / asum[j] : accd[j];
      *reinterpret_cast<float4*>(out_lo + (long)s * F + f) =
          make_float4((float)(c[0] - (double)oh.x), (float)(c[1] - (double)oh.y), (float)(c[2] - (double)oh.z),
                      (float)(c[3] - (double)oh.w));
    }
    if (PER_F) {
      *reinterpret_cast<float4*>(mx_out + (long)s * aF + ac) = mx;
      *reinterpret_cast<float4*>(inv_out + (long)s * aF + ac) = inv;
    } else if (f % fw == 0) {
      mx_out[(long)s * aF + ac] = mx.x;
      inv_out[(long)s * aF + ac] = inv.x;
    }
  }
}

// PER_F as above; otherwise the fw / 4 threads of a logit column are `grp` consecutive lanes of one wave (grp a power of
// two <= 64) and their partial sums of g_a meet in a shuffle reduction
template <bool PER_F>
__global__ __launch_bounds__(1024) void seg_attnpool_bwd_kernel(const float* __restrict__ a, int aF, int fw, int grp,
                                                                const float* __restrict__ mult,
                                                                const float* __restrict__ m, long ldm,
                                                                const int* __restrict__ rowptr,
                                                                const int* __restrict__ ridx, int F,
                                                                const float* __restrict__ out,
                                                                const float* __restrict__ mxs,
                                                                const float* __restrict__ invs,
                                                                const float* __restrict__ g_out,
                                                                float* __restrict__ g_a, float* __restrict__ g_m,
                                                                long ldgm, float* __restrict__ g_mult,
                                                                const float* __restrict__ out_lo) {
  const int s = blockIdx.x;
  const int r0 = rowptr[s], r1 = rowptr[s + 1];
  // every thread of a wave runs the same number of iterations (F rounded up to whole groups by the launch), so the
  // shuffles below are executed by all lanes
  for (int f0 = 4 * threadIdx.x; f0 < ((F + 255) / 256) * 256; f0 += 4 * blockDim.x) {
    const bool live = f0 < F;
    const int f = live ? f0 : 0;
    const int ac = PER_F ? f : f / fw;
    float4 mx, inv;
    if (PER_F) {
      mx = *reinterpret_cast<const float4*>(mxs + (long)s * aF + ac);
      inv = *reinterpret_cast<const float4*>(invs + (long)s * aF + ac);
    } else {
      mx.x = mxs[(long)s * aF + ac]; inv.x = invs[(long)s * aF + ac];
      mx.y = mx.z = mx.w = mx.x; inv.y = inv.z = inv.w = inv.x;
    }
    const float4 o4 = *reinterpret_cast<const float4*>(out + (long)s * F + f);
    // low part of the forward's fp64 sum (see the forward kernel): m - out = (m - out_hi) - out_lo
    const float4 ol = out_lo ? *reinterpret_cast<const float4*>(out_lo + (long)s * F + f) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 go = *reinterpret_cast<const float4*>(g_out + (long)s * F + f);
    for (int rb = r0; rb < r1; rb += AP_U) {
     // (the operands of AP_U rows requested together, round 6: one row per round trip before; same operations per row)
     float4 avv[AP_U], mvv[AP_U];
     float wvv[AP_U];
     long rrv[AP_U];
#pragma unroll
     for (int u = 0; u < AP_U; ++u) {
       const int rc_ = rb + u < r1 ? rb + u : r1 - 1;
       rrv[u] = ridx ? (long)ridx[rc_] : (long)rc_;
       wvv[u] = mult ? mult[rrv[u]] : 1.f;
       if (PER_F) avv[u] = *reinterpret_cast<const float4*>(a + rrv[u] * aF + ac);
       else avv[u].x = a[rrv[u] * aF + ac];
       mvv[u] = *reinterpret_cast<const float4*>(m + rrv[u] * ldm + f);
     }
#pragma unroll
     for (int u = 0; u < AP_U; ++u) {
      if (rb + u >= r1) break;                  // (uniform over the workgroup: no lane skips a shuffle another executes)
      const long r = rrv[u];
      const float w = wvv[u];
      float4 av = avv[u];
      if (!PER_F) av.y = av.z = av.w = av.x;
      const float4 mv = mvv[u];
      float4 al;
      al.x = (expf(av.x - mx.x) * w) / inv.x;    // `inv` holds the segment's denominator sum + eps (see the forward)
      if (PER_F) { al.y = (expf(av.y - mx.y) * w) / inv.y; al.z = (expf(av.z - mx.z) * w) / inv.z; al.w = (expf(av.w - mx.w) * w) / inv.w; }
      else al.y = al.z = al.w = al.x;
      const float4 gm = make_float4(al.x * go.x, al.y * go.y, al.z * go.z, al.w * go.w);
      const float4 dm = make_float4((mv.x - o4.x) - ol.x, (mv.y - o4.y) - ol.y, (mv.z - o4.z) - ol.z, (mv.w - o4.w) - ol.w);
      const float4 t = make_float4(gm.x * dm.x, gm.y * dm.y, gm.z * dm.z, gm.w * dm.w);
      if (live && g_m) *reinterpret_cast<float4*>(g_m + r * ldgm + f) = gm;
      if (PER_F) {
        if (live) *reinterpret_cast<float4*>(g_a + r * aF + ac) = t;
      } else {
        float p = live ? (t.x + t.y) + (t.z + t.w) : 0.f;
        for (int o = grp >> 1; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
        // d out / d mult[r] = e inv sum_f g_out (m - out): from the UN-multiplied softmax term, so that a zero weight
        // (whose row contributes nothing to out) still gets its -- in general non-zero -- gradient
        float q = 0.f;
        if (g_mult) {                                    // kernel argument: uniform branch
          const float e0 = expf(av.x - mx.x) / inv.x;
          q = live ? e0 * ((go.x * dm.x + go.y * dm.y) + (go.z * dm.z + go.w * dm.w)) : 0.f;
          for (int o = grp >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        }
        if (live && (threadIdx.x & (grp - 1)) == 0) {
          g_a[r * aF + ac] = p;
          if (g_mult) g_mult[r] = q;
        }
      }
     }
    }
  }
}

static bool attnpool_ok(int aF, int F, long ldm, const void* a, const void* m, const void* out) {
  if (F <= 0 || aF <= 0 || F % aF != 0 || F % 4 != 0 || ldm % 4 != 0) return false;
  if (((((uintptr_t)m) | ((uintptr_t)out)) & 15) != 0) return false;
  const int fw = F / aF;
  if (fw == 1) return (aF % 4) == 0 && (((uintptr_t)a) & 15) == 0;
  const int grp = fw / 4;
  return fw % 4 == 0 && grp <= 64 && (grp & (grp - 1)) == 0;
}
bool seg_attnpool_fast(int aF, int F, long ldm, const void* a, const void* m, const void* out) {
  return attnpool_ok(aF, F, ldm, a, m, out);
}
static int attnpool_threads(int F) {
  int t = ((F / 4 + 63) / 64) * 64;
  return t > 1024 ? 1024 : t;
}
int seg_attnpool_fwd_launch(const float* a, int aF, const float* mult, const float* m, long ldm, const int* rowptr,
                            const int* ridx, int S, int F, float eps, float* out, float* mx, float* inv, hipStream_t s,
                            float* out_lo) {
  if (S <= 0) return CGAT_OK;
  CGAT_CHECK_ARG(attnpool_ok(aF, F, ldm, a, m, out), "segment_attention_pool: unsupported shape (F=%d, aF=%d)", F, aF);
  CGAT_CHECK_ARG((((uintptr_t)out_lo) & 15) == 0, "segment_attention_pool: out_lo must be 16-byte aligned");
  const int fw = F / aF;
  CGAT_PROF("seg_attnpool_fwd", s);
  if (fw == 1)
    hipLaunchKernelGGL(seg_attnpool_fwd_kernel<true>, dim3(S), dim3(attnpool_threads(F)), 0, s, a, aF, fw, mult, m, ldm,
                       rowptr, ridx, F, eps, out, mx, inv, out_lo);
  else
    hipLaunchKernelGGL(seg_attnpool_fwd_kernel<false>, dim3(S), dim3(attnpool_threads(F)), 0, s, a, aF, fw, mult, m, ldm,
                       rowptr, ridx, F, eps, out, mx, inv, out_lo);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}
int seg_attnpool_bwd_launch(const float* a, int aF, const float* mult, const float* m, long ldm, const int* rowptr,
                            const int* ridx, int S, int F, const float* out, const float* mx, const float* inv, const float* g_out, float* g_a,
                            float* g_m, long ldgm, float* g_mult, hipStream_t s, const float* out_lo) {
  if (S <= 0) return CGAT_OK;
  CGAT_CHECK_ARG(attnpool_ok(aF, F, ldm, a, m, out) && ldgm % 4 == 0 && (((uintptr_t)out_lo) & 15) == 0 && (!g_m || (((uintptr_t)g_m) & 15) == 0),
                 "segment_attention_pool backward: unsupported shape (F=%d, aF=%d)", F, aF);
  CGAT_CHECK_ARG(!g_mult || aF == 1, "segment_attention_pool backward: gradient of the multiplier needs one logit column");
  const int fw = F / aF;
  CGAT_PROF("seg_attnpool_bwd", s);
  if (fw == 1)
    hipLaunchKernelGGL(seg_attnpool_bwd_kernel<true>, dim3(S), dim3(attnpool_threads(F)), 0, s, a, aF, fw, 1, mult, m, ldm,
                       rowptr, ridx, F, out, mx, inv, g_out, g_a, g_m, ldgm, g_mult, out_lo);
  else
    hipLaunchKernelGGL(seg_attnpool_bwd_kernel<false>, dim3(S), dim3(attnpool_threads(F)), 0, s, a, aF, fw, fw / 4, mult, m,
                       ldm, rowptr, ridx, F, out, mx, inv, g_out, g_a, g_m, ldgm, g_mult, out_lo);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}
