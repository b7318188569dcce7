// Row-wise / elementwise kernels around the matrix-core products: LayerNorm(no affine)+tanh
// of the predicted layers (reference Hypernetworksmp.py:103-107), activation derivatives,
// column sums for bias gradients, the H_Net damping mix (Hypernetworksmp.py:310-312).
// All HBM-bound, one wave per row or grid-stride; accurate tanhf/rsqrtf (no fast-math).
#include "common.h"
#include "kernels.h"
#include "mfma_bf16.h"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// y = tanh((u - mean) * rsqrt(var + eps)), biased variance; one wave per row
__global__ void layernorm_tanh_fwd_kernel(const float* __restrict__ u, float* __restrict__ y, int rows, int W,
                                          float eps) {
  int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* x = u + (long)row * W;
  float s = 0.f;
  for (int c = lane; c < W; c += 64) s += x[c];
  float mean = wave_sum(s) / W;
  float v = 0.f;
  for (int c = lane; c < W; c += 64) {
    float d = x[c] - mean;
    v += d * d;
  }
  float rstd = rsqrtf(wave_sum(v) / W + eps);
  for (int c = lane; c < W; c += 64) y[(long)row * W + c] = tanhf((x[c] - mean) * rstd);
}

// gu = rstd * (gx - mean(gx) - xhat * mean(gx * xhat)),  gx = gy * (1 - y^2)
__global__ void layernorm_tanh_bwd_kernel(const float* __restrict__ u, const float* __restrict__ y,
                                          const float* __restrict__ gy, float* __restrict__ gu, int rows, int W,
                                          float eps) {
  int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* x = u + (long)row * W;
  const float* yy = y + (long)row * W;
  const float* g = gy + (long)row * W;
  float s = 0.f;
  for (int c = lane; c < W; c += 64) s += x[c];
  float mean = wave_sum(s) / W;
  float v = 0.f;
  for (int c = lane; c < W; c += 64) {
    float d = x[c] - mean;
    v += d * d;
  }
  float rstd = rsqrtf(wave_sum(v) / W + eps);
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < W; c += 64) {
    float gx = g[c] * (1.f - yy[c] * yy[c]);
    s1 += gx;
    s2 += gx * (x[c] - mean) * rstd;
  }
  s1 = wave_sum(s1) / W;
  s2 = wave_sum(s2) / W;
  for (int c = lane; c < W; c += 64) {
    float gx = g[c] * (1.f - yy[c] * yy[c]);
    float xh = (x[c] - mean) * rstd;
    gu[(long)row * W + c] = rstd * (gx - s1 - xh * s2);
  }
}

int layernorm_tanh_fwd_launch(const float* u, float* y, int rows, int W, float eps, hipStream_t s) {
  if (rows <= 0) return CGAT_OK;
  hipLaunchKernelGGL(layernorm_tanh_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, u, y, rows, W, eps);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

int layernorm_tanh_bwd_launch(const float* u, const float* y, const float* gy, float* gu, int rows, int W, float eps,
                              hipStream_t s) {
  if (rows <= 0) return CGAT_OK;
  hipLaunchKernelGGL(layernorm_tanh_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, u, y, gy, gu, rows, W, eps);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// derivative through an activation, expressed with the post-activation value y
__global__ void act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gy, float* __restrict__ gpre,
                               long n, int act) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float yv = y[i], g = gy[i];
    float d;
    switch (act) {
      case CGAT_ACT_TANH: d = 1.f - yv * yv; break;
      case CGAT_ACT_LEAKY: d = yv > 0.f ? 1.f : 0.01f; break;
      case CGAT_ACT_RELU: d = yv > 0.f ? 1.f : 0.f; break;
      default: d = 1.f;
    }
    gpre[i] = g * d;
  }
}

int act_bwd_launch(const float* y, const float* gy, float* gpre, long n, int act, hipStream_t s);
// the same with max |gpre| folded into gmax[0] (NOT zeroed here): the per-tensor scale the fp16 forms of the kernels
// that consume gpre need (vector-attention backward, edge_hidden_backward_impl); 16-byte vectors, n % 4 == 0
__global__ __launch_bounds__(256) void act_bwd_max_kernel(const float4* __restrict__ y, const float4* __restrict__ gy,
                                                          float4* __restrict__ gpre, long n4, float* __restrict__ gmax) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  float m = 0.f;
  for (; i < n4; i += stride) {
    const float4 yv = y[i], g = gy[i];
    const float4 r = make_float4(g.x * (yv.x > 0.f ? 1.f : 0.01f), g.y * (yv.y > 0.f ? 1.f : 0.01f),
                                 g.z * (yv.z > 0.f ? 1.f : 0.01f), g.w * (yv.w > 0.f ? 1.f : 0.01f));
    gpre[i] = r;
    m = fmaxf(fmaxf(m, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
  }
  block_absmax_commit(m, gmax);
}

static inline int grid_for(long n) {
  long b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// LeakyReLU backward + max |gpre| (gmax zeroed by the caller); falls back to the plain kernel (gmax untouched, returns
// CGAT_ERR_UNSUPPORTED-free: the caller checks `*used`) when the vector form does not apply
int act_bwd_leaky_max_launch(const float* y, const float* gy, float* gpre, long n, float* gmax, hipStream_t s, bool* used) {
  *used = false;
  if (n <= 0) return CGAT_OK;
  if ((n % 4) != 0 || ((((uintptr_t)y) | ((uintptr_t)gy) | ((uintptr_t)gpre)) & 15) != 0)
    return act_bwd_launch(y, gy, gpre, n, CGAT_ACT_LEAKY, s);
  const long n4 = n / 4;
  long b = (n4 + 255) / 256;
  const int grid = (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
  hipLaunchKernelGGL(act_bwd_max_kernel, dim3(grid), dim3(256), 0, s, (const float4*)y, (const float4*)gy, (float4*)gpre, n4,
                     gmax);
  CGAT_LAUNCH_CHECK();
  *used = true;
  return CGAT_OK;
}

int act_bwd_launch(const float* y, const float* gy, float* gpre, long n, int act, hipStream_t s) {
  if (n <= 0) return CGAT_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, y, gy, gpre, n, act);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// ---- column sums: partial[chunk][c] over 128-row chunks (256 threads = 64 columns x 4 row lanes,
// LDS-combined), then a second pass of the same shape over the partials; fixed summation order,
// hence deterministic ----
#define COLSUM_ROWS 128
__global__ void colsum_partial_kernel(const float* __restrict__ x, long ldx, int rows, int cols,
                                      float* __restrict__ partial, float alpha) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int r0 = blockIdx.y * COLSUM_ROWS, r1 = min(rows, r0 + COLSUM_ROWS);
  float s = 0.f;
  if (c < cols)
    for (int r = r0 + rl; r < r1; r += 4) s += x[(long)r * ldx + c];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < cols) partial[(long)blockIdx.y * cols + c] = alpha * (red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}
static inline int colsum_chunks(int rows) { return cdiv(rows > 0 ? rows : 1, COLSUM_ROWS); }
size_t colsum_ws_bytes(int rows, int cols) {
  size_t c1 = colsum_chunks(rows), c2 = colsum_chunks((int)c1);
  return ws_round(c1 * cols, 4) + ws_round(c2 * cols, 4);
}
int colsum_launch(const float* x, long ldx, int rows, int cols, float* out, float alpha, void* ws, size_t ws_bytes,
                  hipStream_t s) {
  if (cols <= 0) return CGAT_OK;
  if (!ws || ws_bytes < colsum_ws_bytes(rows, cols)) {
    cgat_set_error("colsum: workspace too small");
    return CGAT_ERR_WORKSPACE;
  }
  float* bufs[2] = {(float*)ws, (float*)((char*)ws + ws_round((size_t)colsum_chunks(rows) * cols, 4))};
  const float* src = x;
  long ld = ldx;
  int n = rows > 0 ? rows : 0, level = 0;
  while (true) {
    int chunks = colsum_chunks(n);
    float* dst = chunks == 1 ? out : bufs[level & 1];
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(cols, 64), chunks), dim3(256), 0, s, src, ld, n, cols, dst,
                       level == 0 ? alpha : 1.f);
    CGAT_LAUNCH_CHECK();
    if (chunks == 1) break;
    src = dst;
    ld = cols;
    n = chunks;
    ++level;
  }
  return CGAT_OK;
}

// ---- H_Net hyper input: out = d*a + (1-d)*b, d a device scalar ----
__global__ void mix_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ d,
                           float* __restrict__ out, long n) {
  float dv = d[0];
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = dv * a[i] + (1.f - dv) * b[i];
}
int mix_launch(const float* a, const float* b, const float* d, float* out, long n, hipStream_t s) {
  if (n <= 0) return CGAT_OK;
  hipLaunchKernelGGL(mix_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, b, d, out, n);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// ga = d*g ; gb += (1-d)*g ; gd = sum g*(a-b)   (block partials -> fixed-order final sum)
__global__ void mix_bwd_kernel(const float* __restrict__ g, const float* __restrict__ a, const float* __restrict__ b,
                               const float* __restrict__ d, float* __restrict__ ga, float* __restrict__ gb, long n,
                               float* __restrict__ partial) {
  __shared__ float red[4];
  float dv = d[0];
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  float s = 0.f;
  for (; i < n; i += stride) {
    float gv = g[i];
    if (ga) ga[i] = dv * gv;
    if (gb) gb[i] += (1.f - dv) * gv;
    s += gv * (a[i] - b[i]);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sum_partials_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += partial[i];
    out[0] = s;
  }
}
int mix_bwd_launch(const float* g, const float* a, const float* b, const float* d, float* ga, float* gb_accum,
                   float* gd, long n, void* ws, size_t ws_bytes, hipStream_t s) {
  int blocks = grid_for(n);
  if (blocks > 1024) blocks = 1024;
  if (!ws || ws_bytes < (size_t)blocks * 4) {
    cgat_set_error("mix_bwd: workspace too small");
    return CGAT_ERR_WORKSPACE;
  }
  hipLaunchKernelGGL(mix_bwd_kernel, dim3(blocks), dim3(256), 0, s, g, a, b, d, ga, gb_accum, n, (float*)ws);
  CGAT_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, s, (const float*)ws, blocks, gd);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] += alpha * x[i];
}
int axpy_launch(float* y, const float* x, float alpha, long n, hipStream_t s) {
  if (n <= 0) return CGAT_OK;
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, s, y, x, alpha, n);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// out = alpha * x
__global__ void scale_kernel(const float* __restrict__ x, float alpha, float* __restrict__ out, long n4, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long j = i; j < n4; j += stride) {
    float4 v = reinterpret_cast<const float4*>(x)[j];
    v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
    reinterpret_cast<float4*>(out)[j] = v;
  }
  for (long j = 4 * n4 + i; j < n; j += stride) out[j] = alpha * x[j];
}
int scale_launch(const float* x, float alpha, float* out, long n, hipStream_t s) {
  if (n <= 0) return CGAT_OK;
  const bool vec = ((((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0;
  const long n4 = vec ? n / 4 : 0;
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n4 > 0 ? n4 : n)), dim3(256), 0, s, x, alpha, out, n4, n);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

__global__ void copy2d_kernel(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd, int rows,
                              int cols) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)rows * cols;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    long r = i / cols, c = i % cols;
    dst[r * ldd + c] = src[r * lds + c];
  }
}
int copy2d_launch(const float* src, long lds, float* dst, long ldd, int rows, int cols, hipStream_t s) {
  long total = (long)rows * cols;
  if (total <= 0) return CGAT_OK;
  hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for(total)), dim3(256), 0, s, src, lds, dst, ldd, rows, cols);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// up to four such copies in one launch (blockIdx.y = the job): the stacking of MH_A | MH_M first-layer weights and
// biases and the un-stacking of their gradients were four launches per pass (41 per 4-layer step)
__global__ void copy2d_multi_kernel(Copy2DJobs j) {
  const Copy2DJob& b = j.job[blockIdx.y];
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)b.rows * b.cols;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const long r = i / b.cols, c = i % b.cols;
    b.dst[r * b.ldd + c] = b.src[r * b.lds + c];
  }
}
int copy2d_multi_launch(const Copy2DJobs& j, hipStream_t s) {
  long most = 0;
  for (int k = 0; k < j.n; ++k) most = (long)j.job[k].rows * j.job[k].cols > most ? (long)j.job[k].rows * j.job[k].cols : most;
  if (j.n <= 0 || most <= 0) return CGAT_OK;
  hipLaunchKernelGGL(copy2d_multi_kernel, dim3(grid_for(most), j.n), dim3(256), 0, s, j);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// Also the library's memset: NO hipMemsetAsync anywhere in the library (round 4).  A 4-byte hipMemsetAsync captured into a
// hipGraph (cgat_amd.GraphedStep) was not reliably in effect before the kernel that followed it on the SECOND replay
// (ROCm 7.2, memory from the capture's private pool): the slot of a maximum kept the bits a later operator of the
// previous replay had left in the reused block, atomicMax compared against them, the operand scale came out wrong and the
// step returned NaN -- replay 0 and every eager run were fine.  The same zeroing as a kernel node replays bit-identically
// (tests/test_capture.py).
// out[i] = sum_z slabs[z * stride + i], z = 0 .. n-1 in that order, starting from 0.f (what n accumulating passes over a
// zero-filled buffer produce, bit for bit)
__global__ void sum_slabs_kernel(const float* __restrict__ slabs, int n, long stride, float* __restrict__ out, long count) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float s = 0.f;
  for (int z = 0; z < n; ++z) s += slabs[(long)z * stride + i];
  out[i] = s;
}
int sum_slabs_launch(const float* slabs, int n, long stride, float* out, long count, hipStream_t s) {
  if (count <= 0) return CGAT_OK;
  hipLaunchKernelGGL(sum_slabs_kernel, dim3(cdiv(count, 256)), dim3(256), 0, s, slabs, n, stride, out, count);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

__global__ void fill_kernel(float* __restrict__ p, float v, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
int fill_launch(float* p, float v, long n, hipStream_t s) {
  if (n <= 0) return CGAT_OK;
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, v, n);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}
