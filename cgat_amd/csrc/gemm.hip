// fp32 GEMM engine on the CDNA4 f32-input matrix cores (v_mfma_f32_32x32x2_f32).
//
//   C[m,n] = epilogue( alpha * sum_k A(m,k) * B(k,n) )
//
// A and B are addressed through (row stride, "k-major" flag, optional index vectors), so the
// same kernel serves x @ W^T (weights as stored by torch), g @ W, the weight-gradient
// products g^T @ x with the long reduction dimension split over workgroups, and the per-edge
// products whose rows are gathered through the destination-sorted permutation.
//
// Tile: 128 x 128 x 32 per 256-thread workgroup (4 waves, each 64 x 64 = 2 x 2 MFMA blocks of
// 32 x 32, 64 accumulator VGPRs).  Operand tiles are staged K-major in LDS ([k][m], [k][n]) so
// that every MFMA operand fetch is a conflict-free 32-lane row read; global loads are
// 16-byte vectors along whichever dimension is contiguous in memory, double-buffered through
// registers.  The arithmetic is exact fp32 (MFMA f32 = k-ordered fmaf chain).
#include "common.h"
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 128
#define BN 128
#define BK 32

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case CGAT_ACT_TANH: return tanhf(v);
    case CGAT_ACT_LEAKY: return v > 0.f ? v : 0.01f * v;
    case CGAT_ACT_RELU: return v > 0.f ? v : 0.f;
    default: return v;
  }
}

// Loads one 128 x 32 operand tile into 4 float4 registers per thread.
//  KM == false : element (r, k) at base[row(r) * ld + k]      (k contiguous)
//  KM == true  : element (r, k) at base[krow(k) * ld + r]     (r contiguous)
template <bool KM>
struct TileLoader {
  const float* base;
  long ld;
  int R;             // extent of the tile's row dimension (M or N)
  int r0;            // first row of this tile
  const int* rgather;  // !KM: optional row index
  const int* kgather;  // KM: optional k index
  bool vec;          // 16-byte loads legal
  long roff[4];      // !KM: precomputed row offsets (elements), -1 = out of range

  __device__ void init(int tid) {
    if (!KM) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int f = tid + 256 * i;
        int r = r0 + (f >> 3);
        if (r < R) {
          long row = rgather ? (long)rgather[r] : (long)r;
          roff[i] = row * ld;
        } else {
          roff[i] = -1;
        }
      }
    }
  }

  __device__ void load(int tid, int k0, int kend, float4 (&v)[4]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int f = tid + 256 * i;
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!KM) {
        int k = k0 + 4 * (f & 7);
        if (roff[i] >= 0 && k < kend) {
          const float* p = base + roff[i] + k;
          if (vec && k + 3 < kend) {
            t = *reinterpret_cast<const float4*>(p);
          } else {
            t.x = p[0];
            if (k + 1 < kend) t.y = p[1];
            if (k + 2 < kend) t.z = p[2];
            if (k + 3 < kend) t.w = p[3];
          }
        }
      } else {
        int k = k0 + (f >> 5);
        int r = r0 + 4 * (f & 31);
        if (k < kend && r < R) {
          long krow = kgather ? (long)kgather[k] : (long)k;
          const float* p = base + krow * ld + r;
          if (vec && r + 3 < R) {
            t = *reinterpret_cast<const float4*>(p);
          } else {
            t.x = p[0];
            if (r + 1 < R) t.y = p[1];
            if (r + 2 < R) t.z = p[2];
            if (r + 3 < R) t.w = p[3];
          }
        }
      }
      v[i] = t;
    }
  }

  // LDS image is [k][r] with pitch P
  template <int P>
  __device__ void store(int tid, float* lds, const float4 (&v)[4]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int f = tid + 256 * i;
      if (!KM) {
        int r = f >> 3, kq = f & 7;
        float* d = lds + (4 * kq) * P + r;
        d[0] = v[i].x;
        d[P] = v[i].y;
        d[2 * P] = v[i].z;
        d[3 * P] = v[i].w;
      } else {
        int k = f >> 5, rq = f & 31;
        *reinterpret_cast<float4*>(lds + k * P + 4 * rq) = v[i];
      }
    }
  }
};

template <bool AKM, bool BKM>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmParams p) {
  constexpr int PA = AKM ? 132 : 129;
  constexpr int PB = BKM ? 132 : 129;
  __shared__ __attribute__((aligned(16))) float As[2][BK * PA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * PB];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hi = lane >> 5;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  // K range of this split
  const int z = blockIdx.y;
  const int kbeg = z * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);

  TileLoader<AKM> la{p.A, p.lda, p.M, m0, p.a_rgather, nullptr, p.a_vec != 0, {0, 0, 0, 0}};
  TileLoader<BKM> lb{p.B, p.ldb, p.N, n0, nullptr, p.b_kgather, p.b_vec != 0, {0, 0, 0, 0}};
  la.init(tid);
  lb.init(tid);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;

  // two-level summation for long reductions: partial sums over 512 k, then added to the totals
  f32x16 tot[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 16; ++t) tot[i][j][t] = 0.f;
  float4 ra[4], rb[4];
  const int nchunks = (kend - kbeg + BK - 1) / BK;
  if (nchunks > 0) {
    la.load(tid, kbeg, kend, ra);
    lb.load(tid, kbeg, kend, rb);
    la.template store<PA>(tid, As[0], ra);
    lb.template store<PB>(tid, Bs[0], rb);
  }
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int cur = c & 1;
    if ((c & 15) == 0 && c > 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          tot[i][j] += acc[i][j];
#pragma unroll
          for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;
        }
    }
    if (c + 1 < nchunks) {
      la.load(tid, kbeg + (c + 1) * BK, kend, ra);
      lb.load(tid, kbeg + (c + 1) * BK, kend, rb);
    }
    const float* as = As[cur] + hi * PA + wm + r;
    const float* bs = Bs[cur] + hi * PB + wn + r;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float a0 = as[(2 * kk) * PA], a1 = as[(2 * kk) * PA + 32];
      float b0 = bs[(2 * kk) * PB], b1 = bs[(2 * kk) * PB + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (c + 1 < nchunks) {
      la.template store<PA>(tid, As[cur ^ 1], ra);
      lb.template store<PB>(tid, Bs[cur ^ 1], rb);
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] += tot[i][j];
  // ---- epilogue ----
  float* Cbase = p.C;
  const bool slab = p.splits > 1;
  if (slab) Cbase = p.slab + (long)z * p.M * p.N;
  const long ldc = slab ? (long)p.N : p.ldc;
#pragma unroll
  for (int bi = 0; bi < 2; ++bi) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int m = m0 + wm + bi * 32 + (t & 3) + 8 * (t >> 2) + 4 * hi;
      if (m >= p.M) continue;
      long crow = m;
      const float *g1 = nullptr, *g2 = nullptr;
      if (!slab) {
        if (p.c_scatter) crow = p.c_scatter[m];
        if (p.add1) g1 = p.add1 + (long)p.add1_idx[m] * p.ld_add;
        if (p.add2) g2 = p.add2 + (long)p.add2_idx[m] * p.ld_add;
      }
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) {
        const int n = n0 + wn + bj * 32 + r;
        if (n >= p.N) continue;
        float v = acc[bi][bj][t];
        if (!slab) {
          v *= p.alpha;
          if (p.bias) v += p.bias[n];
          if (g1) v += g1[n];
          if (g2) v += g2[n];
          v = act_apply(v, p.act);
          if (p.beta != 0.f) v += p.beta * Cbase[crow * ldc + n];
        }
        Cbase[crow * ldc + n] = v;
      }
    }
  }
}

// out[m*ldc + n] = act(alpha * sum_z slab[z][m][n] + bias[n]) + beta * out
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, int splits, int M, int N, float* __restrict__ C,
                                     long ldc, float alpha, float beta, const float* __restrict__ bias, int act) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)M * N;
  if (i >= total) return;
  int m = (int)(i / N), n = (int)(i % N);
  float s = 0.f;
  for (int z = 0; z < splits; ++z) s += slab[(long)z * total + i];
  s *= alpha;
  if (bias) s += bias[n];
  s = act_apply(s, act);
  float* d = C + (long)m * ldc + n;
  if (beta != 0.f) s += beta * *d;
  *d = s;
}

size_t gemm_ws_bytes(const GemmParams& p) { return p.splits > 1 ? ws_round((size_t)p.splits * p.M * p.N, 4) : 0; }

// Chooses a split count for reductions with a long K and few output tiles.
int gemm_pick_splits(int M, int N, int K) {
  long tiles = (long)cdiv(M, BM) * cdiv(N, BN);
  if (tiles >= 256 || K < 4 * BK * 8) return 1;
  long want = (512 + tiles - 1) / tiles;
  long maxs = K / (BK * 8);  // at least 8 chunks per split
  long s = want < maxs ? want : maxs;
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  return (int)s;
}

int gemm_launch(GemmParams p, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (p.M <= 0 || p.N <= 0) return CGAT_OK;
  CGAT_CHECK_ARG(p.K >= 0, "gemm: K<0");
  if (p.splits < 1) p.splits = 1;
  if (p.splits > 1) {
    CGAT_CHECK_ARG(!p.c_scatter && !p.add1 && !p.add2, "gemm: split-K cannot be combined with scatter/gather-add epilogues");
    int kper = cdiv(p.K, p.splits);
    kper = ((kper + BK - 1) / BK) * BK;
    p.k_per_split = kper;
    p.splits = cdiv(p.K, kper);
    if (p.splits < 1) p.splits = 1;
  }
  if (p.splits == 1) p.k_per_split = p.K > 0 ? p.K : 1;
  if (p.splits > 1) {
    size_t need = gemm_ws_bytes(p);
    if (!ws || ws_bytes < need) {
      cgat_set_error("gemm: split-K workspace too small (%zu < %zu)", ws_bytes, need);
      return CGAT_ERR_WORKSPACE;
    }
    p.slab = (float*)ws;
  }
  // 16-byte vector loads are legal when the contiguous dimension starts 16B-aligned in every row
  auto aligned = [](const float* ptr, long ld) { return (((uintptr_t)ptr) & 15) == 0 && (ld % 4) == 0; };
  p.a_vec = aligned(p.A, p.lda) ? 1 : 0;
  p.b_vec = aligned(p.B, p.ldb) ? 1 : 0;
  dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN), p.splits);
  {
    CGAT_PROF("gemm_f32", stream);
    if (!p.a_kmajor && !p.b_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, stream, p);
    else if (!p.a_kmajor && p.b_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, stream, p);
    else if (p.a_kmajor && !p.b_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, stream, p);
  }
  CGAT_LAUNCH_CHECK();
  if (p.splits > 1) {
    long total = (long)p.M * p.N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, p.slab, p.splits, p.M, p.N,
                       p.C, p.ldc, p.alpha, p.beta, p.bias, p.act);
    CGAT_LAUNCH_CHECK();
  }
  return CGAT_OK;
}
