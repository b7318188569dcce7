// Weight gradients of the width-128 dense layers (hypernetwork trunks, linear terms of the predicted layers):
//     out_k[o][i] = sum_n G[n,o] X_k[n,i]   (k < NX: up to two right operands sharing G),   bsum[o] = sum_n G[n,o]
// i.e. what autograd computes for nn.Linear's weight and bias (reference Hypernetworksmp.py:60-70 BatchLinear,
// 77-83 HyperLinear's U and B terms under autograd).  The reduction runs over the ROW index of two row-major
// operands: 85 MB (129 MB) of reads for 2.7 (5.5) GFLOP -- HBM-bound, and the f32-input MFMA
// v_mfma_f32_32x32x2_f32 takes exactly that layout without any staging: its A operand is a 32 x 2 (m x k) slice
// with lane l holding (m = l % 32, k = l / 32), so with k = the row number a wave's two half-waves read 32
// consecutive floats of two consecutive rows each -- straight from global memory, no LDS, no transposition, exact
// fp32 products.  Replaces, per dense layer, a split-K pass of the generic GEMM engine (67 us), its slab reduction
// (12 us) and a separate column-sum pass over G (11 us).
//
// Tried and dropped (round 1): staging each 32-row batch once per workgroup through a double-buffered LDS image
// (G as four component planes, X as float4 rows; no redundant loads) -- 48 us per launch against 42 us for this form.
//
// Workgroup = 8 waves = 2 row halves x 4 column components: wave (h, w) accumulates
//     D_j[m][n] = sum_rows G[row][4 m + w] * X[row][4 n + j]        j = 0..3
// (the lane's float4 of X gives the B operands of four interleaved column blocks), so a lane's four accumulators at
// register t are four consecutive floats of out[4 m + w][4 n ..]: one 16-byte store.  The two row halves are added
// through LDS, one slab per workgroup, then a deterministic reduction over the slabs.
#include "common.h"
#include "kernels.h"
#include "mfma_bf16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// the workgroup's slab tile `o` = [NX][128][128] + [128] partial sums over rows nbeg .. nend-1
template <int NX>
__device__ __forceinline__ void rows_dw128_body(const float* __restrict__ G, long ldg, const float* __restrict__ X1,
                                                long ldx1, const float* __restrict__ X2, long ldx2,
                                                float* __restrict__ o, int nbeg, int nend, float* red) {
  constexpr int U = 4;                                 // k-steps (2 rows each) per register batch; three batches
                                                       // rotate so that two are in flight while one is consumed (one
                                                       // ahead left the waves waiting on HBM latency: 50 us vs 18 us of MFMA)
  // red: [NX][128][128] + [128] floats of LDS, the second row half's partial sums
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w = wave & 3, h = wave >> 2;
  const int l32 = lane & 31, hi = lane >> 5;
  // this wave's rows: nbeg + 2 U (2 b + h) + 2 u + hi for batch b (the two halves interleave batch by batch)
  const int nbatch = (nend - nbeg + 4 * U - 1) / (4 * U);

  f32x16 acc[NX][4];
#pragma unroll
  for (int k = 0; k < NX; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[k][j][t] = 0.f;
  float cs = 0.f;

  float ga[3][U];
  float4 xa[3][U], xb[3][NX > 1 ? U : 1];
#define DW_LOAD(buf_, b_)                                                                     \
  {                                                                                           \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                           \
      const int row = nbeg + 2 * U * (2 * (b_) + h) + 2 * u + hi;                             \
      const bool ok = row < nend;                                                             \
      const long rr = ok ? row : nbeg;                                                        \
      const float g = G[rr * ldg + 4 * l32 + w];                                              \
      const float4 x = *reinterpret_cast<const float4*>(X1 + rr * ldx1 + 4 * l32);            \
      ga[buf_][u] = ok ? g : 0.f;                                                             \
      xa[buf_][u] = x;                                                                        \
      if constexpr (NX > 1) xb[buf_][u] = *reinterpret_cast<const float4*>(X2 + rr * ldx2 + 4 * l32); \
    }                                                                                         \
  }
#define DW_MFMA(buf_)                                                                         \
  {                                                                                           \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                           \
      const float a = ga[buf_][u];                                                            \
      cs += a;                                                                                \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xa[buf_][u].x, acc[0][0], 0, 0, 0); \
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xa[buf_][u].y, acc[0][1], 0, 0, 0); \
      acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xa[buf_][u].z, acc[0][2], 0, 0, 0); \
      acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xa[buf_][u].w, acc[0][3], 0, 0, 0); \
      if constexpr (NX > 1) {                                                                 \
        acc[NX - 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xb[buf_][u].x, acc[NX - 1][0], 0, 0, 0); \
        acc[NX - 1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xb[buf_][u].y, acc[NX - 1][1], 0, 0, 0); \
        acc[NX - 1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xb[buf_][u].z, acc[NX - 1][2], 0, 0, 0); \
        acc[NX - 1][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xb[buf_][u].w, acc[NX - 1][3], 0, 0, 0); \
      }                                                                                       \
    }                                                                                         \
  }
  // batches past the end load a valid row with a zeroed G value (see DW_LOAD): no guards needed on the loads
  DW_LOAD(0, 0);
  DW_LOAD(1, 1);
  for (int b = 0; b < nbatch; b += 3) {
    DW_LOAD(2, b + 2);
    DW_MFMA(0);
    if (b + 1 < nbatch) {
      DW_LOAD(0, b + 3);
      DW_MFMA(1);
    }
    if (b + 2 < nbatch) {
      DW_LOAD(1, b + 4);
      DW_MFMA(2);
    }
  }
#undef DW_LOAD
#undef DW_MFMA
  // a zeroed G value makes the whole product vanish, so X rows past the end need no masking (they are read from a
  // valid row); bsum: lanes l and l + 32 hold the two k halves of column 4 l32 + w
  cs += __shfl_xor(cs, 32);
  // D layout: register t of lane (l32, hi) is row m = (t & 3) + 8 (t >> 2) + 4 hi, column n = l32
  if (h == 1) {
#pragma unroll
    for (int k = 0; k < NX; ++k)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int m = (t & 3) + 8 * (t >> 2) + 4 * hi;
        *reinterpret_cast<float4*>(&red[(k * 128 + 4 * m + w) * 128 + 4 * l32]) =
            make_float4(acc[k][0][t], acc[k][1][t], acc[k][2][t], acc[k][3][t]);
      }
    if (hi == 0) red[NX * 128 * 128 + 4 * l32 + w] = cs;
  }
  __syncthreads();
  if (h == 0) {
#pragma unroll
    for (int k = 0; k < NX; ++k)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int m = (t & 3) + 8 * (t >> 2) + 4 * hi;
        const int idx = (k * 128 + 4 * m + w) * 128 + 4 * l32;
        const float4 r = *reinterpret_cast<const float4*>(&red[idx]);
        *reinterpret_cast<float4*>(o + idx) =
            make_float4(acc[k][0][t] + r.x, acc[k][1][t] + r.y, acc[k][2][t] + r.z, acc[k][3][t] + r.w);
      }
    if (hi == 0) o[NX * 128 * 128 + 4 * l32 + w] = cs + red[NX * 128 * 128 + 4 * l32 + w];
  }
}

template <int NX>
__global__ __launch_bounds__(512, 1) void rows_dw128_kernel(const float* __restrict__ G, long ldg,
                                                            const float* __restrict__ X1, long ldx1,
                                                            const float* __restrict__ X2, long ldx2,
                                                            float* __restrict__ slab, int rows, int rows_per_wg) {
  constexpr int SLAB = NX * 128 * 128 + 128;           // floats per workgroup: out_1, (out_2,) bsum
  extern __shared__ float red[];
  const int nbeg = blockIdx.x * rows_per_wg;
  rows_dw128_body<NX>(G, ldg, X1, ldx1, X2, ldx2, slab + (long)blockIdx.x * SLAB, nbeg, min(rows, nbeg + rows_per_wg), red);
}

// Many dense layers' weight gradients in ONE launch (the 4 x (n_fc + 2) products of a hypernetwork backward: 24 launches
// + 24 reductions of ~40 + 12 us each, with a dispatch gap between every pair, become two launches).  Unit = (item, row
// split): `splits` workgroups per item, chosen so that all units fit the chip at once; every item has one right operand
// (a layer with two -- the linear terms B, U of a predicted layer share G -- is two items).
__global__ __launch_bounds__(512, 1) void rows_dw128_batch_kernel(DwBatchDesc d, float* __restrict__ slab) {
  extern __shared__ float red[];
  const int item = blockIdx.x / d.splits, sp = blockIdx.x - item * d.splits;
  const float* G = d.it[item].G;
  const float* X = d.it[item].X;
  const int nbeg = sp * d.rows_per_unit;
  rows_dw128_body<1>(G, d.ldg, X, d.ldx, X, d.ldx, slab + (long)blockIdx.x * (16384 + 128), nbeg,
                     min(d.rows, nbeg + d.rows_per_unit), red);
}
// out[o][i] = sum_s slab[item][s][o][i], bsum[o] = sum_s slab[item][s][16384 + o]; grid (516, n items)
__global__ __launch_bounds__(256) void rows_dw128_reduce_batch_kernel(const float* __restrict__ slab, DwBatchDesc d) {
  __shared__ float part[8][32];
  const int item = blockIdx.y;
  const int o = threadIdx.x & 31, zg = threadIdx.x >> 5;
  const long total = 16384 + 128;
  const long i = (long)blockIdx.x * 32 + o;
  const float* sl = slab + (long)item * d.splits * total;
  const int per = (d.splits + 7) / 8;
  const int z0 = zg * per, z1 = min(d.splits, z0 + per);
  float s = 0.f;
  if (i < total)
    for (int z = z0; z < z1; ++z) s += sl[(long)z * total + i];
  part[zg][o] = s;
  __syncthreads();
  if (zg != 0 || i >= total) return;
  s = part[0][o];
#pragma unroll
  for (int g = 1; g < 8; ++g) s += part[g][o];
  if (i < 16384) d.it[item].out[(i >> 7) * d.ldo + (i & 127)] = s;
  else if (d.it[item].bsum) d.it[item].bsum[i - 16384] = s;
}

// out_k[o][i] = sum_s slab[s][k][o][i], bsum[o] = sum_s slab[s][NX][o]: the slab loop of an output is spread over
// eight threads whose partial sums are added in fixed order
__global__ __launch_bounds__(256) void rows_dw128_reduce_kernel(const float* __restrict__ slab, int splits, int nx,
                                                                float* __restrict__ out1, long ldo1,
                                                                float* __restrict__ out2, long ldo2,
                                                                float* __restrict__ bsum) {
  __shared__ float part[8][32];
  const int o = threadIdx.x & 31, zg = threadIdx.x >> 5;
  const long total = (long)nx * 16384 + 128, stride = total;
  const long i = (long)blockIdx.x * 32 + o;
  const int per = (splits + 7) / 8;
  const int z0 = zg * per, z1 = min(splits, z0 + per);
  float s = 0.f;
  if (i < total)
    for (int z = z0; z < z1; ++z) s += slab[(long)z * stride + i];
  part[zg][o] = s;
  __syncthreads();
  if (zg != 0 || i >= total) return;
  s = part[0][o];
#pragma unroll
  for (int g = 1; g < 8; ++g) s += part[g][o];
  if (i < 16384) out1[(i >> 7) * ldo1 + (i & 127)] = s;
  else if (nx > 1 && i < 32768) out2[((i - 16384) >> 7) * ldo2 + (i & 127)] = s;
  else if (bsum) bsum[i - (long)nx * 16384] = s;
}

static int dw_rows_per_wg(int rows) {
  int rps = cdiv(cdiv(rows, 256), 16) * 16;      // one workgroup per CU, whole 16-row double batches
  return rps < 16 ? 16 : rps;
}
bool rows_dw128_fast(const float* G, long ldg, const float* X1, long ldx1, const float* X2, long ldx2) {
  static int off = -1;
  if (off < 0) { const char* e = getenv("CGAT_NO_ROWS_DW"); off = (e && e[0] == '1') ? 1 : 0; }
  return !off && (ldg % 4) == 0 && (ldx1 % 4) == 0 && (((uintptr_t)G) & 15) == 0 && (((uintptr_t)X1) & 15) == 0 &&
         (!X2 || ((ldx2 % 4) == 0 && (((uintptr_t)X2) & 15) == 0));
}
size_t rows_dw128_ws_bytes(int rows, int nx) {
  const int splits = cdiv(rows, dw_rows_per_wg(rows));
  return ws_round((size_t)splits * ((size_t)nx * 16384 + 128), 4);
}
// X2/out2 may be null (one right operand); bsum may be null
int rows_dw128_launch(const float* G, long ldg, const float* X1, long ldx1, float* out1, long ldo1, const float* X2,
                      long ldx2, float* out2, long ldo2, float* bsum, int rows, void* ws, size_t ws_bytes,
                      hipStream_t stream) {
  const int nx = X2 ? 2 : 1;
  if (rows <= 0) {
    for (int o = 0; o < 128; ++o) {
      if (hipMemsetAsync(out1 + o * ldo1, 0, 512, stream) != hipSuccess) return CGAT_ERR_HIP;
      if (X2 && hipMemsetAsync(out2 + o * ldo2, 0, 512, stream) != hipSuccess) return CGAT_ERR_HIP;
    }
    if (bsum && hipMemsetAsync(bsum, 0, 512, stream) != hipSuccess) return CGAT_ERR_HIP;
    return CGAT_OK;
  }
  const size_t need = rows_dw128_ws_bytes(rows, nx);
  if (!ws || ws_bytes < need) {
    cgat_set_error("rows_dw128: workspace too small (%zu < %zu)", ws_bytes, need);
    return CGAT_ERR_WORKSPACE;
  }
  const int rps = dw_rows_per_wg(rows), splits = cdiv(rows, rps);
  const size_t lds = ((size_t)nx * 16384 + 128) * sizeof(float);
  {
    CGAT_PROF("rows_dw", stream);
    if (nx == 2) {
      static bool attr = false;
      if (!attr) {
        CGAT_HIP(hipFuncSetAttribute((const void*)rows_dw128_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
      }
      hipLaunchKernelGGL(rows_dw128_kernel<2>, dim3(splits), dim3(512), lds, stream, G, ldg, X1, ldx1, X2, ldx2,
                         (float*)ws, rows, rps);
    } else {
      static bool attr = false;
      if (!attr) {
        CGAT_HIP(hipFuncSetAttribute((const void*)rows_dw128_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
      }
      hipLaunchKernelGGL(rows_dw128_kernel<1>, dim3(splits), dim3(512), lds, stream, G, ldg, X1, ldx1, X1, ldx1,
                         (float*)ws, rows, rps);
    }
    CGAT_LAUNCH_CHECK();
  }
  const long total = (long)nx * 16384 + 128;
  hipLaunchKernelGGL(rows_dw128_reduce_kernel, dim3(cdiv(total, 32)), dim3(256), 0, stream, (const float*)ws, splits, nx,
                     out1, ldo1, out2, ldo2, bsum);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// ---- batched form ----
static int dw_batch_splits(int n_items, int rows) {
  int sp = 256 / (n_items > 0 ? n_items : 1);          // all units resident at once (one workgroup per CU)
  const int most = cdiv(rows, 64);                     // at least 64 rows per unit
  if (sp > most) sp = most;
  return sp < 1 ? 1 : sp;
}
// Slab space for a batch of AT MOST n_items items: a caller that reserves for its full item list and then batches fewer
// (an operand off the 16-byte grid takes the per-layer launch instead) gets more splits per item -- n * (256 / n) is not
// monotonic in n (24 items: 240 units, 23 items: 253) -- so the reservation covers every count up to n_items.
size_t rows_dw128_batch_ws_bytes(int n_items, int rows) {
  if (n_items < 1) n_items = 1;
  const long most = cdiv(rows > 0 ? rows : 1, 64);
  long units = n_items > 256 ? n_items : 256;
  if (units > (long)n_items * most) units = (long)n_items * most;
  return ws_round((size_t)units * (16384 + 128), 4);
}
bool rows_dw128_batch_fast(const DwBatchDesc& d) {
  if (d.n < 1 || d.n > DW_BATCH_MAX || d.rows < 1) return false;
  for (int i = 0; i < d.n; ++i)
    if (!rows_dw128_fast(d.it[i].G, d.ldg, d.it[i].X, d.ldx, nullptr, 0)) return false;
  return true;
}
// d.n, d.rows, d.ldg, d.ldx, d.ldo and the items filled in by the caller; splits / rows_per_unit are set here
int rows_dw128_batch_launch(DwBatchDesc d, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!rows_dw128_batch_fast(d)) {
    cgat_set_error("rows_dw128_batch: not the fast shape (n = %d, rows = %d)", d.n, d.rows);
    return CGAT_ERR_UNSUPPORTED;
  }
  const size_t need = ws_round((size_t)d.n * dw_batch_splits(d.n, d.rows) * (16384 + 128), 4);   // what THIS batch uses
  if (!ws || ws_bytes < need) {
    cgat_set_error("rows_dw128_batch: workspace too small (%zu < %zu)", ws_bytes, need);
    return CGAT_ERR_WORKSPACE;
  }
  d.splits = dw_batch_splits(d.n, d.rows);
  d.rows_per_unit = cdiv(cdiv(d.rows, d.splits), 16) * 16;   // whole 16-row double batches
  // rounding the unit up to 16 rows can leave trailing units that start past the last row (rows = 650, n = 24: unit 9
  // would start at row 720); the kernel's clamped prologue loads would then read past the operands.  Only units that
  // own a row are launched (never more than the bound above, so the workspace check stands).
  d.splits = cdiv(d.rows, d.rows_per_unit);
  const size_t lds = (16384 + 128) * sizeof(float);
  {
    CGAT_PROF("rows_dw", stream);
    static bool attr = false;
    if (!attr) {
      CGAT_HIP(hipFuncSetAttribute((const void*)rows_dw128_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr = true;
    }
    hipLaunchKernelGGL(rows_dw128_batch_kernel, dim3(d.n * d.splits), dim3(512), lds, stream, d, (float*)ws);
    CGAT_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(rows_dw128_reduce_batch_kernel, dim3(cdiv(16384 + 128, 32), d.n), dim3(256), 0, stream,
                     (const float*)ws, d);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}
