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
#include <string.h>

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
// ---- the same products on the bf16 matrix cores (split arithmetic modes) -------------------------------------------
// The f32-input form above is bound by the f32 MFMA rate (64 FLOP/clk/SIMD): 0.73 ms for the 24 products of a
// hypernetwork backward whose operands are 2 GB (0.33 ms of HBM time), and 0.95 ms per 256 -> 128 second layer of the
// vector-attention networks at E = 1M rows (47 ms of the Lightning-default step).  Here every fp32 value is split
// exactly into three bf16 pieces (24 bits, no scales needed: bf16 has fp32's exponent range) and a product is the six
// cross terms of weight >= 2^-24 as v_mfma_f32_16x16x32_bf16 passes with fp32 accumulation -- the arithmetic of the
// "bf16x6" contraction kernels (mfma_bf16.h) -- at 1/16 of the f32-MFMA cycles per flop x 6 passes.
//
// The reduction runs over the ROW index, so both operands need the row on the MFMA k axis while memory has it on the
// slow axis: a K-step is 32 rows; thread (operand, m = column, half) loads its column's 16 rows of the step with
// 4-byte loads (a wave-instruction = 256 contiguous bytes of one row), splits them and writes 8 consecutive k of one
// column as one ds_write_b128 into the image [plane][m][k] (pitch 80 B: conflict-free 16-byte writes and reads); a
// fragment is then one ds_read_b128.  Two images (double buffer), one barrier per K-step, loads two K-steps ahead.
// Accumulator bias of the bf16 MFMA: odd K-steps are accumulated NEGATED into a second accumulator set
// and subtracted at the end, so the sign-independent rounding bias cancels.
#define DWS_PITCH 80                                   // bytes per column of an image: 32 bf16 + 16 B of padding
#define DWS_IMG (128 * DWS_PITCH)                      // one plane of one operand
__global__ __launch_bounds__(512, 1) void rows_dw128_split_batch_kernel(DwBatchDesc d, float* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [buf 2][operand 2][plane 3][128][DWS_PITCH]
  const int item = blockIdx.x / d.splits, sp = blockIdx.x - item * d.splits;
  const float* G = d.it[item].G;
  const float* X = d.it[item].X;
  const int nbeg = sp * d.rows_per_unit, nend = min(d.rows, nbeg + d.rows_per_unit);
  float* o = slab + (long)blockIdx.x * (16384 + 128);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // loader role
  const int opnd = tid >> 8, m = tid & 127, half = (tid >> 7) & 1;
  const float* src = opnd ? X : G;
  const long ld = opnd ? d.ldx : d.ldg;
  unsigned char* wr_base = lds + (size_t)opnd * 3 * DWS_IMG + (size_t)m * DWS_PITCH + half * 32;
  // consumer role: wave (wo, wi) owns output rows 32 wo .. +31 (two 16-blocks) x columns 64 wi .. +63 (four 16-blocks)
  const int wo = wave & 3, wi = wave >> 2;
  const int n16 = lane & 15, kg = lane >> 4;
  const unsigned char* rdA = lds + (size_t)(32 * wo + n16) * DWS_PITCH + kg * 16;
  const unsigned char* rdB = lds + (size_t)3 * DWS_IMG + (size_t)(64 * wi + n16) * DWS_PITCH + kg * 16;

  f32x4 accp[8], accn[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { accp[i] = f32x4{0.f, 0.f, 0.f, 0.f}; accn[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  float cs = 0.f;                                      // G loaders: column sum over this thread's rows
  const int nsteps = (nend - nbeg + 31) / 32;

  float va[16], vb[16];
#define DWS_LOAD(V_, s_)                                                                     \
  {                                                                                          \
    _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                         \
      const int row = nbeg + 32 * (s_) + 16 * half + u;                                      \
      const bool ok = row < nend;                                                            \
      const float t = src[(long)(ok ? row : nend - 1) * ld + m];                             \
      V_[u] = (ok || opnd) ? t : 0.f;                                                        \
    }                                                                                        \
  }
  // one K-step: split this thread's 16 values into the image `buf`, barrier, 48 matrix-core passes into ACC_
#define DWS_STEP(V_, s_, ACC_, NEG_)                                                         \
  {                                                                                          \
    {                                                                                        \
      float w[16];                                                                           \
      _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                       \
        if (!opnd) cs += V_[u];                                                              \
        w[u] = (NEG_ && !opnd) ? -V_[u] : V_[u];                                             \
      }                                                                                      \
      unsigned char* wp = wr_base + (size_t)((s_) & 1) * 6 * DWS_IMG;                        \
      _Pragma("unroll") for (int h8 = 0; h8 < 2; ++h8) {                                     \
        const float v8[8] = {w[8 * h8], w[8 * h8 + 1], w[8 * h8 + 2], w[8 * h8 + 3],         \
                             w[8 * h8 + 4], w[8 * h8 + 5], w[8 * h8 + 6], w[8 * h8 + 7]};    \
        bf16x8 q1, q2, q3;                                                                   \
        split3_x8(v8, q1, q2, q3);                                                           \
        *reinterpret_cast<bf16x8*>(wp + 16 * h8) = q1;                                       \
        *reinterpret_cast<bf16x8*>(wp + DWS_IMG + 16 * h8) = q2;                             \
        *reinterpret_cast<bf16x8*>(wp + 2 * DWS_IMG + 16 * h8) = q3;                         \
      }                                                                                      \
    }                                                                                        \
    DWS_LOAD(V_, (s_) + 2)   /* unconditional (rows past the end read a valid row and count as 0): branch-free */ \
    /* NOT __syncthreads(): its s_waitcnt vmcnt(0) would drain the loads just issued for two K-steps ahead */ \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
    __builtin_amdgcn_s_barrier();                                                            \
    asm volatile("" ::: "memory");                                                           \
    {                                                                                        \
      const unsigned char* pa = rdA + (size_t)((s_) & 1) * 6 * DWS_IMG;                      \
      const unsigned char* pb = rdB + (size_t)((s_) & 1) * 6 * DWS_IMG;                      \
      bf16x8 a1[2], a2[2], a3[2];                                                            \
      _Pragma("unroll") for (int ob = 0; ob < 2; ++ob) {                                     \
        a1[ob] = *reinterpret_cast<const bf16x8*>(pa + (size_t)16 * ob * DWS_PITCH);         \
        a2[ob] = *reinterpret_cast<const bf16x8*>(pa + DWS_IMG + (size_t)16 * ob * DWS_PITCH);      \
        a3[ob] = *reinterpret_cast<const bf16x8*>(pa + 2 * DWS_IMG + (size_t)16 * ob * DWS_PITCH);  \
      }                                                                                      \
      _Pragma("unroll") for (int ib = 0; ib < 4; ++ib) {                                     \
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(pb + (size_t)16 * ib * DWS_PITCH);               \
        const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(pb + DWS_IMG + (size_t)16 * ib * DWS_PITCH);     \
        const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(pb + 2 * DWS_IMG + (size_t)16 * ib * DWS_PITCH); \
        _Pragma("unroll") for (int ob = 0; ob < 2; ++ob) {                                   \
          f32x4 c = ACC_[4 * ob + ib];                                                       \
          c = mma16<false>(a3[ob], b1, c);             /* smallest terms first */            \
          c = mma16<false>(a1[ob], b3, c);                                                   \
          c = mma16<false>(a2[ob], b2, c);                                                   \
          c = mma16<false>(a2[ob], b1, c);                                                   \
          c = mma16<false>(a1[ob], b2, c);                                                   \
          c = mma16<false>(a1[ob], b1, c);                                                   \
          ACC_[4 * ob + ib] = c;                                                             \
        }                                                                                    \
      }                                                                                      \
    }                                                                                        \
  }
  // Branch-free bodies, an even number of K-steps (a step past the end multiplies zeros): with conditional loads the
  // compiler's wait-count bookkeeping turns conservative at every merge and drains the loads that are meant to stay in
  // flight (it did: vmcnt(15) .. (0) where (31) .. (16) was expected)
  DWS_LOAD(va, 0)
  DWS_LOAD(vb, 1)
  for (int s = 0; s < nsteps; s += 2) {
    DWS_STEP(va, s, accp, false)
    DWS_STEP(vb, s + 1, accn, true)
  }
#undef DWS_LOAD
#undef DWS_STEP
  // D layout: lane (n16, kg), register t = out[16 ob + 4 kg + t + 32 wo][16 ib + n16 + 64 wi]
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int orow = 32 * wo + 16 * ob + 4 * kg + t, icol = 64 * wi + 16 * ib + n16;
        float r = accp[4 * ob + ib][t] - accn[4 * ob + ib][t];
        asm volatile("" : "+v"(r));                    // keep the subtraction out of v_pk_* (note in edgez.hip)
        o[orow * 128 + icol] = r;
      }
  // column sums of G: the two row halves of a column through LDS (the images are free after the last barrier + reads)
  __syncthreads();
  float* red = reinterpret_cast<float*>(lds);
  if (!opnd && half == 1) red[m] = cs;
  __syncthreads();
  if (!opnd && half == 0) o[16384 + m] = cs + red[m];
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

int bilinear_mode();
static bool dws_enabled();
int rows_dw128_batch_launch(DwBatchDesc d, void* ws, size_t ws_bytes, hipStream_t stream);
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
size_t rows_dw128_batch_ws_bytes(int n_items, int rows);
size_t rows_dw128_ws_bytes(int rows, int nx) {
  const int splits = cdiv(rows, dw_rows_per_wg(rows));
  size_t b = ws_round((size_t)splits * ((size_t)nx * 16384 + 128), 4);
  // nx == 1 in the split arithmetic modes runs as a batch of one on the bf16 matrix cores (rows_dw128_launch), whose slab
  // count can exceed this kernel's: reserve and check the larger of the two, so that a caller that sized its scratch
  // with this function (Ctx::dw128) gets the fast route, not CGAT_ERR_WORKSPACE from the inner launch
  if (nx == 1) {
    const size_t c = rows_dw128_batch_ws_bytes(1, rows);
    if (c > b) b = c;
  }
  return b;
}
// X2/out2 may be null (one right operand); bsum may be null
int rows_dw128_launch(const float* G, long ldg, const float* X1, long ldx1, float* out1, long ldo1, const float* X2,
                      long ldx2, float* out2, long ldo2, float* bsum, int rows, void* ws, size_t ws_bytes,
                      hipStream_t stream) {
  const int nx = X2 ? 2 : 1;
  if (rows <= 0) {
    for (int o = 0; o < 128; ++o) {
      CGAT_TRY(fill_launch(out1 + o * ldo1, 0.f, 128, stream));
      if (X2) CGAT_TRY(fill_launch(out2 + o * ldo2, 0.f, 128, stream));
    }
    if (bsum) CGAT_TRY(fill_launch(bsum, 0.f, 128, stream));
    return CGAT_OK;
  }
  const size_t need = rows_dw128_ws_bytes(rows, nx);
  if (!ws || ws_bytes < need) {
    cgat_set_error("rows_dw128: workspace too small (%zu < %zu)", ws_bytes, need);
    return CGAT_ERR_WORKSPACE;
  }
  if (nx == 1 && bilinear_mode() != 0 && dws_enabled()) {   // split arithmetic modes: the bf16 matrix-core form, as a batch of one
    DwBatchDesc b;
    memset(&b, 0, sizeof(b));
    b.n = 1; b.rows = rows; b.ldg = ldg; b.ldx = ldx1; b.ldo = ldo1;
    b.it[0] = {G, X1, out1, bsum};
    return rows_dw128_batch_launch(b, ws, ws_bytes, stream);
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
static bool dws_enabled() {                            // CGAT_ROWS_DW_F32=1: the f32-input kernel in every mode (A/B switch)
  static int v = -1;
  if (v < 0) { const char* e = getenv("CGAT_ROWS_DW_F32"); v = (e && e[0] == '1') ? 0 : 1; }
  return v == 1;
}
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
  const bool split_form = bilinear_mode() != 0 && dws_enabled();   // bf16 matrix cores (exact fp32 MFMA in the f32 mode)
  d.splits = dw_batch_splits(d.n, d.rows);
  d.rows_per_unit = cdiv(cdiv(d.rows, d.splits), 32) * 32;   // whole 32-row K-steps (two 16-row double batches)
  // rounding the unit up to 16 rows can leave trailing units that start past the last row (rows = 650, n = 24: unit 9
  // would start at row 720); the kernel's clamped prologue loads would then read past the operands.  Only units that
  // own a row are launched (never more than the bound above, so the workspace check stands).
  d.splits = cdiv(d.rows, d.rows_per_unit);
  const size_t lds = (16384 + 128) * sizeof(float);
  if (split_form) {
    CGAT_PROF("rows_dw", stream);
    const size_t lds2 = (size_t)2 * 2 * 3 * DWS_IMG;
    static bool attr2 = false;
    if (!attr2) {
      CGAT_HIP(hipFuncSetAttribute((const void*)rows_dw128_split_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      attr2 = true;
    }
    hipLaunchKernelGGL(rows_dw128_split_batch_kernel, dim3(d.n * d.splits), dim3(512), lds2, stream, d, (float*)ws);
    CGAT_LAUNCH_CHECK();
  } else {
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
