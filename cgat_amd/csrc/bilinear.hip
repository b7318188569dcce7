// The hypernetwork contraction without ever materialising the predicted weights.
//
// Reference (CGAT/Hypernetworksmp.py:236-254, 205-209): per row n a Linear(C -> C*C + C) emits
// a C x C matrix W_n and bias b_n (66 KB per row at C = 128), then y_n = W_n v_n + b_n.
// With T[o,i,k] = weight[(o*C + i), k] this is the trilinear form
//        y[n,o] = sum_{i,k} T[o,i,k] v[n,i] z[n,k]  (+ bias-row terms handled by plain GEMMs)
// which is a GEMM whose A operand is the row-wise outer product v (x) z, generated on the fly
// in registers: one v_mul per MFMA.  The three backward products have the same shape under a
// permutation of T's indices, so one kernel serves forward, d/dv and d/dz:
//
//   bilinear_rows :  out[n,c]   = init[n,c] + sum_{a,b} p[n,a] q[n,b] T[a,b,c]
//   bilinear_wgrad:  out[a,b,c] = sum_n p[n,a] q[n,b] r[n,c]
//
// Fast path (NB = NC = 128): 128 rows per workgroup, wave w owns rows 32w..32w+31 and all 128
// output columns (4 accumulator blocks); q lives in 64 VGPRs per lane for the whole kernel, p
// is fetched one scalar per 256 MFMAs, T streams through LDS in 16 KB chunks (contiguous
// 512-byte rows, double-buffered).  MFMA-bound by construction: 32 768 v_mfma_f32_32x32x2_f32
// per wave per 128 rows, 2*C^3 flop per row.
#include "common.h"
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// T is expected in the "interleaved" layout made by bilinear_prepare_T: row (a,b) holds its 128
// output columns as [r = c % 32][cb = c / 32], so that lane r fetches the B operands of its four
// accumulator blocks with ONE ds_read_b128.
// grid = asplit * tiles: workgroup (s, tile) covers a in [s*NA/asplit, (s+1)*NA/asplit) and writes
// a partial slab when asplit > 1 (summed in fixed order by slab_sum_rows_kernel).
// JS = j-steps per LDS chunk (a chunk is 2*JS rows of T = JS KB), FLUSH = number of consecutive
// `a` whose products share one partial accumulator (two-level summation, see below).
template <int JS, int FLUSH, int ABL = 0>  // ABL: timing-only ablations (1 no barrier, 2 no global loads, 3 both): wrong results
__global__ __launch_bounds__(256, 1) void bilinear_rows128_kernel(const float* __restrict__ p, long ldp,
                                                                  const float* __restrict__ q, long ldq,
                                                                  const float* __restrict__ T,
                                                                  const float* __restrict__ init, long ldi,
                                                                  float* __restrict__ out, long ldo, int nrows,
                                                                  int NA, int tiles, int asplit, long slab_stride) {
  constexpr int NCH = 64 / JS;          // chunks per `a`
  constexpr int NP = JS / 4;            // 16-byte pieces per thread per chunk (2*JS rows * 32 pieces / 256 threads)
  __shared__ __attribute__((aligned(16))) float Bs[2][2 * JS * 128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hi = lane >> 5;
  const int split = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const int a_beg = (int)((long)NA * split / asplit), a_end = (int)((long)NA * (split + 1) / asplit);
  const int row0 = tile * 128 + wave * 32;
  const int myrow = row0 + r;
  const long rowc = myrow < nrows ? myrow : nrows - 1;
  if (asplit > 1) {
    out += (long)split * slab_stride;
    if (split > 0) init = nullptr;
  }

  // q[row, 64*hi .. 64*hi+63] stays in registers
  float qreg[64];
  {
    const float4* qp = reinterpret_cast<const float4*>(q + rowc * ldq + 64 * hi);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float4 t = qp[j];
      qreg[4 * j] = t.x; qreg[4 * j + 1] = t.y; qreg[4 * j + 2] = t.z; qreg[4 * j + 3] = t.w;
    }
  }
  f32x16 acc[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      float v = 0.f;
      if (init) {
        int orow = row0 + (t & 3) + 8 * (t >> 2) + 4 * hi;
        if (orow < nrows) v = init[(long)orow * ldi + cb * 32 + r];
      }
      acc[cb][t] = v;
    }

  // chunk (a, jc): T rows  a*128 + 64*kk + JS*jc + jj,  kk in {0,1}, jj < JS  -> LDS row kk*JS + jj.
  // Thread piece i (< NP) covers LDS row f_row + 8*i: kk = (f_row + 8 i) / JS, jj = (f_row + 8 i) % JS.
  const int f_row = tid >> 5, f_cq = tid & 31;
  // named staging registers (an array captured by a lambda is demoted to scratch by hipcc)
  float4 pre0, pre1, pre2, pre3, pre4, pre5, pre6, pre7;
#define BIL_G1(i_, reg_)                                                                                   \
  if constexpr ((i_) < NP) {                                                                               \
    constexpr int kk = (8 * (i_)) / JS;                                                                    \
    reg_ = *reinterpret_cast<const float4*>(tb + ((long)(64 * kk + 8 * (i_) - kk * JS)) * 128);            \
  }
#define BIL_GLOAD(a_, jc_)                                                                                 \
  {                                                                                                        \
    const float* tb = T + ((long)(a_) * 128 + JS * (jc_) + f_row) * 128 + 4 * f_cq;                        \
    BIL_G1(0, pre0) BIL_G1(1, pre1) BIL_G1(2, pre2) BIL_G1(3, pre3)                                        \
    BIL_G1(4, pre4) BIL_G1(5, pre5) BIL_G1(6, pre6) BIL_G1(7, pre7)                                        \
  }
#define BIL_S1(i_, reg_) \
  if constexpr ((i_) < NP) *reinterpret_cast<float4*>(lb + 8 * (i_) * 128) = reg_;
#define BIL_LSTORE(buf_)                                                                                   \
  {                                                                                                        \
    float* lb = &Bs[buf_][f_row * 128 + 4 * f_cq];                                                         \
    BIL_S1(0, pre0) BIL_S1(1, pre1) BIL_S1(2, pre2) BIL_S1(3, pre3)                                        \
    BIL_S1(4, pre4) BIL_S1(5, pre5) BIL_S1(6, pre6) BIL_S1(7, pre7)                                        \
  }
  BIL_GLOAD(a_beg, 0);
  BIL_LSTORE(0);
  float pa = p[rowc * ldp + a_beg];
  __syncthreads();
  // Two-level summation: the 128*FLUSH products of FLUSH consecutive `a` go into fresh accumulators
  // that are then added to the totals -- the error growth of the reference's blocked order
  // (W_n = T z ; y = W_n v), instead of one 16 384-term fp32 chain.
  int buf = 0;
  for (int a2 = a_beg; a2 < a_end; a2 += FLUSH) {
    f32x16 part[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int t = 0; t < 16; ++t) part[cb][t] = 0.f;
    for (int a = a2; a < a2 + FLUSH && a < a_end; ++a) {
      const int an = (a + 1 < a_end) ? a + 1 : a;  // the prefetch after the last chunk re-reads a valid chunk, unused
      float pa_next = p[rowc * ldp + an];
#pragma unroll
      for (int jc = 0; jc < NCH; ++jc) {
        if constexpr (!(ABL & 2)) { if (jc + 1 < NCH) BIL_GLOAD(a, jc + 1) else BIL_GLOAD(an, 0); }
        // B operands of step jj for this lane's four blocks: one 16-byte LDS read, fetched ahead
        const float4* bs = reinterpret_cast<const float4*>(&Bs[buf][(hi * JS) * 128 + 4 * r]);
        float4 bv = bs[0];
#pragma unroll
        for (int jj = 0; jj < JS; ++jj) {
          float4 bn = bv;
          if (jj + 1 < JS) bn = bs[(jj + 1) * 32];
          __builtin_amdgcn_sched_barrier(0);  // keep the next step's LDS read ahead of this step's MFMAs
          const float av = pa * qreg[JS * jc + jj];
          part[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.x, part[0], 0, 0, 0);
          part[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.y, part[1], 0, 0, 0);
          part[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.z, part[2], 0, 0, 0);
          part[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.w, part[3], 0, 0, 0);
          bv = bn;
        }
        if constexpr (!(ABL & 2)) BIL_LSTORE(buf ^ 1);
        if constexpr (!(ABL & 1)) __syncthreads();
        buf ^= 1;
      }
      pa = pa_next;
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] += part[cb];
  }
#undef BIL_G1
#undef BIL_GLOAD
#undef BIL_S1
#undef BIL_LSTORE
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      int orow = row0 + (t & 3) + 8 * (t >> 2) + 4 * hi;
      if (orow < nrows) out[(long)orow * ldo + cb * 32 + r] = acc[cb][t];
    }
}

// out[n, c] = sum_s slab[s][n][c]   (fixed order)
__global__ void slab_sum_rows_kernel(const float* __restrict__ slab, int splits, long slab_stride, int nrows,
                                     float* __restrict__ out, long ldo) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)nrows * 128) return;
  float s = 0.f;
  for (int z = 0; z < splits; ++z) s += slab[(long)z * slab_stride + i];
  out[(i >> 7) * ldo + (i & 127)] = s;
}

// any NA, NB, NC: one thread per output element (used for widths other than 128 and as a
// cross-check of the MFMA kernel in the tests)
__global__ void bilinear_rows_generic_kernel(const float* __restrict__ p, long ldp, const float* __restrict__ q,
                                             long ldq, const float* __restrict__ T, const float* __restrict__ init,
                                             long ldi, float* __restrict__ out, long ldo, int nrows, int NA, int NB,
                                             int NC) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)nrows * NC) return;
  int n = (int)(i / NC), c = (int)(i % NC);
  float s = init ? init[(long)n * ldi + c] : 0.f;
  for (int a = 0; a < NA; ++a) {
    float pa = p[(long)n * ldp + a];
    float t = 0.f;
    for (int b = 0; b < NB; ++b) t = fmaf(q[(long)n * ldq + b], T[((long)a * NB + b) * NC + c], t);
    s = fmaf(pa, t, s);
  }
  out[(long)n * ldo + c] = s;
}

static bool force_generic() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("CGAT_FORCE_GENERIC");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

static bool rows_fast(const float* q, long ldq, int NB, int NC) {
  return NB == 128 && NC == 128 && (ldq % 4) == 0 && (((uintptr_t)q) & 15) == 0 && !force_generic();
}

// how many ways to split the `a` range so that tiles*split fills 256 CUs without a ragged last wave
static int rows_asplit(int nrows) {
  const int tiles = cdiv(nrows, 128);
  int best = 1;
  double best_eff = 0.0;
  for (int sp = 1; sp <= 4; ++sp) {
    double w = (double)tiles * sp / 256.0;
    double eff = w / (double)((long)(w + 0.999999));
    if (eff > best_eff + 0.02) {
      best_eff = eff;
      best = sp;
    }
  }
  return best;
}

bool bilinear_T_interleaved(int NB, int NC) { return NB == 128 && NC == 128 && !force_generic(); }

size_t bilinear_rows_ws_bytes(int nrows, int NA, int NB, int NC) {
  if (!bilinear_T_interleaved(NB, NC)) return 0;
  int sp = rows_asplit(nrows);
  return sp > 1 ? ws_round((size_t)sp * nrows * 128, 4) : 0;
}

// T must come from bilinear_prepare_T (interleaved columns iff bilinear_T_interleaved(NB, NC))
int bilinear_rows_launch(const float* p, long ldp, const float* q, long ldq, const float* T, const float* init,
                         long ldi, float* out, long ldo, int nrows, int NA, int NB, int NC, void* ws, size_t ws_bytes,
                         hipStream_t stream) {
  if (nrows <= 0) return CGAT_OK;
  if (bilinear_T_interleaved(NB, NC)) {
    if (!rows_fast(q, ldq, NB, NC) || (((uintptr_t)T) & 15) != 0) {
      cgat_set_error("bilinear_rows: q and T must be 16-byte aligned with ldq %% 4 == 0 at width 128");
      return CGAT_ERR_ARG;
    }
    const int tiles = cdiv(nrows, 128);
    const int sp = rows_asplit(nrows);
    float* dst = out;
    long dld = ldo, stride = 0;
    if (sp > 1) {
      size_t need = ws_round((size_t)sp * nrows * 128, 4);
      if (!ws || ws_bytes < need) {
        cgat_set_error("bilinear_rows: workspace too small (%zu < %zu)", ws_bytes, need);
        return CGAT_ERR_WORKSPACE;
      }
      dst = (float*)ws;
      dld = 128;
      stride = (long)nrows * 128;
    }
    {
      CGAT_PROF("bilinear_rows", stream);
      static int variant = -1;  // dev knob: CGAT_BIL_VARIANT = <JS><FLUSH>, e.g. 161, 162, 322, 324
      if (variant < 0) {
        const char* ev = getenv("CGAT_BIL_VARIANT");
        variant = ev ? atoi(ev) : 162;
      }
      const char* ev2 = getenv("CGAT_BIL_VARIANT_LIVE");  // re-read on every call (A/B in one process)
      const int v = ev2 ? atoi(ev2) : variant;
#define BIL_LAUNCH(JS_, FL_)                                                                                     \
  hipLaunchKernelGGL((bilinear_rows128_kernel<JS_, FL_>), dim3(tiles * sp), dim3(256), 0, stream, p, ldp, q, ldq, \
                     T, init, ldi, dst, dld, nrows, NA, tiles, sp, stride)
      switch (v) {
        case 161: BIL_LAUNCH(16, 1); break;
        case 901: hipLaunchKernelGGL((bilinear_rows128_kernel<16, 2, 1>), dim3(tiles * sp), dim3(256), 0, stream, p, ldp, q, ldq, T, init, ldi, dst, dld, nrows, NA, tiles, sp, stride); break;
        case 902: hipLaunchKernelGGL((bilinear_rows128_kernel<16, 2, 2>), dim3(tiles * sp), dim3(256), 0, stream, p, ldp, q, ldq, T, init, ldi, dst, dld, nrows, NA, tiles, sp, stride); break;
        case 903: hipLaunchKernelGGL((bilinear_rows128_kernel<16, 2, 3>), dim3(tiles * sp), dim3(256), 0, stream, p, ldp, q, ldq, T, init, ldi, dst, dld, nrows, NA, tiles, sp, stride); break;
        case 164: BIL_LAUNCH(16, 4); break;
        case 321: BIL_LAUNCH(32, 1); break;
        case 322: BIL_LAUNCH(32, 2); break;
        case 324: BIL_LAUNCH(32, 4); break;
        default: BIL_LAUNCH(16, 2); break;
      }
#undef BIL_LAUNCH
    }
    CGAT_LAUNCH_CHECK();
    if (sp > 1) {
      hipLaunchKernelGGL(slab_sum_rows_kernel, dim3(cdiv((long)nrows * 128, 256)), dim3(256), 0, stream,
                         (const float*)ws, sp, stride, nrows, out, ldo);
      CGAT_LAUNCH_CHECK();
    }
  } else {
    CGAT_PROF("bilinear_rows_generic", stream);
    hipLaunchKernelGGL(bilinear_rows_generic_kernel, dim3(cdiv((long)nrows * NC, 256)), dim3(256), 0, stream, p, ldp,
                       q, ldq, T, init, ldi, out, ldo, nrows, NA, NB, NC);
    CGAT_LAUNCH_CHECK();
  }
  return CGAT_OK;
}

// ---------------------------------------------------------------------------------------
// weight gradient: out[a,b,c] = sum_n p[n,a] q[n,b] r[n,c]
// grid (NA, splits): one 128(b) x 128(c) output tile per workgroup over a slice of rows
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void bilinear_wgrad128_kernel(const float* __restrict__ p, long ldp,
                                                                const float* __restrict__ q, long ldq,
                                                                const float* __restrict__ rr, long ldr,
                                                                float* __restrict__ slab, int nrows,
                                                                int rows_per_split, int NA) {
  __shared__ __attribute__((aligned(16))) float qs[2][32 * 128];
  __shared__ __attribute__((aligned(16))) float rs[2][32 * 128];
  __shared__ float ps[2][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hi = lane >> 5;
  const int a = blockIdx.x, z = blockIdx.y;
  const int nbeg = z * rows_per_split;
  const int nend = min(nrows, nbeg + rows_per_split);
  const int wb = (wave >> 1) * 64, wc = (wave & 1) * 64;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;

  // staging registers (named, not an array captured by a lambda: that form went to scratch)
  float4 vq0, vq1, vq2, vq3, vr0, vr1, vr2, vr3;
  float vp = 0.f;
  const int f_n = tid >> 5, f_cq = tid & 31;  // piece i covers chunk row f_n + 8*i
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#define WG_LOAD1(i_, vq_, vr_)                                                  \
  {                                                                             \
    int n = n0_ + f_n + 8 * (i_);                                               \
    if (n < nend) {                                                             \
      vq_ = *reinterpret_cast<const float4*>(q + (long)n * ldq + 4 * f_cq);     \
      vr_ = *reinterpret_cast<const float4*>(rr + (long)n * ldr + 4 * f_cq);    \
    } else {                                                                    \
      vq_ = zero4;                                                              \
      vr_ = zero4;                                                              \
    }                                                                           \
  }
#define WG_GLOAD(n0)                                                            \
  {                                                                             \
    const int n0_ = (n0);                                                       \
    WG_LOAD1(0, vq0, vr0) WG_LOAD1(1, vq1, vr1) WG_LOAD1(2, vq2, vr2) WG_LOAD1(3, vq3, vr3) \
    if (tid < 32) vp = (n0_ + tid < nend) ? p[(long)(n0_ + tid) * ldp + a] : 0.f; \
  }
#define WG_LSTORE(buf)                                                          \
  {                                                                             \
    float* dq = &qs[buf][f_n * 128 + 4 * f_cq];                                 \
    float* dr = &rs[buf][f_n * 128 + 4 * f_cq];                                 \
    *reinterpret_cast<float4*>(dq) = vq0;                                       \
    *reinterpret_cast<float4*>(dq + 8 * 128) = vq1;                             \
    *reinterpret_cast<float4*>(dq + 16 * 128) = vq2;                            \
    *reinterpret_cast<float4*>(dq + 24 * 128) = vq3;                            \
    *reinterpret_cast<float4*>(dr) = vr0;                                       \
    *reinterpret_cast<float4*>(dr + 8 * 128) = vr1;                             \
    *reinterpret_cast<float4*>(dr + 16 * 128) = vr2;                            \
    *reinterpret_cast<float4*>(dr + 24 * 128) = vr3;                            \
    if (tid < 32) ps[buf][tid] = vp;                                            \
  }

  // two-level summation over the (long) row dimension: partial sums of 512 rows
  f32x16 tot[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 16; ++t) tot[i][j][t] = 0.f;
  const int nchunks = (nend - nbeg + 31) / 32;
  if (nchunks > 0) {
    WG_GLOAD(nbeg);
    WG_LSTORE(0);
  }
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int cur = c & 1;
    if (c + 1 < nchunks) WG_GLOAD(nbeg + (c + 1) * 32);
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
    {  // operands of step i+1 are fetched from LDS before the MFMAs of step i issue
      const float* qb = &qs[cur][hi * 128 + wb + r];
      const float* rb = &rs[cur][hi * 128 + wc + r];
      const float* pb = &ps[cur][hi];
      float pv = pb[0], q0 = qb[0], q1 = qb[32], b0 = rb[0], b1 = rb[32];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float pvn = pv, q0n = q0, q1n = q1, b0n = b0, b1n = b1;
        if (i < 15) {
          pvn = pb[2 * (i + 1)];
          q0n = qb[(2 * (i + 1)) * 128];
          q1n = qb[(2 * (i + 1)) * 128 + 32];
          b0n = rb[(2 * (i + 1)) * 128];
          b1n = rb[(2 * (i + 1)) * 128 + 32];
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the next step's LDS reads ahead of this step's MFMAs
        const float a0 = pv * q0, a1 = pv * q1;
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        pv = pvn; q0 = q0n; q1 = q1n; b0 = b0n; b1 = b1n;
      }
    }
    if (c + 1 < nchunks) WG_LSTORE(cur ^ 1);
    __syncthreads();
  }
#undef WG_LOAD1
#undef WG_GLOAD
#undef WG_LSTORE
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] += tot[i][j];
  float* o = slab + ((long)z * NA + a) * 128 * 128;
#pragma unroll
  for (int bi = 0; bi < 2; ++bi)
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      int b = wb + bi * 32 + (t & 3) + 8 * (t >> 2) + 4 * hi;
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) o[(long)b * 128 + wc + bj * 32 + r] = acc[bi][bj][t];
    }
}

__global__ void slab_sum_kernel(const float* __restrict__ slab, int splits, long n, float* __restrict__ out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int z = 0; z < splits; ++z) s += slab[(long)z * n + i];
  out[i] = s;
}

__global__ void bilinear_wgrad_generic_kernel(const float* __restrict__ p, long ldp, const float* __restrict__ q,
                                              long ldq, const float* __restrict__ rr, long ldr,
                                              float* __restrict__ out, int nrows, int NA, int NB, int NC) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)NA * NB * NC) return;
  int c = (int)(i % NC);
  int b = (int)((i / NC) % NB);
  int a = (int)(i / ((long)NC * NB));
  float s = 0.f;
  for (int n = 0; n < nrows; ++n) s = fmaf(p[(long)n * ldp + a] * q[(long)n * ldq + b], rr[(long)n * ldr + c], s);
  out[i] = s;
}

static int wgrad_splits(int nrows, int NA) {
  int s = cdiv(512, NA);                 // aim at >= 2 workgroups per CU
  int maxs = nrows / 256;                // at least 8 chunks of 32 rows per split
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  return s;
}

static bool wgrad_fast(const float* q, long ldq, const float* r, long ldr, int NB, int NC) {
  return NB == 128 && NC == 128 && (ldq % 4) == 0 && (ldr % 4) == 0 && (((uintptr_t)q) & 15) == 0 &&
         (((uintptr_t)r) & 15) == 0 && !force_generic();
}

size_t bilinear_wgrad_ws_bytes(int nrows, int NA, int NB, int NC) {
  if (NB == 128 && NC == 128) return ws_round((size_t)wgrad_splits(nrows, NA) * NA * NB * NC, 4);
  return 0;
}

int bilinear_wgrad_launch(const float* p, long ldp, const float* q, long ldq, const float* r, long ldr, float* out,
                          int nrows, int NA, int NB, int NC, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (wgrad_fast(q, ldq, r, ldr, NB, NC)) {
    int splits = wgrad_splits(nrows, NA);
    size_t need = ws_round((size_t)splits * NA * NB * NC, 4);
    if (!ws || ws_bytes < need) {
      cgat_set_error("bilinear_wgrad: workspace too small (%zu < %zu)", ws_bytes, need);
      return CGAT_ERR_WORKSPACE;
    }
    int rps = cdiv(nrows, splits);
    rps = ((rps + 31) / 32) * 32;
    splits = cdiv(nrows, rps);
    if (splits < 1) splits = 1;
    {
      CGAT_PROF("bilinear_wgrad", stream);
      hipLaunchKernelGGL(bilinear_wgrad128_kernel, dim3(NA, splits), dim3(256), 0, stream, p, ldp, q, ldq, r, ldr,
                         (float*)ws, nrows, rps, NA);
    }
    CGAT_LAUNCH_CHECK();
    long n = (long)NA * NB * NC;
    hipLaunchKernelGGL(slab_sum_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, (const float*)ws, splits, n, out);
    CGAT_LAUNCH_CHECK();
  } else {
    CGAT_PROF("bilinear_wgrad_generic", stream);
    long n = (long)NA * NB * NC;
    hipLaunchKernelGGL(bilinear_wgrad_generic_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, p, ldp, q, ldq, r, ldr,
                       out, nrows, NA, NB, NC);
    CGAT_LAUNCH_CHECK();
  }
  return CGAT_OK;
}

// dst = src with its three indices permuted: dst dims are (n[perm0], n[perm1], n[perm2]).
// interleave != 0 (last dst dim == 128): column c of every dst row is stored at (c % 32) * 4 + c / 32.
__global__ void permute3_kernel(const float* __restrict__ src, float* __restrict__ dst, int n0, int n1, int n2,
                                int perm0, int perm1, int perm2, int interleave) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)n0 * n1 * n2;
  if (i >= total) return;
  int dims[3] = {n0, n1, n2};
  int d1 = dims[perm1], d2 = dims[perm2];
  int zs = (int)(i % d2);              // stored position inside the dst row
  int z = interleave ? ((zs & 3) * 32 + (zs >> 2)) : zs;
  int y = (int)((i / d2) % d1);
  int x = (int)(i / ((long)d2 * d1));
  int idx[3];
  idx[perm0] = x; idx[perm1] = y; idx[perm2] = z;
  dst[i] = src[((long)idx[0] * n1 + idx[1]) * n2 + idx[2]];
}

int permute3_launch(const float* src, float* dst, int n0, int n1, int n2, int perm0, int perm1, int perm2,
                    int interleave, hipStream_t stream) {
  long total = (long)n0 * n1 * n2;
  if (total <= 0) return CGAT_OK;
  hipLaunchKernelGGL(permute3_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, src, dst, n0, n1, n2, perm0, perm1,
                     perm2, interleave);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// The B operand of bilinear_rows for a [n0,n1,n2] tensor viewed with permuted indices.
int bilinear_prepare_T(const float* src, float* dst, int n0, int n1, int n2, int perm0, int perm1, int perm2,
                       hipStream_t stream) {
  int dims[3] = {n0, n1, n2};
  return permute3_launch(src, dst, n0, n1, n2, perm0, perm1, perm2,
                         bilinear_T_interleaved(dims[perm1], dims[perm2]) ? 1 : 0, stream);
}
