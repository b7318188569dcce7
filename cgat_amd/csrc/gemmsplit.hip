// The dense-product engine of gemm.hip on the bf16 matrix cores: the same GemmParams (strides, k-major flags, gathers,
// blocked A, row-wise outer-product operands, split-K slabs, bias / gathered addends / activation / beta epilogue), the
// same 128 x 128 x 32 workgroup tile and XCD-aware tile order, but the fp32 operands are split EXACTLY into three bf16
// planes (x = x1 + x2 + x3, mfma_bf16.h) on their way into LDS and the product runs as the six v_mfma_f32_16x16x32_bf16
// passes whose partial products are >= 2^-24 of the largest (a1b1, a1b2, a2b1, a2b2, a1b3, a3b1; smallest first, from a
// zero accumulator per 32-deep chunk; the chunk sums are added in fp32 by the vector ALU -- see the loop).  bf16 keeps fp32's exponent, so there are no scales to find and no pre-pass over the operands.
//
// Why: the f32-input matrix instruction (v_mfma_f32_32x32x2_f32) has a 157 TFLOP/s roof, the six bf16 passes 2 500 / 6
// = 417.  Everything that is not a width-128 special case runs here -- the G-row networks (output head, Roost, crystal
// pooling), widths other than 128 (per-edge products, hypernetwork contractions as outer-product operands), the bias
// products -- in the split arithmetic modes; the f32 mode keeps gemm_f32_kernel (exact fp32 products).
//
// Workgroup: 4 waves, each 64 x 64 of the tile = 4 x 4 blocks of 16 x 16 (64 accumulator VGPRs).  LDS: one stage of
// [plane][row][32 k] bf16 images (row pitch 80 B: the 16-byte fragment reads of 16 consecutive rows hit 16 distinct
// bank groups), 30 KB per operand; global loads for chunk c + 1 are in flight (registers) while chunk c multiplies, two
// workgroups per CU cover each other's store phases.  A wave keeps the 12 A fragments of a chunk and streams the B
// fragments one 16-column block at a time: 24 ds_read_b128 per 96 MFMAs.
#include "common.h"
#include "kernels.h"
#include "mfma_bf16.h"

#define SBM 128
#define SBN 128
#define SBK 32
#define SPITCH 80                       // bytes per row of a plane image
#define SPLANE (128 * SPITCH)           // bytes per plane
#define SOPER (3 * SPLANE)              // bytes per operand

__device__ __forceinline__ float sg_act(float v, int act) {
  switch (act) {
    case CGAT_ACT_TANH: return tanhf(v);
    case CGAT_ACT_LEAKY: return v > 0.f ? v : 0.01f * v;
    case CGAT_ACT_RELU: return v > 0.f ? v : 0.f;
    default: return v;
  }
}

// One 128 x 32 operand tile as 4 float4 per thread.
//  !KM: element (r, k) at base[row(r) * ld + k]; piece i = row (tid >> 3) + 32 i, k = 4 (tid & 7) .. + 3
//   KM: element (r, k) at base[krow(k) * ld + r]; piece i = k 4 (tid & 7) + i, rows 4 (tid >> 3) .. + 3
//  OUT: row-wise outer product (GemmParams::a_outer / b_outer), see gemm.hip
template <bool KM, bool OUT>
struct SplitLoader {
  const float* base;
  long ld;
  int R, r0;
  const int* rgather;
  const int* kgather;
  bool vec;
  long blk;
  const float* obase;
  long old_;
  int on;
  long roff[4], oroff[4];
  bool fast;
  long coff;
  int ocol;

  __device__ void init(int tid) {
    fast = vec && (r0 + 128 <= R);
    coff = 0;
    ocol = 0;
    if (!KM) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = r0 + (tid >> 3) + 32 * i;
        roff[i] = -1;
        oroff[i] = 0;
        if (r < R) {
          roff[i] = (rgather ? (long)rgather[r] : (long)r) * ld;
          if (OUT) oroff[i] = (long)r * old_;
        }
      }
    } else {
      const int r = r0 + 4 * (tid >> 3);
      coff = blk ? (long)(r >> 7) * blk + (r & 127) : (long)r;
      if (OUT) { ocol = r / on; coff = r - ocol * on; }
    }
  }
  __device__ float outer_at(long row, int j) const {
    const int a = j / on;
    return obase[row * old_ + a] * base[row * ld + (j - a * on)];
  }
  __device__ void load(int tid, int k0, int kend, float4 (&v)[4]) const {
    if (fast && k0 + 32 <= kend) {
      if (!KM) {
        if (OUT) {
          const int kk = k0 + 4 * (tid & 7);
          const int a = kk / on, b = kk - a * on;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 t = *reinterpret_cast<const float4*>(base + roff[i] + b);
            const float s = obase[oroff[i] + a];
            v[i] = make_float4(t.x * s, t.y * s, t.z * s, t.w * s);
          }
        } else {
          const long ko = (blk ? (long)(k0 >> 7) * blk + (k0 & 127) : (long)k0) + 4 * (tid & 7);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4*>(base + roff[i] + ko);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int k = k0 + 4 * (tid & 7) + i;
          const long krow = kgather ? (long)kgather[k] : (long)k;
          v[i] = *reinterpret_cast<const float4*>(base + krow * ld + coff);
          if (OUT) {
            const float s = obase[krow * old_ + ocol];
            v[i] = make_float4(v[i].x * s, v[i].y * s, v[i].z * s, v[i].w * s);
          }
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t[4] = {0.f, 0.f, 0.f, 0.f};
      if (!KM) {
        const int k = k0 + 4 * (tid & 7);
        if (roff[i] >= 0) {
          if (OUT) {
            const long row = r0 + (tid >> 3) + 32 * i;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (k + j < kend) t[j] = outer_at(row, k + j);
          } else if (k < kend) {
            const float* p = base + roff[i] + (blk ? (long)(k >> 7) * blk + (k & 127) : (long)k);
            if (vec && k + 3 < kend) {
              const float4 q = *reinterpret_cast<const float4*>(p);
              t[0] = q.x; t[1] = q.y; t[2] = q.z; t[3] = q.w;
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (k + j < kend) t[j] = p[j];
            }
          }
        }
      } else {
        const int k = k0 + 4 * (tid & 7) + i;
        const int r = r0 + 4 * (tid >> 3);
        if (k < kend && r < R) {
          if (OUT) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (r + j < R) t[j] = outer_at(k, r + j);
          } else {
            const long krow = kgather ? (long)kgather[k] : (long)k;
            const float* p = base + krow * ld + (blk ? (long)(r >> 7) * blk + (r & 127) : (long)r);
            if (vec && r + 3 < R) {
              const float4 q = *reinterpret_cast<const float4*>(p);
              t[0] = q.x; t[1] = q.y; t[2] = q.z; t[3] = q.w;
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (r + j < R) t[j] = p[j];
            }
          }
        }
      }
      v[i] = make_float4(t[0], t[1], t[2], t[3]);
    }
  }
  // four consecutive k of one row -> 8 bytes in each of the three planes
  // sx = 0 or 0x80008000: flips the sign of every stored bf16 (exact: the planes of -x are the negated planes of x)
  __device__ static void put(char* lds, int row, int kq, float a, float b, float c, float d, unsigned sx) {
    unsigned w1a, w2a, w3a, w1b, w2b, w3b;
    split3_pair(a, b, w1a, w2a, w3a);
    split3_pair(c, d, w1b, w2b, w3b);
    char* p = lds + row * SPITCH + kq * 8;
    *reinterpret_cast<uint2*>(p) = make_uint2(w1a ^ sx, w1b ^ sx);
    *reinterpret_cast<uint2*>(p + SPLANE) = make_uint2(w2a ^ sx, w2b ^ sx);
    *reinterpret_cast<uint2*>(p + 2 * SPLANE) = make_uint2(w3a ^ sx, w3b ^ sx);
  }
  __device__ void store(int tid, char* lds, const float4 (&v)[4], unsigned sx = 0u) const {
    if (!KM) {
#pragma unroll
      for (int i = 0; i < 4; ++i) put(lds, (tid >> 3) + 32 * i, tid & 7, v[i].x, v[i].y, v[i].z, v[i].w, sx);
    } else {
      const int r = 4 * (tid >> 3), kq = tid & 7;
      put(lds, r + 0, kq, v[0].x, v[1].x, v[2].x, v[3].x, sx);
      put(lds, r + 1, kq, v[0].y, v[1].y, v[2].y, v[3].y, sx);
      put(lds, r + 2, kq, v[0].z, v[1].z, v[2].z, v[3].z, sx);
      put(lds, r + 3, kq, v[0].w, v[1].w, v[2].w, v[3].w, sx);
    }
  }
};

template <bool AKM, bool BKM, int OUTER>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(GemmParams p) {
  constexpr int PC = 132;
  __shared__ __attribute__((aligned(16))) char lds_all[(2 * SOPER) > SBM * PC * 4 ? (2 * SOPER) : SBM * PC * 4];
  char* As = lds_all;
  char* Bs = lds_all + SOPER;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kg = lane >> 4;
  // tile order: as gemm_f32_kernel (gemm.hip)
  const int tiles_n = (p.N + SBN - 1) / SBN, tiles_m = (p.M + SBM - 1) / SBM;
  int tile_m, tile_n, z;
  {
    const bool swap = p.splits <= 1 && tiles_m < 8 && tiles_n > tiles_m;
    const int inner = p.splits > 1 ? tiles_m * tiles_n : (swap ? tiles_m : tiles_n);
    const int outer = p.splits > 1 ? p.splits : (swap ? tiles_n : tiles_m);
    const int L = blockIdx.x;
    int o, i;
    if (outer >= 8) {
      const int j = L >> 3;
      o = (L & 7) + 8 * (j / inner);
      i = j % inner;
    } else {
      o = L % outer;
      i = L / outer;
    }
    if (o >= outer) return;
    if (p.splits > 1) { z = o; tile_m = i / tiles_n; tile_n = i % tiles_n; }
    else { z = 0; tile_m = swap ? i : o; tile_n = swap ? o : i; }
  }
  const int m0 = tile_m * SBM, n0 = tile_n * SBN;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int kbeg = z * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);

  SplitLoader<AKM, OUTER == 1> la{p.A, p.lda, p.M, m0, p.a_rgather, nullptr, p.a_vec != 0, p.a_block, p.a_outer, p.ld_a_outer,
                                 p.outer_n, {0, 0, 0, 0}, {0, 0, 0, 0}, false, 0, 0};
  SplitLoader<BKM, OUTER == 2> lb{p.B, p.ldb, p.N, n0, nullptr, p.b_kgather, p.b_vec != 0, 0, p.b_outer, p.ld_b_outer,
                                 p.outer_n, {0, 0, 0, 0}, {0, 0, 0, 0}, false, 0, 0};
  la.init(tid);
  lb.init(tid);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float4 ra[4], rb[4];
  const int nchunks = (kend - kbeg + SBK - 1) / SBK;
  const int rot = (p.splits > 1 && nchunks > 1) ? (int)(((unsigned)z * 37u + (unsigned)tile_m * 11u + (unsigned)tile_n * 5u) % (unsigned)nchunks) : 0;
#define SCHUNK_K(c_) (kbeg + (((c_) + rot) >= nchunks ? ((c_) + rot - nchunks) : ((c_) + rot)) * SBK)
  if (nchunks > 0) {
    la.load(tid, SCHUNK_K(0), kend, ra);
    lb.load(tid, SCHUNK_K(0), kend, rb);
  }
  const char* ap = As + (wm + i16) * SPITCH + kg * 16;
  const char* bp = Bs + (wn + i16) * SPITCH + kg * 16;
  // The bf16 matrix instruction's fp32 accumulator rounds with a sign-independent bias (towards -inf-like, ~ -5e-11 of
  // the running sum per accumulation: DESIGN.md §2, tools/bf16x3_probe.hip), coherent over all outputs of a long
  // reduction: -1.9e-6 of the result at K = 83 340 when the six passes of every chunk accumulated straight into the
  // running sum (tools/gemm_engine_probe.py).  So the six passes of a chunk accumulate from ZERO (the instruction only
  // ever sees a 32-deep partial sum) and the chunk's sum is added to the running sum by the vector ALU, which rounds
  // to nearest: one rounding at full magnitude per chunk instead of six (64 v_add_f32 per 96 passes).  What is left is
  // the bias INSIDE a chunk's sum, proportional to the magnitude of its terms whatever their signs: -1.7e-9 of
  // sum |a||b| on every output (tools/gemm_bias_probe.py: 300 x the statistical expectation of the mean, i.e. coherent
  // -- a weighted sum of many outputs, such as the gradient of a scale parameter, collects it).  Odd chunks therefore
  // multiply -A (stored negated: exact) and their sum is SUBTRACTED: the chunk biases alternate in sign and cancel.
  for (int c = 0; c < nchunks; ++c) {
    la.store(tid, As, ra, (c & 1) ? 0x80008000u : 0u);
    lb.store(tid, Bs, rb);
    __syncthreads();
    if (c + 1 < nchunks) {
      la.load(tid, SCHUNK_K(c + 1), kend, ra);
      lb.load(tid, SCHUNK_K(c + 1), kend, rb);
    }
    const float sgf = (c & 1) ? -1.f : 1.f;
    bf16x8 a1[4], a2[4], a3[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a1[r] = *reinterpret_cast<const bf16x8*>(ap + r * 16 * SPITCH);
      a2[r] = *reinterpret_cast<const bf16x8*>(ap + r * 16 * SPITCH + SPLANE);
      a3[r] = *reinterpret_cast<const bf16x8*>(ap + r * 16 * SPITCH + 2 * SPLANE);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(bp + cb * 16 * SPITCH);
      const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(bp + cb * 16 * SPITCH + SPLANE);
      const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(bp + cb * 16 * SPITCH + 2 * SPLANE);
      // first operand = the B-tile fragment: the lane then holds C[m = i16][n = 4 kg .. 4 kg + 3] (16-byte epilogue pieces)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
        t = mma16<false>(b1, a3[r], t);
        t = mma16<false>(b3, a1[r], t);
        t = mma16<false>(b2, a2[r], t);
        t = mma16<false>(b1, a2[r], t);
        t = mma16<false>(b2, a1[r], t);
        t = mma16<false>(b1, a1[r], t);
        acc[r][cb] += t * sgf;                      // (one fused multiply-add per element: +-1 * t is exact)
      }
    }
    __syncthreads();
  }
#undef SCHUNK_K
  // ---- epilogue: the accumulator tile through LDS, then row pieces (16-byte where legal) ----
  float* Cbase = p.C;
  const bool slab = p.splits > 1;
  if (slab) Cbase = p.slab + (long)z * p.M * p.N;
  const long ldc = slab ? (long)p.N : p.ldc;
  float* Cs = reinterpret_cast<float*>(lds_all);
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const f32x4 t = acc[r][cb];
      *reinterpret_cast<float4*>(&Cs[(wm + 16 * r + i16) * PC + wn + 16 * cb + 4 * kg]) = make_float4(t[0], t[1], t[2], t[3]);
    }
  __syncthreads();
  if (p.c_vec) {
    const int c4 = tid & 31;
    const int n = n0 + 4 * c4;
    const bool ncol = n < p.N;          // N % 4 == 0 on this path
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!slab && p.bias && ncol) bias4 = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int row = (tid >> 5) + 8 * i;
      const int m = m0 + row;
      if (m >= p.M || !ncol) continue;
      float4 v = *reinterpret_cast<const float4*>(&Cs[row * PC + 4 * c4]);
      long crow = m;
      if (!slab) {
        v.x = v.x * p.alpha + bias4.x; v.y = v.y * p.alpha + bias4.y;
        v.z = v.z * p.alpha + bias4.z; v.w = v.w * p.alpha + bias4.w;
        if (p.c_scatter) crow = p.c_scatter[m];
        if (p.add1) {
          const float4 g = *reinterpret_cast<const float4*>(p.add1 + (long)p.add1_idx[m] * p.ld_add + n);
          v.x += g.x; v.y += g.y; v.z += g.z; v.w += g.w;
        }
        if (p.add2) {
          const float4 g = *reinterpret_cast<const float4*>(p.add2 + (long)p.add2_idx[m] * p.ld_add + n);
          v.x += g.x; v.y += g.y; v.z += g.z; v.w += g.w;
        }
        v.x = sg_act(v.x, p.act); v.y = sg_act(v.y, p.act);
        v.z = sg_act(v.z, p.act); v.w = sg_act(v.w, p.act);
        if (p.beta != 0.f) {
          const float4 c0 = *reinterpret_cast<const float4*>(Cbase + crow * ldc + n);
          v.x += p.beta * c0.x; v.y += p.beta * c0.y; v.z += p.beta * c0.z; v.w += p.beta * c0.w;
        }
      }
      *reinterpret_cast<float4*>(Cbase + crow * ldc + n) = v;
    }
    return;
  }
  // element by element (N % 4 != 0 or unaligned C / addends)
  for (int idx = tid; idx < SBM * SBN; idx += 256) {
    const int row = idx >> 7, col = idx & 127;
    const int m = m0 + row, n = n0 + col;
    if (m >= p.M || n >= p.N) continue;
    float v = Cs[row * PC + col];
    long crow = m;
    if (!slab) {
      v = v * p.alpha + (p.bias ? p.bias[n] : 0.f);
      if (p.c_scatter) crow = p.c_scatter[m];
      if (p.add1) v += p.add1[(long)p.add1_idx[m] * p.ld_add + n];
      if (p.add2) v += p.add2[(long)p.add2_idx[m] * p.ld_add + n];
      v = sg_act(v, p.act);
      if (p.beta != 0.f) v += p.beta * Cbase[crow * ldc + n];
    }
    Cbase[crow * ldc + n] = v;
  }
}

// called by gemm_launch (gemm.hip) with every derived field of p filled in and the grid it computed
int gemm_split_launch(const GemmParams& p, unsigned grid, hipStream_t stream) {
  CGAT_PROF("gemm_split", stream);
  if (p.a_outer) hipLaunchKernelGGL((gemm_split_kernel<false, true, 1>), dim3(grid), dim3(256), 0, stream, p);
  else if (p.b_outer) hipLaunchKernelGGL((gemm_split_kernel<true, true, 2>), dim3(grid), dim3(256), 0, stream, p);
  else if (!p.a_kmajor && !p.b_kmajor) hipLaunchKernelGGL((gemm_split_kernel<false, false, 0>), dim3(grid), dim3(256), 0, stream, p);
  else if (!p.a_kmajor && p.b_kmajor) hipLaunchKernelGGL((gemm_split_kernel<false, true, 0>), dim3(grid), dim3(256), 0, stream, p);
  else if (p.a_kmajor && !p.b_kmajor) hipLaunchKernelGGL((gemm_split_kernel<true, false, 0>), dim3(grid), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((gemm_split_kernel<true, true, 0>), dim3(grid), dim3(256), 0, stream, p);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}
