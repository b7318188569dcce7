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
// OUT: the operand is a row-wise outer product (GemmParams::a_outer / b_outer): the element at flat index j of a row is
// obase[row * old_ + j / on] * base[row * ld + j % on]; no gather, no blocked layout in that form.
template <bool KM, bool OUT = false>
struct TileLoader {
  const float* base;
  long ld;
  int R;             // extent of the tile's row dimension (M or N)
  int r0;            // first row of this tile
  const int* rgather;  // !KM: optional row index
  const int* kgather;  // KM: optional k index
  bool vec;          // 16-byte loads legal
  long blk;          // 128-wide column-block stride (0 = plain layout)
  long roff[4];      // !KM: precomputed row offsets (elements), -1 = out of range
  bool fast;         // the whole 128-row tile is in range and 16-byte loads are legal: no per-piece predicates
  long coff;         // KM: precomputed offset of this thread's 4 columns
  const float* obase;  // OUT: the outer factor, its row stride and the width of the inner factor
  long old_;
  int on;
  long oroff[4];     // OUT, !KM: row offsets into obase
  int ocol;          // OUT, KM: this thread's column of obase (coff / on; coff then holds coff % on)

  __device__ void init(int tid) {
    fast = vec && (r0 + 128 <= R);
    coff = 0;
    ocol = 0;
    if (!KM) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int f = tid + 256 * i;
        int r = r0 + (f >> 3);
        if (r < R) {
          long row = rgather ? (long)rgather[r] : (long)r;
          roff[i] = row * ld;
          if (OUT) oroff[i] = (long)r * old_;
        } else {
          roff[i] = -1;
          if (OUT) oroff[i] = 0;
        }
      }
    } else {
      const int r = r0 + 4 * (tid & 31);
      coff = blk ? (long)(r >> 7) * blk + (r & 127) : (long)r;
      if (OUT) { ocol = r / on; coff = r - ocol * on; }
    }
  }

  // one element of the outer-product operand: flat index j of row `row` (KM: `row` is the k index)
  __device__ float outer_at(long row, int j) const {
    const int a = j / on;
    return obase[row * old_ + a] * base[row * ld + (j - a * on)];
  }

  __device__ void load(int tid, int k0, int kend, float4 (&v)[4]) const {
    if (fast && k0 + 32 <= kend) {  // wave-uniform: interior tile and chunk, straight 16-byte loads
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
          return;
        }
        const long ko = (blk ? (long)(k0 >> 7) * blk + (k0 & 127) : (long)k0) + 4 * (tid & 7);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4*>(base + roff[i] + ko);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int k = k0 + (tid >> 5) + 8 * i;
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
    if (OUT) {   // edges of the outer-product form: element by element
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i;
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        if (!KM) {
          const int k = k0 + 4 * (f & 7);
          if (roff[i] >= 0) {
            const long row = r0 + (f >> 3);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (k + j < kend) t[j] = outer_at(row, k + j);
          }
        } else {
          const int k = k0 + (f >> 5);
          const int r = r0 + 4 * (f & 31);
          if (k < kend) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (r + j < R) t[j] = outer_at(k, r + j);
          }
        }
        v[i] = make_float4(t[0], t[1], t[2], t[3]);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int f = tid + 256 * i;
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!KM) {
        int k = k0 + 4 * (f & 7);
        if (roff[i] >= 0 && k < kend) {
          const float* p = base + roff[i] + (blk ? (long)(k >> 7) * blk + (k & 127) : (long)k);
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
          const float* p = base + krow * ld + (blk ? (long)(r >> 7) * blk + (r & 127) : (long)r);
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

// ABL: timing-only ablations (1 no barrier, 2 no global loads, 4 no LDS stores); OUTER: 1 = A, 2 = B is an outer product
template <bool AKM, bool BKM, int ABL = 0, int OUTER = 0>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmParams p) {
  constexpr int PA = AKM ? 132 : 129;
  constexpr int PB = BKM ? 132 : 129;
  // one LDS block: [As0 | As1 | Bs0 | Bs1]; reused as the 128 x 132 C tile by the vector epilogue
  __shared__ __attribute__((aligned(16))) float lds_all[(2 * BK * PA + 2 * BK * PB) > BM * 132 ? (2 * BK * PA + 2 * BK * PB) : BM * 132];
  float(*As)[BK * PA] = reinterpret_cast<float(*)[BK * PA]>(lds_all);
  float(*Bs)[BK * PB] = reinterpret_cast<float(*)[BK * PB]>(lds_all + 2 * BK * PA);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hi = lane >> 5;
  // XCD-aware work mapping.  Workgroups are dealt round-robin over the 8 XCDs (each with its own
  // L2), so linear ids L and L+8 share an L2.  The "outer" index (K-split for split-K products,
  // row tile otherwise) is striped over the XCDs and the tiles that share its operands ("inner")
  // run back to back on ONE XCD: the shared operand tile is fetched into one L2 once instead
  // of into up to eight.  Pure placement: any mapping gives the same results.
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  int tile_m, tile_n, z;
  {
    // (fewer than 8 row tiles: the column tiles are striped instead -- with two row tiles and 512 column tiles, the
    // weight gradient of a width-256 contraction, the row-striped order put all the work on two of the eight XCDs)
    const bool swap = p.splits <= 1 && tiles_m < 8 && tiles_n > tiles_m;
    const int inner = p.splits > 1 ? tiles_m * tiles_n : (swap ? tiles_m : tiles_n);
    const int outer = p.splits > 1 ? p.splits : (swap ? tiles_n : tiles_m);
    const int L = blockIdx.x;
    int o, i;
    if (outer >= 8) {
      const int j = L >> 3;
      o = (L & 7) + 8 * (j / inner);
      i = j % inner;
    } else {   // fewer outer indices than XCDs (2..7 K-splits, tiny products): plain order, every XCD gets work
      o = L % outer;
      i = L / outer;
    }
    if (o >= outer) return;
    if (p.splits > 1) {
      z = o;
      tile_m = i / tiles_n;
      tile_n = i % tiles_n;
    } else {
      z = 0;
      tile_m = swap ? i : o;
      tile_n = swap ? o : i;
    }
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  // K range of this split
  const int kbeg = z * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);

  TileLoader<AKM, OUTER == 1> la{p.A, p.lda, p.M, m0, p.a_rgather, nullptr, p.a_vec != 0, p.a_block, {0, 0, 0, 0}, false, 0,
                                p.a_outer, p.ld_a_outer, p.outer_n, {0, 0, 0, 0}, 0};
  TileLoader<BKM, OUTER == 2> lb{p.B, p.ldb, p.N, n0, nullptr, p.b_kgather, p.b_vec != 0, 0, {0, 0, 0, 0}, false, 0,
                                p.b_outer, p.ld_b_outer, p.outer_n, {0, 0, 0, 0}, 0};
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
  // Split-K streams start at offsets that are multiples of large powers of two and advance in
  // lockstep, i.e. they all sit on the same HBM channels at the same time.  Each workgroup
  // therefore walks its K range from a different (fixed, id-derived) starting chunk, wrapping
  // around: same products, deterministic order, streams spread over the channels.
  const int rot = (p.splits > 1 && nchunks > 1) ? (int)(((unsigned)z * 37u + (unsigned)tile_m * 11u + (unsigned)tile_n * 5u) % (unsigned)nchunks) : 0;
#define CHUNK_K(c_) (kbeg + (((c_) + rot) >= nchunks ? ((c_) + rot - nchunks) : ((c_) + rot)) * BK)
  if (nchunks > 0) {
    la.load(tid, CHUNK_K(0), kend, ra);
    lb.load(tid, CHUNK_K(0), kend, rb);
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
    if constexpr (!(ABL & 2)) {
      if (c + 1 < nchunks) {
        la.load(tid, CHUNK_K(c + 1), kend, ra);
        lb.load(tid, CHUNK_K(c + 1), kend, rb);
      }
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
    if constexpr (!(ABL & 4)) {
      if (c + 1 < nchunks) {
        la.template store<PA>(tid, As[cur ^ 1], ra);
        lb.template store<PB>(tid, Bs[cur ^ 1], rb);
      }
    }
    if constexpr (!(ABL & 1)) __syncthreads();
  }

#undef CHUNK_K
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] += tot[i][j];
  // ---- epilogue ----
  float* Cbase = p.C;
  const bool slab = p.splits > 1;
  if (slab) Cbase = p.slab + (long)z * p.M * p.N;
  const long ldc = slab ? (long)p.N : p.ldc;

  if (p.c_vec) {
    // Vector epilogue: the 128 x 128 accumulator tile goes through LDS (the operand buffers are
    // free now) so that every global access of the epilogue -- the store, the gathered addends,
    // the previous C -- is a 16-byte piece of a 512-byte row segment instead of one dword per
    // lane (the dword form is store-issue-bound: 62 -> 29 TFLOP/s on the per-edge product).
    constexpr int PC = 132;
    float* Cs = lds_all;
    __syncthreads();
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int bj = 0; bj < 2; ++bj)
#pragma unroll
        for (int t = 0; t < 16; ++t)
          Cs[(wm + bi * 32 + (t & 3) + 8 * (t >> 2) + 4 * hi) * PC + wn + bj * 32 + r] = acc[bi][bj][t];
    __syncthreads();
    const int c4 = tid & 31;            // 16-byte column piece
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
        v.x = act_apply(v.x, p.act); v.y = act_apply(v.y, p.act);
        v.z = act_apply(v.z, p.act); v.w = act_apply(v.w, p.act);
        if (p.beta != 0.f) {
          const float4 c0 = *reinterpret_cast<const float4*>(Cbase + crow * ldc + n);
          v.x += p.beta * c0.x; v.y += p.beta * c0.y; v.z += p.beta * c0.z; v.w += p.beta * c0.w;
        }
      }
      *reinterpret_cast<float4*>(Cbase + crow * ldc + n) = v;
    }
    return;
  }

  // Scalar epilogue (N % 4 != 0 or unaligned C / addends): one dword per lane.
  const int nA = n0 + wn + r, nB = nA + 32;
  const bool okA = nA < p.N, okB = nB < p.N;
  const float biasA = (!slab && p.bias && okA) ? p.bias[nA] : 0.f;
  const float biasB = (!slab && p.bias && okB) ? p.bias[nB] : 0.f;
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
        const int n = bj ? nB : nA;
        if (!(bj ? okB : okA)) continue;
        float v = acc[bi][bj][t];
        if (!slab) {
          v = v * p.alpha + (bj ? biasB : biasA);
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
// Fixed summation tree (deterministic): the z range is cut into ZG contiguous groups summed in order by ZG threads
// per output, whose partial sums are then added in group order.  With hundreds of slabs and a few thousand outputs
// (the [128,128] weight gradients) one thread per output is a 300-long chain of dependent loads.
template <int ZG>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, int splits, int M, int N,
                                                            float* __restrict__ C, long ldc, float alpha, float beta,
                                                            const float* __restrict__ bias, int act) {
  constexpr int OUT = 256 / ZG;                  // outputs per workgroup
  __shared__ float part[ZG][OUT];
  const int o = threadIdx.x % OUT, zg = threadIdx.x / OUT;
  const long total = (long)M * N;
  const long i = (long)blockIdx.x * OUT + o;
  const int per = (splits + ZG - 1) / ZG;
  const int z0 = zg * per, z1 = min(splits, z0 + per);
  float s = 0.f;
  if (i < total)
    for (int z = z0; z < z1; ++z) s += slab[(long)z * total + i];
  if (ZG > 1) {
    part[zg][o] = s;
    __syncthreads();
    if (zg != 0) return;
    s = part[0][o];
#pragma unroll
    for (int g = 1; g < ZG; ++g) s += part[g][o];
  }
  if (i >= total) return;
  const int m = (int)(i / N), n = (int)(i % N);
  s *= alpha;
  if (bias) s += bias[n];
  s = act_apply(s, act);
  float* d = C + (long)m * ldc + n;
  if (beta != 0.f) s += beta * *d;
  *d = s;
}

int splitk_reduce_launch(const float* slab, int splits, int M, int N, float* C, long ldc, hipStream_t stream, float alpha,
                         float beta, const float* bias, int act) {
  const long total = (long)M * N;
  if (total <= 0) return CGAT_OK;
  if (total < 256 * 1024 && splits >= 32)   // few outputs, many slabs: spread the slab loop over 8 threads
    hipLaunchKernelGGL(splitk_reduce_kernel<8>, dim3(cdiv(total, 32)), dim3(256), 0, stream, slab, splits, M, N, C, ldc,
                       alpha, beta, bias, act);
  else
    hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3(cdiv(total, 256)), dim3(256), 0, stream, slab, splits, M, N, C, ldc,
                       alpha, beta, bias, act);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

size_t gemm_ws_bytes(const GemmParams& p) { return p.splits > 1 ? ws_round((size_t)p.splits * p.M * p.N, 4) : 0; }

// Chooses a split count for reductions with a long K and few output tiles.  Workgroups are
// striped over the 8 XCDs (see the kernel) and an XCD holds 32 CUs x 2 resident workgroups = 64
// slots, so the choice is the q (splits = 8q) whose q * tiles workgroups per XCD fill whole
// rounds of 64 best: 43 splits x 12 tiles left 8 workgroups per XCD for a second round and ran
// 1.7x slower than 128 splits (3 exact rounds).
int gemm_pick_splits(int M, int N, int K) {
  const long tiles = (long)cdiv(M, BM) * cdiv(N, BN);
  if (tiles >= 256 || K < 4 * BK * 8) return 1;
  const long qmax = K / (8L * BK * 8);  // at least 8 chunks per split
  if (qmax < 1) return K >= 2 * BK * 8 ? (int)(K / (BK * 8)) : 1;
  int best_q = 1;
  double best = -1.0;
  for (long q = 1; q <= qmax && q <= 64; ++q) {
    const long per_xcd = q * tiles;
    const long rounds = (per_xcd + 63) / 64;
    double eff = (double)per_xcd / (64.0 * rounds);
    if (rounds > 4) eff -= 0.02 * (rounds - 4);  // slab traffic grows with the split count
    if (eff > best + 1e-9) {
      best = eff;
      best_q = (int)q;
    }
  }
  return 8 * best_q;
}

// Few output tiles and a long reduction -- the G-row networks at the harness' shipped batch of 64 crystals: M = 64 rows
// against 1024 x 1024 weights = 8 workgroups, each a 32-step dependent load -> LDS -> MFMA chain on an otherwise empty
// chip (130 us).  K is cut so that every workgroup runs two chunks and the partial tiles are summed in split order by
// splitk_reduce (which also applies bias / activation / beta): same result whatever else runs, a different -- fixed --
// summation order than the unsplit product.
int gemm_pick_splits_skinny(int M, int N, int K) {
  const long tiles = (long)cdiv(M, BM) * cdiv(N, BN);
  if (tiles > 32 || K < 512) return 1;
  int s = K / (2 * BK);
  while (s > 1 && tiles * s > 512) s >>= 1;
  return s < 2 ? 1 : s;
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
  if (p.a_block) {
    CGAT_CHECK_ARG(p.lda == 128 && (p.a_block % 4) == 0, "gemm: blocked A needs lda == 128");
  }
  p.a_vec = aligned(p.A, p.lda) ? 1 : 0;
  p.b_vec = aligned(p.B, p.ldb) ? 1 : 0;
  if (p.a_outer || p.b_outer) {
    CGAT_CHECK_ARG(p.outer_n > 0 && !(p.a_outer && p.b_outer) && !p.a_rgather && !p.a_block && !p.b_kgather && p.b_kmajor &&
                       (p.a_outer ? !p.a_kmajor : p.a_kmajor),
                   "gemm: outer-product operands come as (A outer, row-major; B k-major) or (A k-major; B outer, k-major)");
    // a 16-byte piece must stay inside one block of the inner factor
    if (p.a_outer && p.outer_n % 4 != 0) p.a_vec = 0;
    if (p.b_outer && p.outer_n % 4 != 0) p.b_vec = 0;
  }
  {
    bool cv = (p.N % 4) == 0;
    if (p.splits > 1) cv = cv && ((((uintptr_t)p.slab) & 15) == 0);
    else {
      cv = cv && aligned(p.C, p.ldc);
      if (p.bias) cv = cv && ((((uintptr_t)p.bias) & 15) == 0);
      if (p.add1) cv = cv && aligned(p.add1, p.ld_add);
      if (p.add2) cv = cv && aligned(p.add2, p.ld_add);
    }
    p.c_vec = cv ? 1 : 0;
  }
  // 1-D grid in XCD-striped order (see the kernel): 8 * inner * ceil(outer / 8) workgroups
  const int tiles_m = cdiv(p.M, BM), tiles_n = cdiv(p.N, BN);
  const bool swap = p.splits <= 1 && tiles_m < 8 && tiles_n > tiles_m;   // (as in the kernel)
  const int inner = p.splits > 1 ? tiles_m * tiles_n : (swap ? tiles_m : tiles_n);
  const int outer = p.splits > 1 ? p.splits : (swap ? tiles_n : tiles_m);
  dim3 grid(outer >= 8 ? 8 * inner * cdiv(outer, 8) : outer * inner);
  // split arithmetic modes: the six-pass bf16 form of the same product (gemmsplit.hip; CGAT_GEMM_SPLIT=0: this engine)
  static const bool split_on = [] { const char* e = getenv("CGAT_GEMM_SPLIT"); return !(e && e[0] == '0'); }();
  bool on_split = split_on && bilinear_mode() != 0;
#ifdef CGAT_DEV_ABLATIONS
  if (getenv("CGAT_GEMM_ABL")) on_split = false;
#endif
  if (on_split) {
    CGAT_TRY(gemm_split_launch(p, grid.x, stream));
  } else {
    CGAT_PROF("gemm_f32", stream);
#ifdef CGAT_DEV_ABLATIONS   // timing-only variants (wrong results): only in builds made for tools/gemm_probe.py
    const char* ab = getenv("CGAT_GEMM_ABL");
    const int abl = ab ? atoi(ab) : 0;
#else
    const int abl = 0;
#endif
    if (p.a_outer) hipLaunchKernelGGL((gemm_f32_kernel<false, true, 0, 1>), grid, dim3(256), 0, stream, p);
    else if (p.b_outer) hipLaunchKernelGGL((gemm_f32_kernel<true, true, 0, 2>), grid, dim3(256), 0, stream, p);
    else if (abl == 1 && !p.a_kmajor && !p.b_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<false, false, 1>), grid, dim3(256), 0, stream, p);
    else if (abl == 2 && !p.a_kmajor && !p.b_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<false, false, 2>), grid, dim3(256), 0, stream, p);
    else if (abl == 4 && !p.a_kmajor && !p.b_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<false, false, 4>), grid, dim3(256), 0, stream, p);
    else if (abl == 6 && !p.a_kmajor && !p.b_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<false, false, 6>), grid, dim3(256), 0, stream, p);
    else if (abl == 7 && !p.a_kmajor && !p.b_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<false, false, 7>), grid, dim3(256), 0, stream, p);
    else if (!p.a_kmajor && !p.b_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, stream, p);
    else if (!p.a_kmajor && p.b_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, stream, p);
    else if (p.a_kmajor && !p.b_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, stream, p);
  }
  CGAT_LAUNCH_CHECK();
  if (p.splits > 1) {
    long total = (long)p.M * p.N;
    CGAT_TRY(splitk_reduce_launch(p.slab, p.splits, p.M, p.N, p.C, p.ldc, stream, p.alpha, p.beta, p.bias, p.act));
  }
  return CGAT_OK;
}
