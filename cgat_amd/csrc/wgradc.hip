// Hypernetwork weight gradient  out[l][a,b,c] = sum_n p[l][n,a] q[l][n,b] r[l][n,c]  (the autograd of the predicted
// Linear(C -> C*C + C) head, reference Hypernetworksmp.py:206-208, 236-240) in the f16x3c arithmetic: 24-bit operands
// as x = h + l + t (mfma_bf16.h), 3 fp16 passes + 3 six-bit correction passes = 3.75 pass-equivalents, where the
// bf16 form of the same contraction (bilinear.hip, bilinear_wgrad128_bf16_kernel<6>) pays six.  One launch covers every
// predicted layer of a hypernetwork, with no row split when the units fill the chip (round 5).
//
// The reduction runs over the ROWS n, and the row operand of the matrix instruction is the product p[n,a] * q[n,b]:
// it is formed and split on the fly, per `a`.  Structure (512 threads, one workgroup per CU, two waves per SIMD):
//  * wave (grp, wb) owns out[a0 + grp][32 wb .. +32][0 .. 128): 64 accumulator registers + 64 of the two-level sum.
//  * a chunk = 64 rows.  Per chunk a wave has two phases:
//      SPLIT  32 products per lane -> fp16 planes H, L (the A fragments of four 32x32x16 k-steps), the 24th bit T as a
//             bf16 plane, and from the three planes the 6-bit images of ONE 32x32x64 instruction each
//             (v_cvt_scalef32_pk32_{bf6,fp6}_{f16,bf16}: the planes ARE the instruction's source registers);
//      MFMA   per 32-column block: 3 correction instructions (t*h + h*t + l*l, K = 64) + 4 k-steps x (l*h, h*l, h*h).
//    SPLIT is vector work (~190 instructions) and waits for q; MFMA is 60 matrix instructions (1920 cycles).  The two
//    waves that share a SIMD (grp 0 / grp 1 of one wb) run the phases in OPPOSITE order inside an iteration -- grp 0
//    [SPLIT c, MFMA c], grp 1 [MFMA c-1, SPLIT c] -- so that one of them is always feeding the matrix pipe while the
//    other splits, and the q loads of a SPLIT need no register-resident prefetch (their latency is the partner's
//    matrix time).
//  * r arrives by LDS-DMA as a prepared stream of 50 KB per chunk (fp16 planes in B-fragment order + 6-bit images),
//    three ring slots; q is read by each wave directly in fragment order (qF: 8 coalesced 1-KB loads per chunk);
//    p as two 64-float rows per chunk through the ring slot.
//  * ONE workgroup barrier per iteration publishes the next slot and retires the oldest.
#include "kernels.h"
#include "mfma_bf16.h"
#include "wgrad_batch.h"
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef __bf16 bf16x32 __attribute__((ext_vector_type(32)));

#define WGC_ROWS 64                         // rows per chunk
#define WGC_PL_B 32768                      // fp16 planes of a chunk: [k-step 4][plane h,l][column block 4][lane 64][16 B]
#define WGC_IM_B 18432                      // 6-bit images: [image l6,h6,t6][column block 4][lane 64 x 16 B | lane 64 x 8 B]
#define WGC_RS_B (WGC_PL_B + WGC_IM_B)      // the r stream of a chunk: 50 pieces of 1 KB
#define WGC_P_B 2048                        // the two staged p rows, one 1-KB LDS-DMA piece each (256 B used)
#define WGC_SLOT_B (WGC_P_B + WGC_RS_B)     // ring slot: [p rows][r stream]
#define WGC_SLOTS 3
#define WGC_NDMA0 7                         // LDS-DMA pieces per chunk a grp-0 wave issues (of 13 per wb; grp 1: the rest)
#ifndef WGC_PRIO
#define WGC_PRIO 2
#endif
#if !defined(CGAT_DEV_ABLATIONS)   // the product build: the timing-only variants below do not exist, whatever -DWGC_ABL says
#undef WGC_ABL
#define WGC_ABL 0
#elif !defined(WGC_ABL)
#define WGC_ABL 0                           // timing-only ablations (wrong results): 1 no q loads, 2 no split arithmetic,
#endif                                      // 4 no LDS fragment reads, 8 no LDS-DMA, 16 no matrix instructions

// ------------------------------- operand preparation -------------------------------
// mx[4 * layer + which] = max |tensor|, which 0 / 1 / 2 = p / q / r  (mx zeroed before)
__global__ void wgc_absmax_kernel(WgradPrepDesc d, int l0, long ldp, long ldq, long ldr, int rows, int NA,
                                  float* __restrict__ mx) {
  const int layer = blockIdx.y / 3, which = blockIdx.y % 3;
  const float* t = which == 0 ? d.p[layer] : (which == 1 ? d.q[layer] : d.r[layer]);
  const long ld = which == 0 ? ldp : (which == 1 ? ldq : ldr);
  const int cols = which == 0 ? NA : 128;
  float m = 0.f;
  if (cols == 128 && (ld & 3) == 0 && (((uintptr_t)t) & 15) == 0) {
    const int c4 = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    for (long n = (long)blockIdx.x * 8 + r0; n < rows; n += (long)gridDim.x * 8) {
      const float4 v = *reinterpret_cast<const float4*>(t + n * ld + 4 * c4);
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)rows * cols; i += (long)gridDim.x * blockDim.x)
      m = fmaxf(m, fabsf(t[(i / cols) * ld + (i % cols)]));
  }
  block_absmax_commit(m, mx + 4 * (l0 + layer) + which);
}

// One 64-row chunk of one operand of one layer per workgroup (256 threads; blockIdx.z = operand):
//  pT [128][rows_pad]      = (p * 2^k * sign(n))^T: 2^k from max|p| max|q| (the products must fit fp16), sign(n) = -1 in the
//                            odd 512-row groups of n's row split (the kernel's partial sums alternate in sign)
//  qF [chunk][wb 4][j 8][lane 64][4]  = q in the order a lane of wave wb consumes it: value i = 4 j + e of lane (r, hi)
//                            is q[n0 + 16 (i >> 3) + 8 hi + (i & 7)][32 wb + r]
//  Rs [chunk][50 KB]       = fp16 planes of 2^k' r in B-fragment order + the three 6-bit images (same value order)
// Rows beyond `rows` and columns of p beyond NA are zero.
__global__ __launch_bounds__(256) void wgc_prep_kernel(WgradPrepDesc d, int l0, long ldp, long ldq, long ldr, int rows,
                                                       int NA, int rows_pad, int rows_per_split,
                                                       float* __restrict__ pT, float* __restrict__ qF,
                                                       unsigned char* __restrict__ Rs, long sT, long sR_bytes,
                                                       const float* __restrict__ mx) {
  __shared__ float tile[64][129];
  const int layer = blockIdx.y, chunk = blockIdx.x, n0 = chunk * WGC_ROWS, tid = threadIdx.x;
  const int which = blockIdx.z;      // 0 p, 1 q, 2 r: one operand per workgroup (three independent streams in flight per chunk)
  const float* mxl = mx + 4 * (l0 + layer);
  // ---- p: transpose, scale, sign ----
  if (which == 0) {
    float spq, ipq;
    pow2_scale(mxl[0] * mxl[1], spq, ipq);
    if ((((n0 % rows_per_split) >> 6) >> 3) & 1) spq = -spq;       // a 64-row chunk never straddles a 512-row group
    const float* p = d.p[layer];
    for (int i = tid; i < 64 * 128; i += 256) {
      const int n = i >> 7, c = i & 127;
      tile[n][c] = (n0 + n < rows && c < NA) ? p[(long)(n0 + n) * ldp + c] * spq : 0.f;
    }
    __syncthreads();
    float* o = pT + (long)(l0 + layer) * sT;
    for (int i = tid; i < 128 * 64; i += 256) {
      const int c = i >> 6, n = i & 63;
      o[(long)c * rows_pad + n0 + n] = tile[n][c];
    }
    __syncthreads();
  }
  // ---- q: fragment order ----
  if (which == 1) {
    const float* q = d.q[layer];
    for (int i = tid; i < 64 * 128; i += 256) {
      const int n = i >> 7, c = i & 127;
      tile[n][c] = (n0 + n < rows) ? q[(long)(n0 + n) * ldq + c] : 0.f;
    }
    __syncthreads();
    float4* o = reinterpret_cast<float4*>(qF + (long)(l0 + layer) * sT + (long)chunk * (64 * 128));
    for (int i = tid; i < 4 * 8 * 64; i += 256) {
      const int lane = i & 63, j = (i >> 6) & 7, wb = i >> 9;
      const int r = lane & 31, hi = lane >> 5, b = 32 * wb + r;
      const int nb = 16 * (j >> 1) + 8 * hi + 4 * (j & 1);
      o[i] = make_float4(tile[nb][b], tile[nb + 1][b], tile[nb + 2][b], tile[nb + 3][b]);
    }
    __syncthreads();
  }
  // ---- r: fp16 planes + 6-bit images; thread = (column block, lane) ----
  if (which == 2) {
    const float* rr = d.r[layer];
    float sr, ir;
    pow2_scale(mxl[2], sr, ir);
    for (int i = tid; i < 64 * 128; i += 256) {
      const int n = i >> 7, c = i & 127;
      tile[n][c] = (n0 + n < rows) ? rr[(long)(n0 + n) * ldr + c] * sr : 0.f;
    }
    __syncthreads();
    const int lane = tid & 63, cb = tid >> 6, col = lane & 31, hi = lane >> 5, c = 32 * cb + col;
    unsigned H[16], L[16], T[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {                      // value pair (2k, 2k+1): k-step k >> 2, element 2 (k & 3)
      const int n = 16 * (k >> 2) + 8 * hi + 2 * (k & 3);
      const float a = tile[n][c], b = tile[n + 1][c];
      const f16x2 h = __builtin_convertvector((f32x2){a, b}, f16x2);
      const float ra = a - (float)h[0], rb = b - (float)h[1];
      const f16x2 l = __builtin_convertvector((f32x2){ra, rb}, f16x2);
      const float ta = ra - (float)l[0], tb = rb - (float)l[1];
      const bf16x2 t = __builtin_convertvector((f32x2){ta, tb}, bf16x2);
      H[k] = __builtin_bit_cast(unsigned, h);
      L[k] = __builtin_bit_cast(unsigned, l);
      T[k] = __builtin_bit_cast(unsigned, t);
    }
    unsigned char* o = Rs + (long)(l0 + layer) * sR_bytes + (long)chunk * WGC_RS_B;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      *reinterpret_cast<uint4*>(o + ((s * 2 + 0) * 4 + cb) * 1024 + lane * 16) = make_uint4(H[4 * s], H[4 * s + 1], H[4 * s + 2], H[4 * s + 3]);
      *reinterpret_cast<uint4*>(o + ((s * 2 + 1) * 4 + cb) * 1024 + lane * 16) = make_uint4(L[4 * s], L[4 * s + 1], L[4 * s + 2], L[4 * s + 3]);
    }
    f16x32 hv, lv;
    bf16x32 tv;
    __builtin_memcpy(&hv, H, 64);
    __builtin_memcpy(&lv, L, 64);
    __builtin_memcpy(&tv, T, 64);
    const u32x6 l6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(lv, 1.0f);          // the instruction divides by its scale
    const u32x6 h6 = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(hv, 4096.0f);
    const u32x6 t6 = __builtin_amdgcn_cvt_scalef32_pk32_bf6_bf16(tv, 0x1p-12f);
    unsigned char* im = o + WGC_PL_B;
    *reinterpret_cast<uint4*>(im + (0 * 4 + cb) * 1536 + lane * 16) = make_uint4(l6[0], l6[1], l6[2], l6[3]);
    *reinterpret_cast<uint2*>(im + (0 * 4 + cb) * 1536 + 1024 + lane * 8) = make_uint2(l6[4], l6[5]);
    *reinterpret_cast<uint4*>(im + (1 * 4 + cb) * 1536 + lane * 16) = make_uint4(h6[0], h6[1], h6[2], h6[3]);
    *reinterpret_cast<uint2*>(im + (1 * 4 + cb) * 1536 + 1024 + lane * 8) = make_uint2(h6[4], h6[5]);
    *reinterpret_cast<uint4*>(im + (2 * 4 + cb) * 1536 + lane * 16) = make_uint4(t6[0], t6[1], t6[2], t6[3]);
    *reinterpret_cast<uint2*>(im + (2 * 4 + cb) * 1536 + 1024 + lane * 8) = make_uint2(t6[4], t6[5]);
  }
}

// ------------------------------- the contraction -------------------------------
__device__ __forceinline__ f32x16 wgc_mma6(const unsigned (&a)[6], uint4 b0, uint2 b1, f32x16 c, const int fmt) {
  i32x8 A, B;
  A[0] = a[0]; A[1] = a[1]; A[2] = a[2]; A[3] = a[3]; A[4] = a[4]; A[5] = a[5]; A[6] = 0; A[7] = 0;
  B[0] = b0.x; B[1] = b0.y; B[2] = b0.z; B[3] = b0.w; B[4] = b1.x; B[5] = b1.y; B[6] = 0; B[7] = 0;
  if (fmt == 2) return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 2, 2, 0, 0, 0, 0);   // fp6 x fp6
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 3, 3, 0, 0, 0, 0);                 // bf6 x bf6
}
__device__ __forceinline__ f32x16 wgc_mma16(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(512, 2) void bilinear_wgrad128_f16c_kernel(const float* __restrict__ pT_,
                                                                        const float* __restrict__ qF_,
                                                                        const unsigned char* __restrict__ Rs_,
                                                                        const float* __restrict__ mx_,
                                                                        WgradBatchDesc u) {
#if defined(WGC_STAMPS) && !defined(CGAT_DEV_ABLATIONS)
#error "WGC_STAMPS is a diagnostic build: add -DCGAT_DEV_ABLATIONS"
#endif
#ifdef WGC_STAMPS   // diagnostic build (tools/wgrad_stamps.py): s_memtime at the phase boundaries of iterations 100..103 of
  // workgroup 0, left in out[0] instead of that workgroup's results
  __shared__ __attribute__((aligned(16))) unsigned char smem[WGC_SLOTS * WGC_SLOT_B + 2048];
#define WGC_TS(k_)                                                                                       \
  if (c >= 100 && c < 104) {                                                                             \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                          \
    if (lane == 0) reinterpret_cast<unsigned long long*>(smem + WGC_SLOTS * WGC_SLOT_B)[(wave * 4 + (c - 100)) * 8 + (k_)] = t_; \
  }
#define WGC_TS7(dep_) { unsigned d_ = dep_[0]; asm volatile("" : "+v"(d_)); WGC_TS(7) }   /* after the value exists */
#else
  __shared__ __attribute__((aligned(16))) unsigned char smem[WGC_SLOTS * WGC_SLOT_B];
#define WGC_TS(k_)
#define WGC_TS7(dep_)
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wb = wave & 3;
  const int total = u.n_layers * u.splits * u.npairs, streams = u.n_layers * u.splits;
  const bool xcd_map = total % 8 == 0 && streams <= 8 && 8 % streams == 0 && u.npairs % (8 / streams) == 0 &&
                       gridDim.x % 8 == 0;
  const int rows_pad = u.rows_pad;
  const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;

  for (int v = blockIdx.x; v < total; v += gridDim.x) {
    // XCD-aware placement (speed only): the workgroups sharing an XCD share ONE (layer, split) stream and run in near
    // lockstep, so the stream enters that L2 once
    int stream, pair;
    if (xcd_map) {
      const int xcd = v & 7, w = v >> 3, xps = 8 / streams;
      stream = xcd / xps;
      pair = (xcd % xps) * (total / 8) + w;
    } else {
      stream = v / u.npairs;
      pair = v % u.npairs;
    }
    stream = __builtin_amdgcn_readfirstlane(stream);
    pair = __builtin_amdgcn_readfirstlane(pair);
    const int layer = stream / u.splits, z = stream % u.splits;
    const int a0 = pair * 2;
    const int nbeg = z * u.rows_per_split;
    const int nend = min(rows_pad, nbeg + u.rows_per_split);
    const int nchunks = (nend - nbeg) / WGC_ROWS;   // rows_per_split and rows_pad are multiples of 64
    const int cg0 = nbeg / WGC_ROWS;
    const char* pT = reinterpret_cast<const char*>(pT_ + (long)layer * u.sT + (long)a0 * rows_pad);
    const char* qF = reinterpret_cast<const char*>(qF_ + (long)layer * u.sT) + ((long)cg0 * 4 + wb) * 8192;   // uniform
    const unsigned char* Rs = Rs_ + (long)layer * u.sR + (long)cg0 * WGC_RS_B;
    const float* mx = mx_ + 4 * layer;

    f32x16 acc[4], tot[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int t = 0; t < 16; ++t) { acc[cb][t] = 0.f; tot[cb][t] = 0.f; }
    float inv_all;
    {
      float spq, ipq, sr, ir;
      pow2_scale(mx[0] * mx[1], spq, ipq);
      pow2_scale(mx[2], sr, ir);
      inv_all = ipq * ir;
    }
    if (nchunks > 0) {
      // chunk ci -> ring slot ci % 3, 52 pieces: 0..49 the r stream, 50 / 51 the p rows of a0 / a0 + 1; wave wb of the
      // issuing group takes the 13 pieces k = wb + 4 m.  In the loop the grp-1 waves issue them ONE PER TWO GROUPS inside
      // their matrix phase (WGC_STEP): issued as a burst by the grp-0 waves before their split, the 13 pieces cost those
      // waves ~800 cycles of their critical path, and the burst of 52 KB landing in LDS beside the fragment reads of the
      // partner's matrix phase and the p reads of the split stretched BOTH (any two of the three cost nothing, all three
      // +0.45 ms per launch: tools/wgrad_ablate.sh, round 5).  The chunk index is clamped: the last iterations re-load
      // the last chunk into a slot nobody reads.
#define WGC_DMA_PIECE(m_, sl_, rb_, cc_, vo16_, vo4_)   /* branch-free: every piece is one 1-KB global_load_lds_dwordx4 */ \
  if (!(WGC_ABL & 8)) {                                                                      \
    /* the bases are laundered so that each piece's address arithmetic happens HERE: hoisted to the top of the phase   \
       (13 pieces x 64-bit source + LDS address) it spilled 38 SGPRs into vector registers and 159 of those to scratch */ \
    const unsigned char* rbl_ = (rb_);                                                       \
    unsigned sll_ = (sl_);                                                                   \
    asm volatile("" : "+s"(rbl_), "+s"(sll_));                                               \
    const int k_ = wb + 4 * (m_);                                                            \
    if ((m_) < 12) {                                                                         \
      glds_b128(rbl_ + k_ * 1024, vo16_, (unsigned)__builtin_amdgcn_readfirstlane((int)(sll_ + WGC_P_B + k_ * 1024))); \
    } else {   /* k_ = 48, 49: r stream; 50, 51: the p row of a0 + k_ - 50 (256 B, fetched four times over) */     \
      const bool isr_ = wb < 2;                                                              \
      const unsigned char* src_ = isr_ ? rbl_ + k_ * 1024                                    \
                                       : reinterpret_cast<const unsigned char*>(pT) + ((long)(wb - 2) * rows_pad + nbeg + (long)(cc_) * WGC_ROWS) * 4; \
      const unsigned dst_ = isr_ ? sll_ + WGC_P_B + k_ * 1024 : sll_ + (wb - 2) * 1024;      \
      glds_b128(src_, isr_ ? (vo16_) : ((vo16_) & 255u), (unsigned)__builtin_amdgcn_readfirstlane((int)dst_)); \
    }                                                                                        \
  }
#define WGC_DMA(ci_, M0_, M1_)   /* pieces m in [M0_, M1_) of this wave */                   \
  {                                                                                          \
    const int cc_ = (ci_) < nchunks ? (ci_) : nchunks - 1;                                   \
    const unsigned sl_ = sbase + (unsigned)((ci_) % WGC_SLOTS) * WGC_SLOT_B;                 \
    int ln_ = lane;                                                                          \
    asm volatile("" : "+v"(ln_));   /* per-lane offsets are re-derived at every use: nothing lane-dependent stays live */ \
    const unsigned voff16 = (unsigned)ln_ * 16, voff4 = (unsigned)ln_ * 4;                   \
    const unsigned char* rb_ = Rs + (long)cc_ * WGC_RS_B;                                    \
    _Pragma("unroll") for (int m_ = (M0_); m_ < (M1_); ++m_) WGC_DMA_PIECE(m_, sl_, rb_, cc_, voff16, voff4) \
  }
      unsigned H[16], L[16];           // fp16 planes of the wave's products: the A fragments of the four k-steps
      unsigned A6l[6], A6h[6], A6t[6]; // their 6-bit images
#pragma unroll
      for (int i = 0; i < 16; ++i) { H[i] = 0; L[i] = 0; }
#pragma unroll
      for (int i = 0; i < 6; ++i) { A6l[i] = 0; A6h[i] = 0; A6t[i] = 0; }

      // SPLIT: the 32 products of chunk ci_ -> H, L, A6*.  PRE_ / POST_: what the caller issues between the q loads and
      // their wait (grp 0: the LDS-DMA of the next chunk, so that the wait need not cover it)
#define WGC_SPLIT(ci_, AFTER_LOADS_)                                                                              \
  {                                                                                                               \
    int lq_ = lane;                                                                                               \
    asm volatile("" : "+v"(lq_));                                                                                 \
    const char* qp_ = qF + (long)(ci_) * 32768;                                                                   \
    f32x4 qv_[8];                                                                                                 \
    /* issued from inline asm: the compiler cannot see the LDS-DMA below and would wait vmcnt(0) -- i.e. for the \
       DMA -- before the last q value; the values are tied to the counted wait instead */                       \
    if (WGC_ABL & 1) {                                                                                            \
      _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) qv_[j_] = (f32x4){1.f, 2.f, 3.f, 4.f};                     \
    } else {                                                                                                      \
      const unsigned qo_ = (unsigned)lq_ * 16u;                                                                   \
      asm volatile("global_load_dwordx4 %0, %8, %9\n\tglobal_load_dwordx4 %1, %8, %9 offset:1024\n\t"             \
                   "global_load_dwordx4 %2, %8, %9 offset:2048\n\tglobal_load_dwordx4 %3, %8, %9 offset:3072\n\t" \
                   "global_load_dwordx4 %4, %8, %10\n\tglobal_load_dwordx4 %5, %8, %10 offset:1024\n\t"           \
                   "global_load_dwordx4 %6, %8, %10 offset:2048\n\tglobal_load_dwordx4 %7, %8, %10 offset:3072"  \
                   : "=&v"(qv_[0]), "=&v"(qv_[1]), "=&v"(qv_[2]), "=&v"(qv_[3]), "=&v"(qv_[4]), "=&v"(qv_[5]),    \
                     "=&v"(qv_[6]), "=&v"(qv_[7])                                                                 \
                   : "v"(qo_), "s"(qp_), "s"(qp_ + 4096)                                                          \
                   : "memory");                                                                                   \
    }                                                                                                             \
    AFTER_LOADS_                                                                                                  \
    asm volatile("" : "+v"(qv_[0]), "+v"(qv_[1]), "+v"(qv_[2]), "+v"(qv_[3]), "+v"(qv_[4]), "+v"(qv_[5]),         \
                      "+v"(qv_[6]), "+v"(qv_[7]));   /* the values exist behind the counted wait, not before */   \
    const unsigned char* ps_ = smem + (unsigned)((ci_) % WGC_SLOTS) * WGC_SLOT_B + grp * 1024 + (lq_ >> 5) * 32;         \
    unsigned T_[16];                                                                                              \
    _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                                            \
      if (WGC_ABL & 2) { T_[2 * j_] = 0; T_[2 * j_ + 1] = 0; continue; }                                          \
      const float4 pv_ = *reinterpret_cast<const float4*>(ps_ + (j_ >> 1) * 64 + (j_ & 1) * 16);                   \
      const float x0_ = qv_[j_].x * pv_.x, x1_ = qv_[j_].y * pv_.y, x2_ = qv_[j_].z * pv_.z, x3_ = qv_[j_].w * pv_.w; \
      const f16x2 h0_ = __builtin_convertvector((f32x2){x0_, x1_}, f16x2);                                        \
      const f16x2 h1_ = __builtin_convertvector((f32x2){x2_, x3_}, f16x2);                                        \
      const float r0_ = x0_ - (float)h0_[0], r1_ = x1_ - (float)h0_[1], r2_ = x2_ - (float)h1_[0], r3_ = x3_ - (float)h1_[1]; \
      const f16x2 l0_ = __builtin_convertvector((f32x2){r0_, r1_}, f16x2);                                        \
      const f16x2 l1_ = __builtin_convertvector((f32x2){r2_, r3_}, f16x2);                                        \
      const bf16x2 t0_ = __builtin_convertvector((f32x2){r0_ - (float)l0_[0], r1_ - (float)l0_[1]}, bf16x2);      \
      const bf16x2 t1_ = __builtin_convertvector((f32x2){r2_ - (float)l1_[0], r3_ - (float)l1_[1]}, bf16x2);      \
      H[2 * j_] = __builtin_bit_cast(unsigned, h0_); H[2 * j_ + 1] = __builtin_bit_cast(unsigned, h1_);           \
      L[2 * j_] = __builtin_bit_cast(unsigned, l0_); L[2 * j_ + 1] = __builtin_bit_cast(unsigned, l1_);           \
      T_[2 * j_] = __builtin_bit_cast(unsigned, t0_); T_[2 * j_ + 1] = __builtin_bit_cast(unsigned, t1_);         \
    }                                                                                                             \
    WGC_TS(6)                                                                                                     \
    f16x32 hv_, lv_;                                                                                              \
    bf16x32 tv_;                                                                                                  \
    __builtin_memcpy(&hv_, H, 64);                                                                                \
    __builtin_memcpy(&lv_, L, 64);                                                                                \
    __builtin_memcpy(&tv_, T_, 64);                                                                               \
    const u32x6 l6_ = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(lv_, 1.0f);                                      \
    WGC_TS7(l6_)                                                                                                  \
    const u32x6 h6_ = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(hv_, 4096.0f);                                   \
    const u32x6 t6_ = __builtin_amdgcn_cvt_scalef32_pk32_bf6_bf16(tv_, 0x1p-12f);                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) { A6l[i_] = l6_[i_]; A6h[i_] = h6_[i_]; A6t[i_] = t6_[i_]; } \
  }

      // MFMA: chunk ci_ from ring slot ci_ % 3 with the fragments SPLIT left behind: 28 groups = per 32-column block four
      // k-steps of three fp16 passes (96 cycles) with the three correction instructions (32 cycles each) between them.
      // The B operands of group g + 3 are requested before group g issues (four rotating register sets, order pinned
      // with sched_barrier), i.e. 160-224 cycles of matrix work ahead of their use: left to itself the compiler requests
      // each operand right before its use and the wave sits out the LDS latency 20 times per chunk, and with two groups
      // of lead a matrix phase still took 2700 cycles for 1920 of matrix work (tools/wgrad_stamps.py).  512-row partial
      // sums (8 chunks) go to `tot` with alternating sign (the sign is in the staged p: cancels the matrix instruction's
      // accumulator rounding bias, see bilinear_rows128_ring16_kernel).
#define WGC_LOADG(g_, X_, Y_)                                                                                     \
  {                                                                                                               \
    constexpr int cb_ = (g_) / 7, k_ = (g_) % 7;                                                                  \
    if constexpr (WGC_ABL & 4) {                                                                                  \
    } else if constexpr (k_ & 1) {                                                                                \
      constexpr int im_ = k_ == 1 ? 1 : (k_ == 3 ? 2 : 0);       /* B image: h6 (for t*h), t6 (for h*t), l6 */    \
      X_ = *reinterpret_cast<const uint4*>(sl_ + WGC_PL_B + (im_ * 4 + cb_) * 1536);                              \
      const uint2 y2_ = *reinterpret_cast<const uint2*>(sl8_ + WGC_PL_B + (im_ * 4 + cb_) * 1536 + 1024);         \
      Y_.x = y2_.x; Y_.y = y2_.y;                                                                                 \
    } else {                                                                                                      \
      X_ = *reinterpret_cast<const uint4*>(sl_ + (((k_ >> 1) * 2 + 0) * 4 + cb_) * 1024);                         \
      Y_ = *reinterpret_cast<const uint4*>(sl_ + (((k_ >> 1) * 2 + 1) * 4 + cb_) * 1024);                         \
    }                                                                                                             \
  }
#define WGC_EXECG(g_, X_, Y_, PART_)   /* PART_ 0: the group's first matrix instruction; 1: the rest */              \
  {                                                                                                               \
    constexpr int cb_ = (g_) / 7, k_ = (g_) % 7;                                                                  \
    if constexpr (WGC_ABL & 16) {                                                                                 \
    } else if constexpr (k_ & 1) {                                                                                \
      if constexpr (PART_ == 0) {                                                                                 \
        if constexpr (k_ == 1) acc[cb_] = wgc_mma6(A6t, X_, make_uint2(Y_.x, Y_.y), acc[cb_], 3);                 \
        else if constexpr (k_ == 3) acc[cb_] = wgc_mma6(A6h, X_, make_uint2(Y_.x, Y_.y), acc[cb_], 3);            \
        else acc[cb_] = wgc_mma6(A6l, X_, make_uint2(Y_.x, Y_.y), acc[cb_], 2);                                   \
      }                                                                                                           \
    } else {                                                                                                      \
      constexpr int s_ = k_ >> 1;                                                                                 \
      const uint4 ah_ = make_uint4(H[4 * s_], H[4 * s_ + 1], H[4 * s_ + 2], H[4 * s_ + 3]);                       \
      const uint4 al_ = make_uint4(L[4 * s_], L[4 * s_ + 1], L[4 * s_ + 2], L[4 * s_ + 3]);                       \
      if constexpr (PART_ == 0) acc[cb_] = wgc_mma16(al_, X_, acc[cb_]);                                          \
      else {                                                                                                      \
        acc[cb_] = wgc_mma16(ah_, Y_, acc[cb_]);                                                                  \
        acc[cb_] = wgc_mma16(ah_, X_, acc[cb_]);                                                                  \
      }                                                                                                           \
    }                                                                                                             \
  }
#define WGC_STEP(g_, XA_, YA_, XC_, YC_)   /* group g_ from set A; group g_ + 3 requested into set C */            \
  {  /* the requests go BEHIND the group's first matrix instruction: a request that has to queue at the LDS then     \
        holds up the wave (in-order issue) while the matrix pipe has 32 cycles of work, not while it is empty */    \
    WGC_EXECG(g_, XA_, YA_, 0)                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    if constexpr ((g_) + 3 < 28) WGC_LOADG((g_) + 3 < 28 ? (g_) + 3 : 0, XC_, YC_)                                 \
    if constexpr (host_ && ((g_) & 1) && (g_) < 26) WGC_DMA_PIECE((g_) >> 1, dsl_, drb_, dcc_, dvo16_, dvo4_)      \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    WGC_EXECG(g_, XA_, YA_, 1)                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
  }
#define WGC_MFMA(ci_, HOST_, DCI_)   /* HOST_ 1: the LDS-DMA of chunk DCI_ is issued along the way */              \
  {                                                                                                               \
    constexpr bool host_ = HOST_;                                                                                 \
    if (((ci_) & 7) == 0 && (ci_) > 0) {                                                                          \
      const float sg_ = ((((ci_) >> 3) - 1) & 1) ? -1.f : 1.f;                                                    \
      _Pragma("unroll") for (int cb_ = 0; cb_ < 4; ++cb_)                                                         \
        _Pragma("unroll") for (int t_ = 0; t_ < 16; ++t_) {                                                       \
          tot[cb_][t_] = fmaf(acc[cb_][t_], sg_, tot[cb_][t_]);                                                   \
          acc[cb_][t_] = 0.f;                                                                                     \
        }                                                                                                         \
    }                                                                                                             \
    int lm_ = lane;                                                                                               \
    asm volatile("" : "+v"(lm_));                                                                                 \
    const unsigned char* sl_ = smem + (unsigned)((ci_) % WGC_SLOTS) * WGC_SLOT_B + WGC_P_B + lm_ * 16;            \
    const unsigned char* sl8_ = smem + (unsigned)((ci_) % WGC_SLOTS) * WGC_SLOT_B + WGC_P_B + lm_ * 8;            \
    const int dcc_ = (DCI_) < nchunks ? (DCI_) : nchunks - 1;                                                     \
    const unsigned dsl_ = sbase + (unsigned)((DCI_) % WGC_SLOTS) * WGC_SLOT_B;                                    \
    const unsigned char* drb_ = Rs + (long)dcc_ * WGC_RS_B;                                                       \
    const unsigned dvo16_ = (unsigned)lm_ * 16, dvo4_ = (unsigned)lm_ * 4;                                        \
    uint4 X0_ = {}, Y0_ = {}, X1_ = {}, Y1_ = {}, X2_ = {}, Y2_ = {}, X3_ = {}, Y3_ = {};                         \
    __builtin_amdgcn_s_setprio(WGC_PRIO);   /* the wave in its matrix phase goes first at the SIMD's issue port */ \
    WGC_LOADG(0, X0_, Y0_)                                                                                        \
    WGC_LOADG(1, X1_, Y1_)                                                                                        \
    WGC_LOADG(2, X2_, Y2_)                                                                                        \
    WGC_STEP(0, X0_, Y0_, X3_, Y3_) WGC_STEP(1, X1_, Y1_, X0_, Y0_) WGC_STEP(2, X2_, Y2_, X1_, Y1_)               \
    WGC_STEP(3, X3_, Y3_, X2_, Y2_) WGC_STEP(4, X0_, Y0_, X3_, Y3_) WGC_STEP(5, X1_, Y1_, X0_, Y0_)               \
    WGC_STEP(6, X2_, Y2_, X1_, Y1_) WGC_STEP(7, X3_, Y3_, X2_, Y2_) WGC_STEP(8, X0_, Y0_, X3_, Y3_)               \
    WGC_STEP(9, X1_, Y1_, X0_, Y0_) WGC_STEP(10, X2_, Y2_, X1_, Y1_) WGC_STEP(11, X3_, Y3_, X2_, Y2_)             \
    WGC_STEP(12, X0_, Y0_, X3_, Y3_) WGC_STEP(13, X1_, Y1_, X0_, Y0_) WGC_STEP(14, X2_, Y2_, X1_, Y1_)            \
    WGC_STEP(15, X3_, Y3_, X2_, Y2_) WGC_STEP(16, X0_, Y0_, X3_, Y3_) WGC_STEP(17, X1_, Y1_, X0_, Y0_)            \
    WGC_STEP(18, X2_, Y2_, X1_, Y1_) WGC_STEP(19, X3_, Y3_, X2_, Y2_) WGC_STEP(20, X0_, Y0_, X3_, Y3_)            \
    WGC_STEP(21, X1_, Y1_, X0_, Y0_) WGC_STEP(22, X2_, Y2_, X1_, Y1_) WGC_STEP(23, X3_, Y3_, X2_, Y2_)            \
    WGC_STEP(24, X0_, Y0_, X3_, Y3_) WGC_STEP(25, X1_, Y1_, X0_, Y0_) WGC_STEP(26, X2_, Y2_, X1_, Y1_)            \
    WGC_STEP(27, X3_, Y3_, X2_, Y2_)                                                                              \
    __builtin_amdgcn_s_setprio(0);                                                                                \
  }

      // chunks 0 and 1 as bursts (an `else` branch for the first iteration inside the loop made the compiler peel it and
      // spill 150 registers)
      if (grp == 0) WGC_DMA(0, 0, 13) else WGC_DMA(1, 0, 13)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // iteration c: chunk c + 1 is fetched into the slot whose last readers (grp 1, MFMA of chunk c - 2) passed the
      // barrier that ended iteration c - 1; the barrier that ends iteration c publishes it
      if (grp == 0) {
#pragma clang loop unroll(disable)
        for (int c = 0; c <= nchunks; ++c) {
          unsigned pf = 0;
          WGC_TS(0)
          if (c < nchunks) {
            // q of the NEXT chunk is pulled into this XCD's L2 by one load per wave whose 64 lanes touch the 64 lines of
            // the wave's 8-KB block (result unused; the register stays reserved until the wait that ends the iteration):
            // q is consumed where it lands, so without this every workgroup of an XCD -- they share a stream and run in
            // lockstep -- would sit out the same HBM miss in every chunk
            WGC_SPLIT(c, WGC_DMA(c + 1, 0, WGC_NDMA0)
                      {
                        const char* nb_ = qF + (long)(c + 1 < nchunks ? c + 1 : c) * 32768;
                        asm volatile("global_load_dword %0, %1, %2" : "=v"(pf) : "v"((unsigned)lq_ * 128u), "s"(nb_) : "memory");
                      }
                      wait_vmcnt<(WGC_ABL & 8) ? 1 : WGC_NDMA0 + 1>(); WGC_TS(1))
            WGC_TS(2)
            WGC_MFMA(c, 0, 0)
            WGC_TS(3)
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          asm volatile("" ::"v"(pf));
          WGC_TS(4)
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          WGC_TS(5)
        }
      } else {
#pragma clang loop unroll(disable)
        for (int c = 0; c <= nchunks; ++c) {
          WGC_TS(0)
          if (c > 0) WGC_MFMA(c - 1, 0, 0)
          WGC_TS(1)
          if (c < nchunks) WGC_SPLIT(c, WGC_DMA(c + 1, WGC_NDMA0, 13)   /* (iteration 0 repeats part of the prologue's chunk 1: harmless) */
                                     wait_vmcnt<(WGC_ABL & 8) ? 0 : 13 - WGC_NDMA0>(); WGC_TS(2))
          WGC_TS(3)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          WGC_TS(4)
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          WGC_TS(5)
        }
      }
#undef WGC_DMA
#undef WGC_DMA_PIECE
#undef WGC_SPLIT
#undef WGC_MFMA
#undef WGC_STEP
#undef WGC_EXECG
#undef WGC_LOADG
    }
    const int a = a0 + grp;
#ifdef WGC_STAMPS
    if (blockIdx.x == 0) {
      __syncthreads();
      if (tid < 256) reinterpret_cast<unsigned long long*>(u.out[0])[tid] =
          reinterpret_cast<unsigned long long*>(smem + WGC_SLOTS * WGC_SLOT_B)[tid];
    } else
#endif
    if (a < u.NA) {
      float* o = u.splits == 1 ? u.out[layer] + (long)a * 128 * 128
                               : u.slab + (((long)layer * u.splits + z) * u.NA + a) * 128 * 128;
      const float sg_last = (nchunks > 0 && (((nchunks - 1) >> 3) & 1)) ? -1.f : 1.f;
      // the lane id is laundered so that the per-lane store addresses are computed HERE, once per unit, instead of being
      // hoisted out of the unit loop and kept alive across the main loop
      int tl = tid;
      asm volatile("" : "+v"(tl));
      const int e_r = tl & 31, e_hi = (tl >> 5) & 1, e_wb = (tl >> 6) & 3;
      float* ol = o + (long)(e_wb * 32 + 4 * e_hi) * 128 + e_r;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        acc[cb] = (acc[cb] * sg_last + tot[cb]) * inv_all;
#pragma unroll
        for (int t = 0; t < 16; ++t) ol[((t & 3) + 8 * (t >> 2)) * 128 + cb * 32] = acc[cb][t];
      }
    }
    __syncthreads();   // the ring is re-filled by the next unit's prologue
  }
}

// out[layer][i] = sum_z slab[layer][z][i]
__global__ void wgc_slab_sum_kernel(const float* __restrict__ slab, int splits, long n, WgradBatchDesc u) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* sl = slab + (long)blockIdx.y * splits * n;
  float s = 0.f;
  for (int z = 0; z < splits; ++z) s += sl[(long)z * n + i];
  u.out[blockIdx.y][i] = s;
}

// ------------------------------- host side -------------------------------
// row splits per layer: enough units to fill the chip once (more only adds slab traffic), at least 4 chunks per split;
// a split is a whole number of 512-row sign groups unless it is the only one
int wgradc_pick(int n_layers, int nrows, int NA, int* rps_out) {
  const int npairs = cdiv(NA, 2), np = cdiv(nrows, WGC_ROWS) * WGC_ROWS;
  // work units (layer x row split x pair of a) ~ one per CU; CGAT_WGC_UNITS: tuning knob for the side-stream launch,
  // where units finer than the launch's workgroups balance a grid that is not a divisor of 256
  static const int target = [] { const char* e = getenv("CGAT_WGC_UNITS"); const int v = e ? atoi(e) : 256; return v >= 64 && v <= 2048 ? v : 256; }();
  int splits = target / (n_layers * npairs);
  if (splits > np / 256) splits = np / 256;
  if (splits < 1) splits = 1;
  int rps = cdiv(np / WGC_ROWS, splits) * WGC_ROWS;
  if (rps_out) *rps_out = rps;
  return cdiv(np, rps);
}
size_t wgradc_ws(int n_layers, int nrows, int NA, int splits, size_t* o_pT, size_t* o_qF, size_t* o_Rs, size_t* o_slab,
                 size_t* o_mx) {
  const size_t np = (size_t)cdiv(nrows, WGC_ROWS) * WGC_ROWS;
  size_t off = 0;
  *o_pT = off; off += ws_round((size_t)n_layers * np * 128, 4);
  *o_qF = off; off += ws_round((size_t)n_layers * np * 128, 4);
  *o_Rs = off; off += ws_round((size_t)n_layers * (np / WGC_ROWS) * WGC_RS_B, 1);
  *o_slab = off; if (splits > 1) off += ws_round((size_t)n_layers * splits * NA * 128 * 128, 4);
  *o_mx = off; off += 256;
  return off;
}
size_t wgradc_ws_bytes(int n_layers, int nrows, int NA) {
  size_t a, b, c, d, e;
  return wgradc_ws(n_layers, nrows, NA, wgradc_pick(n_layers, nrows, NA, nullptr), &a, &b, &c, &d, &e);
}

// operands of layers [l0, l0 + n) of an n_layers batch (their maxima, scaled transposes, fragment orders, planes, images)
int wgradc_prep(int l0, int n, int n_layers, const float* const* p, long ldp, const float* const* q, long ldq,
                const float* const* r, long ldr, int nrows, int NA, void* ws, size_t ws_bytes, hipStream_t stream) {
  const int np = cdiv(nrows, WGC_ROWS) * WGC_ROWS;
  int rps = 0;
  const int splits = wgradc_pick(n_layers, nrows, NA, &rps);
  size_t o_pT, o_qF, o_Rs, o_slab, o_mx;
  const size_t need = wgradc_ws(n_layers, nrows, NA, splits, &o_pT, &o_qF, &o_Rs, &o_slab, &o_mx);
  if (!ws || ws_bytes < need) {
    cgat_set_error("bilinear_wgrad (f16x3c): workspace too small (%zu < %zu)", ws_bytes, need);
    return CGAT_ERR_WORKSPACE;
  }
  WgradPrepDesc pd;
  memset(&pd, 0, sizeof(pd));
  for (int l = 0; l < n; ++l) { pd.p[l] = p[l]; pd.q[l] = q[l]; pd.r[l] = r[l]; }
  float* mx = (float*)((char*)ws + o_mx);
  CGAT_TRY(fill_launch(mx + 4 * l0, 0.f, 4 * n, stream));
  hipLaunchKernelGGL(wgc_absmax_kernel, dim3(nrows < 2048 ? cdiv(nrows, 8) : 256, 3 * n), dim3(256), 0, stream, pd, l0, ldp,
                     ldq, ldr, nrows, NA, mx);
  CGAT_LAUNCH_CHECK();
  hipLaunchKernelGGL(wgc_prep_kernel, dim3(np / WGC_ROWS, n, 3), dim3(256), 0, stream, pd, l0, ldp, ldq, ldr, nrows, NA, np, rps,
                     (float*)((char*)ws + o_pT), (float*)((char*)ws + o_qF), (unsigned char*)ws + o_Rs, (long)np * 128,
                     (long)(np / WGC_ROWS) * WGC_RS_B, (const float*)mx);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

int wgradc_launch(int n_layers, const float* const* p, long ldp, const float* const* q, long ldq, const float* const* r,
                  long ldr, float* const* out, int nrows, int NA, void* ws, size_t ws_bytes, hipStream_t stream,
                  int max_wgs, bool prepared) {
  if (max_wgs <= 0 || max_wgs > 256) max_wgs = 256;
  const int npairs = cdiv(NA, 2);
  const int np = cdiv(nrows, WGC_ROWS) * WGC_ROWS;
  int rps = 0;
  const int splits = wgradc_pick(n_layers, nrows, NA, &rps);
  size_t o_pT, o_qF, o_Rs, o_slab, o_mx;
  const size_t need = wgradc_ws(n_layers, nrows, NA, splits, &o_pT, &o_qF, &o_Rs, &o_slab, &o_mx);
  if (!ws || ws_bytes < need) {
    cgat_set_error("bilinear_wgrad (f16x3c): workspace too small (%zu < %zu)", ws_bytes, need);
    return CGAT_ERR_WORKSPACE;
  }
  if (!prepared) CGAT_TRY(wgradc_prep(0, n_layers, n_layers, p, ldp, q, ldq, r, ldr, nrows, NA, ws, ws_bytes, stream));
  WgradBatchDesc u;
  memset(&u, 0, sizeof(u));
  for (int l = 0; l < n_layers; ++l) u.out[l] = out[l];
  u.slab = (float*)((char*)ws + o_slab);
  u.sT = (long)np * 128;
  u.sR = (long)(np / WGC_ROWS) * WGC_RS_B;
  u.n_layers = n_layers; u.splits = splits; u.npairs = npairs; u.NA = NA; u.rows_pad = np; u.rows_per_split = rps;
  const int units = n_layers * splits * npairs;
  const int grid = units < max_wgs ? units : max_wgs;
  {
    CGAT_PROF("bilinear_wgrad", stream);
    hipLaunchKernelGGL(bilinear_wgrad128_f16c_kernel, dim3(grid), dim3(512), 0, stream, (const float*)((char*)ws + o_pT),
                       (const float*)((char*)ws + o_qF), (const unsigned char*)ws + o_Rs, (const float*)((char*)ws + o_mx), u);
  }
  CGAT_LAUNCH_CHECK();
#ifdef WGC_STAMPS
  return CGAT_OK;
#endif
  if (splits > 1) {
    const long n = (long)NA * 128 * 128;
    hipLaunchKernelGGL(wgc_slab_sum_kernel, dim3(cdiv(n, 256), n_layers), dim3(256), 0, stream, (const float*)u.slab, splits,
                       n, u);
    CGAT_LAUNCH_CHECK();
  }
  return CGAT_OK;
}
