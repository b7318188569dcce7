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
#include <string.h>

#include "common.h"
#include "kernels.h"
#include "mfma_bf16.h"
#include "wgrad_batch.h"

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

// ---------------------------------------------------------------------------------------
// Split-bf16 form of the same contraction (the default arithmetic, "bf16x6").  Every fp32 operand x is
// written x = x1 + x2 + x3 with bf16 pieces (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2):
// 24 significant bits in all) and a product a*b is accumulated in fp32 as
//     a3b1 + a1b3 + a2b2 + a2b1 + a1b2 + a1b1        (six bf16 MFMA passes, smallest terms first);
// products of bf16 pairs are exact in fp32 and the dropped terms are <= 2^-24 relative.  Measured on
// MI355X (tools/bf16x3_probe.hip): max-norm relative error vs fp64 4.0e-7 / 9.4e-7 / 3.2e-6 at K = 128 /
// 2048 / 16384, against 4.5e-7 / 1.2e-6 / 4.2e-6 for the f32-input MFMA chain -- the same accuracy, at
// a matrix-core ceiling of 2500/6 = 417 TFLOP/s of fp32-equivalent work instead of 157.  PASSES = 3
// keeps only a2b1 + a1b2 + a1b1 (~3e-6; mode "bf16x3", never the default).
//
// The bf16 MFMA's accumulator alignment drops low bits floor-wise (measured: a sign-independent bias of
// about -5e-11 of the running sum per accumulation step; the f32-input MFMA rounds to nearest), which
// over 16 384-term sums is coherent across rows.  Consecutive partial sums are therefore accumulated
// with opposite product signs ((-1)^a folded into the prepared T and into p) so the biases cancel.
//
// "Scale-after" evaluation order:
//     out[n,c] = init[n,c] + sum_a p[n,a] * ( sum_b q[n,b] T[a,b,c] )
// The inner sum is a K = 128 GEMM whose row operand q never changes: its three bf16 planes are split
// ONCE and stay in 96 VGPRs for the whole kernel, T arrives pre-split, so the loop holds no operand
// arithmetic at all -- LDS fragment reads and MFMAs only.  The 128*6 exact products of one `a` go into
// fresh fp32 accumulators (the inner level of the two-level summation), which are then scaled by
// p[n,a] and added to the totals (one v_fma per 3 MFMAs).  The product is computed transposed
// (D[c,n]: T fragment = the MFMA's A operand, q fragment = B) so that every accumulator register of a
// lane belongs to one of its two rows and p[n,a] is a per-lane scalar.
//
// T reaches LDS by LDS-DMA (global_load_lds: no staging VGPRs, no ds_write) into a 4-slot ring of 24-KB
// chunks with the loads of chunk i+3 in flight while chunk i is consumed; one raw s_barrier per chunk
// behind a counted vmcnt.  The p column of each `a` is staged the same way two `a` ahead.  Fragments
// are read one 12-MFMA group ahead -- across the barrier too, because chunk i+1 was already retired
// and published by the barrier that ended chunk i-1 -- so a wave never waits on LDS latency with an
// empty matrix pipe.
//   RAW: chunk j is issued in iteration j-3, retired by this wave's `vmcnt(3)` at the end of iteration
//        j-2, published by the barrier that follows; first read in iteration j-1 (group-0 prefetch).
//   WAR: chunk j+3 overwrites the slot of chunk j-1, all of whose reads were consumed by MFMAs issued
//        before the barrier that ended iteration j-1; the glds is issued after that barrier.
// The LDS-DMA is issued from inline asm: hipcc would otherwise wait vmcnt(0) before every ds_read that
// follows a glds builtin (it cannot tell ring slots apart).  No other vector-memory instruction may
// appear in the loop, so the counts are exact: per iteration 3 T loads, preceded (chunk 0 of each `a`)
// by one p load.
//
// MFMA shape: v_mfma_f32_16x16x32_bf16.  Same flop per cycle as 32x32x16, but under the chip's power
// management it sustains a higher clock (MI355X_MICROARCH.md "DVFS give-back" item 7); measured here
// on the same kernel structure: 1.39 ms vs 1.54 ms per 83 340-row launch.
// Wave tile 32 rows x 128 columns = 2 row blocks x 8 column blocks of 16; a lane owns rows
// n = (lane & 15) and 16 + (lane & 15) and, per column block, columns 4 (lane >> 4) + 0..3.
// T layout (prepare_T_bf16_kernel), holding (-1)^a T[a]:
//   Tq[a][half = c/64][kh = b/64][s2 = (b/32)%2][piece][cb = (c%64)/16][kg = (b%32)/8][i = c%16][j = b%8]
// one chunk = (a, half, kh) = 2 k-steps x 3 planes x 4 column blocks x 1 KB; a fragment is one ds_read_b128.

// Timing-only ablations of the fp16 form at 83 340 rows (tools/ring_ablation.py; ABL bits: 1 no barrier, 2 no global
// loads, 4 no fragment reads, 8 no flush -- the last two let the compiler drop MFMAs and are not usable as skeleton
// times): 754 us as is, 729 without the barrier, 682 without the LDS-DMA loads, 673 without both, against 520 us of
// pure MFMA issue at the 1.9 GHz the chip holds here.  Tried and dropped (round 1): 4 waves x 64 rows per workgroup
// (one wave per SIMD on the 512-register budget, every T fragment feeding four row blocks instead of two, i.e. half
// the LDS fragment traffic): 777 us -- what it saves in LDS reads it loses by having no second wave to cover the
// flush, the barrier wait and the accumulator-register copies.
template <int PASSES, int ABL = 0>
__global__ __launch_bounds__(512, 2) void bilinear_rows128_ring16_kernel(
    const float* __restrict__ p, long ldp, const float* __restrict__ q, long ldq, const uint4* __restrict__ Tq,
    const float* __restrict__ init, long ldi, float* __restrict__ out, long ldo, int nrows, int NA, int tiles, int asplit,
    long slab_stride, int vec_io, const float* __restrict__ tmax) {
  constexpr bool F16 = PASSES == 2;             // two fp16 planes, three passes (mfma_bf16.h); tmax = max |T|
  constexpr int NP = F16 ? 2 : 3;               // planes per operand
  constexpr int CH16 = 2 * NP * 4 * 64;         // 16-byte pieces per chunk = 24 KB (16 KB)
  constexpr int PST = 8 * 64;
  __shared__ uint4 smem[4 * CH16 + 4 * PST / 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kg = lane >> 4;
  const int split = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const int a_beg = (int)((long)NA * split / asplit), a_end = (int)((long)NA * (split + 1) / asplit);
  const int row_w = tile * 256 + wave * 32;
  const int row_a = row_w + n16, row_b = row_w + 16 + n16;           // the lane's two output rows
  const long rowc_a = row_a < nrows ? row_a : nrows - 1, rowc_b = row_b < nrows ? row_b : nrows - 1;
  const int row_st = row_w + (lane & 31);                            // the row whose p this lane stages
  const long rowc_st = row_st < nrows ? row_st : nrows - 1;
  if (asplit > 1) {
    out += (long)split * slab_stride;
    if (split > 0) init = nullptr;
  }
  const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const unsigned wave_t = __builtin_amdgcn_readfirstlane(sbase + wave * 1024);
  const unsigned wave_p = __builtin_amdgcn_readfirstlane(sbase + 4 * CH16 * 16 + wave * 256);
  const bf16x8* ring = reinterpret_cast<const bf16x8*>(smem) + lane;
  const float* pst = reinterpret_cast<const float*>(smem + 4 * CH16) + wave * 64 + n16;
  p += (long)tile * 256 * ldp;                                       // scalar tile base + 32-bit lane offsets
  const unsigned prow_off = (unsigned)((rowc_st - (long)tile * 256) * ldp * 4);
  const unsigned t_off = (unsigned)tid * 16;
  const long last_chunk = (long)a_end * 4 - 1;

  // q[row, 32 s + 8 kg + j] for both row blocks as three bf16 (two fp16) planes: qf[plane][2 s + nb]
  bf16x8 q1[8], q2[8], q3[F16 ? 1 : 8];
  float rs_a = 1.f, rs_b = 1.f;                 // F16: 1 / (scale of the lane's q row * scale of T)
  if constexpr (F16) {
    float qv[2][32];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4* qp = reinterpret_cast<const float4*>(q + (nb ? rowc_b : rowc_a) * ldq + 32 * s + 8 * kg);
        const float4 t0 = qp[0], t1 = qp[1];
        qv[nb][8 * s + 0] = t0.x; qv[nb][8 * s + 1] = t0.y; qv[nb][8 * s + 2] = t0.z; qv[nb][8 * s + 3] = t0.w;
        qv[nb][8 * s + 4] = t1.x; qv[nb][8 * s + 5] = t1.y; qv[nb][8 * s + 6] = t1.z; qv[nb][8 * s + 7] = t1.w;
      }
    float st, it;
    pow2_scale(tmax[0], st, it);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float m = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) m = fmaxf(m, fabsf(qv[nb][j]));
      m = fmaxf(m, __shfl_xor(m, 16));          // the row's 128 values live in the four lanes n16 + 16 kg
      m = fmaxf(m, __shfl_xor(m, 32));
      float sq, iq;
      pow2_scale(m, sq, iq);
      (nb ? rs_b : rs_a) = iq * it;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = qv[nb][8 * s + j] * sq;
        split2_x8_f16(v, q1[2 * s + nb], q2[2 * s + nb]);
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const float4* qp = reinterpret_cast<const float4*>(q + (nb ? rowc_b : rowc_a) * ldq + 32 * s + 8 * kg);
        const float4 t0 = qp[0], t1 = qp[1];
        const float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        split3_x8(v, q1[2 * s + nb], q2[2 * s + nb], q3[2 * s + nb]);
      }
  }
  // acc[2 cb8 + nb][t] = out[row(nb)][16 cb8 + 4 kg + t]
  f32x4 acc[16];
#pragma unroll
  for (int cb = 0; cb < 8; ++cb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int row = nb ? row_b : row_a;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (init && row < nrows) {
        const float* ip = init + (long)row * ldi + 16 * cb + 4 * kg;
        if (vec_io) v = *reinterpret_cast<const float4*>(ip);
        else v = make_float4(ip[0], ip[1], ip[2], ip[3]);
      }
      acc[2 * cb + nb][0] = v.x; acc[2 * cb + nb][1] = v.y; acc[2 * cb + nb][2] = v.z; acc[2 * cb + nb][3] = v.w;
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#define RG_TLOAD(gi_)                                                                          \
  {                                                                                            \
    const long gi = (gi_) < last_chunk ? (gi_) : last_chunk;                                   \
    const uint4* tb = Tq + gi * CH16;                                                          \
    const unsigned dst = wave_t + (unsigned)((gi_) & 3) * (CH16 * 16);                         \
    glds_b128(tb, t_off, dst);                                                                 \
    glds_b128(tb + 512, t_off, dst + 8192);                                                    \
    if (NP == 3) glds_b128(tb + 1024, t_off, dst + 16384);                                     \
  }
#define RG_PLOAD(a_)                                                                           \
  {                                                                                            \
    const int aa = (a_) < a_end ? (a_) : a_end - 1;                                            \
    glds_b32(p + aa, prow_off, wave_p + (unsigned)((a_) & 3) * (PST * 4));                     \
  }
  RG_PLOAD(a_beg);
  RG_PLOAD(a_beg + 1);
  RG_TLOAD((long)a_beg * 4 + 0);
  RG_TLOAD((long)a_beg * 4 + 1);
  RG_TLOAD((long)a_beg * 4 + 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  bf16x8 fa1, fa2, fa3, fb1, fb2, fb3;
  // group (s2, cb): the three planes of one 16-column block at one k-step
#define RG_READ(F1_, F2_, F3_, slot_, s2_, cb_)                                                \
  {                                                                                            \
    const bf16x8* fp = ring + (slot_) * (CH16) + (((s2_) * NP) * 4 + (cb_)) * 64;              \
    if (!(ABL & 4) || (slot_) + (s2_) + (cb_) == 0) {                                          \
      F1_ = fp[0];                                                                             \
      F2_ = fp[4 * 64];                                                                        \
      if (PASSES >= 6) F3_ = fp[8 * 64];                                                       \
    }                                                                                          \
  }
#define RG_MFMA1(F1_, F2_, F3_, qi_, P_)                                                       \
  {                                                                                            \
    if (PASSES >= 6) {                                                                         \
      P_ = mma16<F16>(F3_, q1[qi_], P_);                                                       \
      P_ = mma16<F16>(F1_, q3[qi_], P_);                                                       \
      P_ = mma16<F16>(F2_, q2[qi_], P_);                                                       \
    }                                                                                          \
    P_ = mma16<F16>(F2_, q1[qi_], P_);                                                         \
    P_ = mma16<F16>(F1_, q2[qi_], P_);                                                         \
    P_ = mma16<F16>(F1_, q1[qi_], P_);                                                         \
  }
#define RG_MFMA(F1_, F2_, F3_, s_, cb_)                                                        \
  {                                                                                            \
    RG_MFMA1(F1_, F2_, F3_, 2 * (s_) + 0, part[2 * (cb_) + 0])                                 \
    RG_MFMA1(F1_, F2_, F3_, 2 * (s_) + 1, part[2 * (cb_) + 1])                                 \
  }
  RG_READ(fa1, fa2, fa3, 0, 0, 0);
  if constexpr ((ABL & 4) != 0) { fb1 = fa1; fb2 = fa2; fb3 = fa3; }
  f32x4 part[8];
  for (int a = a_beg; a < a_end; ++a) {
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      const int half = ch >> 1, c2 = ch & 1;
      if constexpr (!(ABL & 2)) {
        RG_TLOAD((long)a * 4 + ch + 3);
        if (ch == 0) RG_PLOAD(a + 2);          // AFTER the T loads: see the wait below
      }
      if (c2 == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) part[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int s = 2 * c2 + s2;
#pragma unroll
        for (int cbp = 0; cbp < 2; ++cbp) {
          // column block 2 cbp on set A (read 2 cbp + 1 into set B first), then 2 cbp + 1 on set B
          RG_READ(fb1, fb2, fb3, ch, s2, 2 * cbp + 1);
          __builtin_amdgcn_sched_barrier(0);
          RG_MFMA(fa1, fa2, fa3, s, 2 * cbp);
          if (cbp == 0) RG_READ(fa1, fa2, fa3, ch, s2, 2)
          else if (s2 == 0) RG_READ(fa1, fa2, fa3, ch, 1, 0)
          else RG_READ(fa1, fa2, fa3, (ch + 1) & 3, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          RG_MFMA(fb1, fb2, fb3, s, 2 * cbp + 1);
        }
      }
      if (c2 == 1 && !(ABL & 8)) {
        float pva = pst[(a & 3) * PST], pvb = pst[(a & 3) * PST + 16];
        if constexpr (F16) { pva *= rs_a; pvb *= rs_b; }
        const float pas_a = (a & 1) ? -pva : pva, pas_b = (a & 1) ? -pvb : pvb;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            acc[2 * (4 * half + cb) + 0][t] = fmaf(pas_a, part[2 * cb + 0][t], acc[2 * (4 * half + cb) + 0][t]);
            acc[2 * (4 * half + cb) + 1][t] = fmaf(pas_b, part[2 * cb + 1][t], acc[2 * (4 * half + cb) + 1][t]);
          }
      }
      // chunk i + 2 (issued one iteration ago) must have landed.  Younger than it: this iteration's T loads and, for
      // ch < 2, the p load issued right behind the T loads of ch == 0 (needed two `a` later; issued BEFORE them, as it
      // used to be, every ch == 0 wait drained it -- a 64-line strided load -- within one k-step)
      if (ch < 2) wait_vmcnt<NP + 1>();
      else wait_vmcnt<NP>();
      if constexpr (!(ABL & 1)) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef RG_TLOAD
#undef RG_PLOAD
#undef RG_READ
#undef RG_MFMA1
#undef RG_MFMA
#pragma unroll
  for (int cb = 0; cb < 8; ++cb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int row = nb ? row_b : row_a;
      if (row < nrows) {
        float* op = out + (long)row * ldo + 16 * cb + 4 * kg;
        const f32x4 v = acc[2 * cb + nb];
        if (vec_io) *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
        else { op[0] = v[0]; op[1] = v[1]; op[2] = v[2]; op[3] = v[3]; }
      }
    }
}

// ---------------------------------------------------------------------------------------
// f16x3c form of the forward contraction (round 4): 24-bit operands at 1.25 x the matrix time of the fp16 form.
// Same workgroup shape, ring, scale-after flush and sign alternation as bilinear_rows128_ring16_kernel<2>; what differs:
//  * a chunk is (a, column half, PAIR of 16-column blocks) over the whole K = 128 (prepare_T_f16c_kernel's image: 16 KB of
//    fp16 planes + 9 KB of 6-bit images, contiguous), so the partial accumulators of a chunk are 2 blocks x 2 row blocks
//    = 16 registers instead of 32 and are flushed at the end of every chunk -- that is what makes room for the 36 registers
//    of the row operand's 6-bit images;
//  * per chunk 8 groups of 6 fp16 MFMAs (k-step s, block cb2) and, riding with the first six groups, the three
//    correction terms (t6 x h6, h6 x t6, l6 x l6: one v_mfma_f32_16x16x128_f8f6f4 per row block each) of the chunk's two
//    column blocks; every fragment is read one group ahead;
//  * 25 LDS-DMA pieces of 1 KB per chunk: three per wave and a fourth by wave 0 (its counted waits allow one more).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void bilinear_rows128_ring16c_kernel(
    const float* __restrict__ p, long ldp, const float* __restrict__ q, long ldq, const uint4* __restrict__ Tq,
    const float* __restrict__ init, long ldi, float* __restrict__ out, long ldo, int nrows, int NA, int tiles, int asplit,
    long slab_stride, int vec_io, const float* __restrict__ tmax) {
  constexpr int CH16 = F16C_CHUNK16;
  constexpr int PST = 8 * 64;
  __shared__ uint4 smem[4 * CH16 + 4 * PST / 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int n16 = lane & 15, kg = lane >> 4;
  const int split = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const int a_beg = (int)((long)NA * split / asplit), a_end = (int)((long)NA * (split + 1) / asplit);
  const int row_w = tile * 256 + wave * 32;
  const int row_a = row_w + n16, row_b = row_w + 16 + n16;           // the lane's two output rows
  const long rowc_a = row_a < nrows ? row_a : nrows - 1, rowc_b = row_b < nrows ? row_b : nrows - 1;
  const int row_st = row_w + (lane & 31);                            // the row whose p this lane stages
  const long rowc_st = row_st < nrows ? row_st : nrows - 1;
  if (asplit > 1) {
    out += (long)split * slab_stride;
    if (split > 0) init = nullptr;
  }
  const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const unsigned wave_t = __builtin_amdgcn_readfirstlane(sbase + wave * 1024);
  const unsigned wave_p = __builtin_amdgcn_readfirstlane(sbase + 4 * CH16 * 16 + wave * 256);
  const bf16x8* ring = reinterpret_cast<const bf16x8*>(smem) + lane;
  const unsigned char* cring = reinterpret_cast<const unsigned char*>(smem) + 16384;
  const float* pst = reinterpret_cast<const float*>(smem + 4 * CH16) + wave * 64 + n16;
  p += (long)tile * 256 * ldp;                                       // scalar tile base + 32-bit lane offsets
  const unsigned prow_off = (unsigned)((rowc_st - (long)tile * 256) * ldp * 4);
  const unsigned t_off = (unsigned)tid * 16;
  const long last_chunk = (long)a_end * 4 - 1;

  // q[row, 32 s + 8 kg + j] of both row blocks, scaled per row: two fp16 planes qf[2 s + nb] + the three 6-bit images
  bf16x8 q1[8], q2[8];
  frag6 ql6[2], qh6[2], qt6[2];
  float rs_a, rs_b;                             // 1 / (scale of the lane's q row * scale of T)
  {
    float qv[2][32];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4* qp = reinterpret_cast<const float4*>(q + (nb ? rowc_b : rowc_a) * ldq + 32 * s + 8 * kg);
        const float4 t0 = qp[0], t1 = qp[1];
        qv[nb][8 * s + 0] = t0.x; qv[nb][8 * s + 1] = t0.y; qv[nb][8 * s + 2] = t0.z; qv[nb][8 * s + 3] = t0.w;
        qv[nb][8 * s + 4] = t1.x; qv[nb][8 * s + 5] = t1.y; qv[nb][8 * s + 6] = t1.z; qv[nb][8 * s + 7] = t1.w;
      }
    float st, it;
    pow2_scale(tmax[0], st, it);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float m = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) m = fmaxf(m, fabsf(qv[nb][j]));
      m = fmaxf(m, __shfl_xor(m, 16));          // the row's 128 values live in the four lanes n16 + 16 kg
      m = fmaxf(m, __shfl_xor(m, 32));
      float sq, iq;
      pow2_scale(m, sq, iq);
      (nb ? rs_b : rs_a) = iq * it;
#pragma unroll
      for (int j = 0; j < 32; ++j) qv[nb][j] *= sq;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = qv[nb][8 * s + j];
        split2_x8_f16(v, q1[2 * s + nb], q2[2 * s + nb]);
      }
      f16c_pack32(qv[nb], ql6[nb], qh6[nb], qt6[nb]);   // element 8 s + j <-> k = 32 s + 8 kg + j, as in the image
    }
  }
  // acc[2 cb8 + nb][t] = out[row(nb)][16 cb8 + 4 kg + t]
  f32x4 acc[16];
#pragma unroll
  for (int cb = 0; cb < 8; ++cb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int row = nb ? row_b : row_a;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (init && row < nrows) {
        const float* ip = init + (long)row * ldi + 16 * cb + 4 * kg;
        if (vec_io) v = *reinterpret_cast<const float4*>(ip);
        else v = make_float4(ip[0], ip[1], ip[2], ip[3]);
      }
      acc[2 * cb + nb][0] = v.x; acc[2 * cb + nb][1] = v.y; acc[2 * cb + nb][2] = v.z; acc[2 * cb + nb][3] = v.w;
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ... and the compiler must KNOW that these loads have landed: it does not see the wait above, keeps them pending across
  // the loop header and waits for them at their first use -- the flush of every chunk, an `s_waitcnt vmcnt(0)` inside the
  // loop that also drains the LDS-DMA pieces just issued for three chunks ahead (round 5: found in the ISA after the
  // kernel's ablations showed the stream costing 0.3 ms per launch that nothing else accounted for)
#pragma unroll
  for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(acc[i]));

#if !defined(CGAT_DEV_ABLATIONS)   // the product build: the timing-only variants below do not exist, whatever -DRC_ABL says
#undef RC_ABL
#define RC_ABL 0
#elif !defined(RC_ABL)
#define RC_ABL 0   // timing-only ablations (wrong results; tools/rc_ablate.sh): 1 no LDS-DMA, 2 no fragment reads, 4 no matrix
#endif             // instructions, 8 no flush, 16 no wait + barrier per chunk
#define RC_TLOAD(gi_)                                                                          \
  { if (!(RC_ABL & 1)) {                                                                         \
    const long gi = (gi_) < last_chunk ? (gi_) : last_chunk;                                   \
    const uint4* tb = Tq + gi * CH16;                                                          \
    const unsigned dst = wave_t + (unsigned)((gi_) & 3) * (CH16 * 16);                         \
    glds_b128(tb, t_off, dst);                                                                 \
    glds_b128(tb + 512, t_off, dst + 8192);                                                    \
    glds_b128(tb + 1024, t_off, dst + 16384);                                                  \
    if (wave_u == 0) glds_b128(tb + 1536, t_off, dst + 24576);                                 \
  } }
#define RC_PLOAD(a_)                                                                           \
  { if (!(RC_ABL & 1)) {                                                                         \
    const int aa = (a_) < a_end ? (a_) : a_end - 1;                                            \
    glds_b32(p + aa, prow_off, wave_p + (unsigned)((a_) & 3) * (PST * 4));                     \
  } }
  // everything except the N_ youngest vector-memory operations of this wave (wave 0: + its extra piece) has landed
#define RC_WAIT(N_) { if (wave_u == 0) wait_vmcnt<(N_) + 1>(); else wait_vmcnt<(N_)>(); }
  RC_PLOAD(a_beg);
  RC_PLOAD(a_beg + 1);
  RC_TLOAD((long)a_beg * 4 + 0);
  RC_TLOAD((long)a_beg * 4 + 1);
  RC_TLOAD((long)a_beg * 4 + 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  bf16x8 fa1, fa2, fb1, fb2;
  frag6 ce;
  // group g = 2 s + cb2 of a chunk: the two planes of block cb2 at k-step s
#define RC_READ(F1_, F2_, slot_, g_)                                                           \
  { if (!(RC_ABL & 2)) {                                                                         \
    const bf16x8* fp = ring + (slot_) * CH16 + ((((g_) >> 1) * 2) * 2 + ((g_) & 1)) * 64;      \
    F1_ = fp[0];                                                                               \
    F2_ = fp[2 * 64];                                                                          \
  } }
  // 6-bit fragment j = 3 cb2 + term of a chunk
#define RC_CREAD(slot_, j_)                                                                    \
  { if (!(RC_ABL & 2)) {                                                                         \
    const unsigned char* cp = cring + (slot_) * (CH16 * 16) + (j_) * 1536;                     \
    const uint4 u_ = *reinterpret_cast<const uint4*>(cp + lane * 16);                          \
    const uint2 w_ = *reinterpret_cast<const uint2*>(cp + 1024 + lane * 8);                    \
    ce.w[0] = u_.x; ce.w[1] = u_.y; ce.w[2] = u_.z; ce.w[3] = u_.w; ce.w[4] = w_.x; ce.w[5] = w_.y; \
  } }
#define RC_MFMA(F1_, F2_, g_)                                                                  \
  { if (RC_ABL & 4) { part[2 * ((g_) & 1)][0] += (float)F1_[0] + (float)F2_[1]; } else           \
  {                                                                                            \
    _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) {                                         \
      f32x4& P_ = part[2 * ((g_) & 1) + nb];                                                   \
      P_ = mma16<true>(F2_, q1[2 * ((g_) >> 1) + nb], P_);                                     \
      P_ = mma16<true>(F1_, q2[2 * ((g_) >> 1) + nb], P_);                                     \
      P_ = mma16<true>(F1_, q1[2 * ((g_) >> 1) + nb], P_);                                     \
    }                                                                                          \
  } }
  // correction fragment j (held in ce) into the partial accumulators of its block
#define RC_CORR(j_)                                                                            \
  { if (RC_ABL & 4) { part[2 * ((j_) / 3)][1] += __uint_as_float(ce.w[0]); } else                \
  {                                                                                            \
    _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) {                                         \
      f32x4& P_ = part[2 * ((j_) / 3) + nb];                                                   \
      if ((j_) % 3 == 0) P_ = f16c_mma_th(ce, qh6[nb], P_);                                    \
      else if ((j_) % 3 == 1) P_ = f16c_mma_ht(ce, qt6[nb], P_);                               \
      else P_ = f16c_mma_ll(ce, ql6[nb], P_);                                                  \
    }                                                                                          \
  } }
  RC_READ(fa1, fa2, 0, 0);
  RC_CREAD(0, 0);
  f32x4 part[4];
  for (int a = a_beg; a < a_end; ++a) {
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {           // chunk (a, half = ch >> 1, block pair ch & 1) sits in ring slot ch
      RC_TLOAD((long)a * 4 + ch + 3);
      if (ch == 0) RC_PLOAD(a + 2);            // AFTER the T loads: see the wait below
#pragma unroll
      for (int i = 0; i < 4; ++i) part[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int gp = 0; gp < 4; ++gp) {         // groups 2 gp (set A) and 2 gp + 1 (set B)
        RC_READ(fb1, fb2, ch, 2 * gp + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (2 * gp < 6) { RC_CORR(2 * gp); RC_CREAD(ch, 2 * gp + 1); }
        RC_MFMA(fa1, fa2, 2 * gp);
        if (gp < 3) RC_READ(fa1, fa2, ch, 2 * gp + 2)
        else RC_READ(fa1, fa2, (ch + 1) & 3, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (2 * gp + 1 < 6) {
          RC_CORR(2 * gp + 1);
          if (2 * gp + 2 < 6) RC_CREAD(ch, 2 * gp + 2)
          else RC_CREAD((ch + 1) & 3, 0);
        }
        RC_MFMA(fb1, fb2, 2 * gp + 1);
      }
      __builtin_amdgcn_sched_barrier(0);       // the flush stays HERE: moved into the next chunk it would keep two sets
      if (RC_ABL & 8) { acc[4 * ch][0] += part[0][0] + part[1][1] + part[2][2] + part[3][3]; } else
      {                                        // of partial accumulators alive
        float pva = pst[(a & 3) * PST] * rs_a, pvb = pst[(a & 3) * PST + 16] * rs_b;
        const float pas_a = (a & 1) ? -pva : pva, pas_b = (a & 1) ? -pvb : pvb;
#pragma unroll
        for (int cb2 = 0; cb2 < 2; ++cb2)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            acc[2 * (2 * ch + cb2) + 0][t] = fmaf(pas_a, part[2 * cb2 + 0][t], acc[2 * (2 * ch + cb2) + 0][t]);
            acc[2 * (2 * ch + cb2) + 1][t] = fmaf(pas_b, part[2 * cb2 + 1][t], acc[2 * (2 * ch + cb2) + 1][t]);
          }
        // ... and is COMPLETE here: the wave-dependent wait below is control flow, and without this the compiler sinks
        // all four flushes of an `a` behind its last barrier (64 partial accumulators alive instead of 16)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(acc[4 * ch + i]));
      }
      __builtin_amdgcn_sched_barrier(0);
      // chunk i + 2 (issued one iteration ago) must have landed.  Younger than it: this iteration's three (four) T loads
      // and, for ch < 2, the p load issued right behind the T loads of ch == 0
      if (!(RC_ABL & 16)) {
        if (ch < 2) RC_WAIT(4)
        else RC_WAIT(3)
        __builtin_amdgcn_s_barrier();
      }
      asm volatile("" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef RC_TLOAD
#undef RC_PLOAD
#undef RC_WAIT
#undef RC_READ
#undef RC_CREAD
#undef RC_MFMA
#undef RC_CORR
  // the thread id is laundered so that the output addresses are computed HERE instead of being hoisted above the main
  // loop and kept alive (= spilled) across it
  int tl = tid;
  asm volatile("" : "+v"(tl));
  const int e_row = tile * 256 + (tl >> 6) * 32 + (tl & 15), e_kg = (tl >> 4) & 3;
#pragma unroll
  for (int cb = 0; cb < 8; ++cb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int row = e_row + 16 * nb;
      if (row < nrows) {
        float* op = out + (long)row * ldo + 16 * cb + 4 * e_kg;
        const f32x4 v = acc[2 * cb + nb];
        if (vec_io) *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
        else { op[0] = v[0]; op[1] = v[1]; op[2] = v[2]; op[3] = v[3]; }
      }
    }
}

// ---------------------------------------------------------------------------------------
// f16x3c form of the fused backward contraction (round 4): out1 = init1 + sum_a p[:,a] M[:,a,:] and the partial sums of
// dv[n,a] = sum_c zz[n,c] M[n,a,c] from ONE contraction M[n,a,c] = sum_b q[n,b] T[a,b,c] (bilinear_rows128_dual_kernel's
// algebra), on bilinear_rows128_ring16c_kernel's machinery (same prepared image, chunks, ring, correction terms).
//
// Loop order: the 32-column chunk index ch = (half, block pair) is the OUTER loop, `a` the inner one.  The second
// gradient needs zz[n, c] for the chunk's columns at every flush; with `a` outermost (the forward kernel's order) that
// is either 64 registers for all 128 columns -- the dual kernel pays them by giving a wave only 64 columns, i.e. 128
// rows per workgroup and twice the L2 -> LDS traffic per row, which is what bounds it -- or a reload per chunk (tried:
// +41 % vector-memory traffic, 1.42 ms against 1.18 without the loads).  With ch outermost a wave holds the zz values and
// the output accumulators of ONE chunk column range at a time (16 + 16 registers instead of 64 + 64), covers all 128
// columns of its 32 rows in four phases, and a workgroup is 256 rows like the forward kernel's.  The prepared image is
// read in (ch, a) order -- chunks are contiguous 25-KB pieces either way -- and every chunk of T is still streamed once
// per workgroup.  dv comes out as four partial sums per (row, a), one per phase: dvp[ch][a][row] (dual_finish_kernel
// adds them).  The ring slot of a chunk is its sequence number mod 4 (run-time: one address add per chunk).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void bilinear_rows128_dualc_kernel(
    const float* __restrict__ p, long ldp, const float* __restrict__ q, long ldq, const float* __restrict__ zz, long ldz,
    const uint4* __restrict__ Tq, const float* __restrict__ init, long ldi, float* __restrict__ out, long ldo,
    float* __restrict__ dvp, int dv_ld, int nrows, int NA, int tiles, int asplit, long slab_stride, int vec_io,
    const float* __restrict__ tmax) {
  constexpr int CH16 = F16C_CHUNK16;
  constexpr int PST = 8 * 64;
  __shared__ uint4 smem[4 * CH16 + 4 * PST / 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int n16 = lane & 15, kg = lane >> 4;
  const int split = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const int a_beg = (int)((long)NA * split / asplit), a_end = (int)((long)NA * (split + 1) / asplit);
  // chunks of this workgroup: sequence number n = ch * (a_end - a_beg) + (a - a_beg)
  const int row_w = tile * 256 + wave * 32;
  const int row_a = row_w + n16, row_b = row_w + 16 + n16;           // the lane's two output rows
  const long rowc_a = row_a < nrows ? row_a : nrows - 1, rowc_b = row_b < nrows ? row_b : nrows - 1;
  const int row_st = row_w + (lane & 31);                            // the row whose p this lane stages
  const long rowc_st = row_st < nrows ? row_st : nrows - 1;
  if (asplit > 1) {
    out += (long)split * slab_stride;
    if (split > 0) init = nullptr;
  }
  const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const unsigned wave_t = __builtin_amdgcn_readfirstlane(sbase + wave * 1024);
  const unsigned wave_p = __builtin_amdgcn_readfirstlane(sbase + 4 * CH16 * 16 + wave * 256);
  const bf16x8* ring0 = reinterpret_cast<const bf16x8*>(smem) + lane;
  const float* pst = reinterpret_cast<const float*>(smem + 4 * CH16) + wave * 64 + n16;
  p += (long)tile * 256 * ldp;                                       // scalar tile base + 32-bit lane offsets
  const unsigned prow_off = (unsigned)((rowc_st - (long)tile * 256) * ldp * 4);
  const unsigned t_off = (unsigned)tid * 16;

  // q[row, 32 s + 8 kg + j] of both row blocks, scaled per row: two fp16 planes qf[2 s + nb] + the three 6-bit images
  bf16x8 q1[8], q2[8];
  frag6 ql6[2], qh6[2], qt6[2];
  float rs_a, rs_b;                             // 1 / (scale of the lane's q row * scale of T)
  {
    float qv[2][32];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4* qp = reinterpret_cast<const float4*>(q + (nb ? rowc_b : rowc_a) * ldq + 32 * s + 8 * kg);
        const float4 t0 = qp[0], t1 = qp[1];
        qv[nb][8 * s + 0] = t0.x; qv[nb][8 * s + 1] = t0.y; qv[nb][8 * s + 2] = t0.z; qv[nb][8 * s + 3] = t0.w;
        qv[nb][8 * s + 4] = t1.x; qv[nb][8 * s + 5] = t1.y; qv[nb][8 * s + 6] = t1.z; qv[nb][8 * s + 7] = t1.w;
      }
    float st, it;
    pow2_scale(tmax[0], st, it);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float m = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) m = fmaxf(m, fabsf(qv[nb][j]));
      m = fmaxf(m, __shfl_xor(m, 16));          // the row's 128 values live in the four lanes n16 + 16 kg
      m = fmaxf(m, __shfl_xor(m, 32));
      float sq, iq;
      pow2_scale(m, sq, iq);
      (nb ? rs_b : rs_a) = iq * it;
#pragma unroll
      for (int j = 0; j < 32; ++j) qv[nb][j] *= sq;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = qv[nb][8 * s + j];
        split2_x8_f16(v, q1[2 * s + nb], q2[2 * s + nb]);
      }
      f16c_pack32(qv[nb], ql6[nb], qh6[nb], qt6[nb]);   // element 8 s + j <-> k = 32 s + 8 kg + j, as in the image
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // chunk (ch_, a_) with sequence number n_ -> ring slot n_ % 4; its p column -> staging slot n_ % 4 (each wave stages
  // and reads its own rows).  Past the end the last chunk is loaded again (keeps the counted waits uniform).
#define DC_TLOAD(n_, ch_, a_)                                                                  \
  {                                                                                            \
    const int ci_ = __builtin_amdgcn_readfirstlane((a_) * 4 + (ch_));   /* uniform: the DMA base must be scalar */ \
    const uint4* tb = Tq + (long)ci_ * CH16;                                                   \
    const unsigned dst = wave_t + (unsigned)__builtin_amdgcn_readfirstlane((n_) & 3) * (CH16 * 16); \
    glds_b128(tb, t_off, dst);                                                                 \
    glds_b128(tb + 512, t_off, dst + 8192);                                                    \
    glds_b128(tb + 1024, t_off, dst + 16384);                                                  \
    if (wave_u == 0) glds_b128(tb + 1536, t_off, dst + 24576);                                 \
  }
#define DC_PLOAD(n_, a_)                                                                       \
  glds_b32(p + __builtin_amdgcn_readfirstlane(a_), prow_off,                                   \
           wave_p + (unsigned)__builtin_amdgcn_readfirstlane((n_) & 3) * (PST * 4));
  // (ta, tch) = the chunk three sequence numbers ahead of the one being computed, advanced in (ch, a) order
  int ta = a_beg, tch = 0;
#define DC_ADVANCE() { if (ta + 1 < a_end) ++ta; else if (tch < 3) { ta = a_beg; ++tch; } }
  // everything except the N_ youngest vector-memory operations of this wave (wave 0: + its extra piece) has landed
#define DC_WAIT(N_) { if (wave_u == 0) wait_vmcnt<(N_) + 1>(); else wait_vmcnt<(N_)>(); }
  DC_TLOAD(0, tch, ta); DC_PLOAD(0, ta); DC_ADVANCE();
  DC_TLOAD(1, tch, ta); DC_PLOAD(1, ta); DC_ADVANCE();
  DC_TLOAD(2, tch, ta); DC_PLOAD(2, ta); DC_ADVANCE();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  bf16x8 fa1, fa2, fb1, fb2;
  frag6 ce;
  // group g = 2 s + cb2 of the chunk at ring_: the two planes of block cb2 at k-step s
#define DC_READ(F1_, F2_, ring_, g_)                                                           \
  {                                                                                            \
    const bf16x8* fp = (ring_) + ((((g_) >> 1) * 2) * 2 + ((g_) & 1)) * 64;                    \
    F1_ = fp[0];                                                                               \
    F2_ = fp[2 * 64];                                                                          \
  }
  // 6-bit fragment j = 3 cb2 + term of the chunk at ring_
#define DC_CREAD(ring_, j_)                                                                    \
  {                                                                                            \
    const unsigned char* cp = reinterpret_cast<const unsigned char*>((ring_) - lane) + 16384 + (j_) * 1536; \
    const uint4 u_ = *reinterpret_cast<const uint4*>(cp + lane * 16);                          \
    const uint2 w_ = *reinterpret_cast<const uint2*>(cp + 1024 + lane * 8);                    \
    ce.w[0] = u_.x; ce.w[1] = u_.y; ce.w[2] = u_.z; ce.w[3] = u_.w; ce.w[4] = w_.x; ce.w[5] = w_.y; \
  }
#define DC_MFMA(F1_, F2_, g_)                                                                  \
  {                                                                                            \
    _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) {                                         \
      f32x4& P_ = part[2 * ((g_) & 1) + nb];                                                   \
      P_ = mma16<true>(F2_, q1[2 * ((g_) >> 1) + nb], P_);                                     \
      P_ = mma16<true>(F1_, q2[2 * ((g_) >> 1) + nb], P_);                                     \
      P_ = mma16<true>(F1_, q1[2 * ((g_) >> 1) + nb], P_);                                     \
    }                                                                                          \
  }
#define DC_CORR(j_)                                                                            \
  {                                                                                            \
    _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) {                                         \
      f32x4& P_ = part[2 * ((j_) / 3) + nb];                                                   \
      if ((j_) % 3 == 0) P_ = f16c_mma_th(ce, qh6[nb], P_);                                    \
      else if ((j_) % 3 == 1) P_ = f16c_mma_ht(ce, qt6[nb], P_);                               \
      else P_ = f16c_mma_ll(ce, ql6[nb], P_);                                                  \
    }                                                                                          \
  }
  DC_READ(fa1, fa2, ring0, 0);
  DC_CREAD(ring0, 0);
  f32x4 part[4];
  int n = 0;                                   // sequence number of the chunk being computed
#pragma clang loop unroll(disable)
  for (int ch = 0; ch < 4; ++ch) {             // columns 32 ch .. 32 ch + 31
    // this phase's output accumulators and zz values: acc[2 cb2 + nb][t] <-> (row(nb), column 32 ch + 16 cb2 + 4 kg + t)
    f32x4 acc[4], zq[4];
#pragma unroll
    for (int cb2 = 0; cb2 < 2; ++cb2)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int row = nb ? row_b : row_a;
        const int col = 32 * ch + 16 * cb2 + 4 * kg;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (init && row < nrows) {
          const float* ip = init + (long)row * ldi + col;
          if (vec_io) v = *reinterpret_cast<const float4*>(ip);
          else v = make_float4(ip[0], ip[1], ip[2], ip[3]);
        }
        acc[2 * cb2 + nb] = f32x4{v.x, v.y, v.z, v.w};
        const float4 z4 = *reinterpret_cast<const float4*>(zz + (nb ? rowc_b : rowc_a) * ldz + col);
        zq[2 * cb2 + nb] = f32x4{z4.x, z4.y, z4.z, z4.w};
      }
    float* dvc = dvp + (long)ch * NA * dv_ld;   // this phase's partial sums: dvp[ch][a][row]
#pragma clang loop unroll(disable)
    for (int a = a_beg; a < a_end; ++a, ++n) {
      const bf16x8* ring = ring0 + (n & 3) * CH16;
      const bf16x8* ringn = ring0 + ((n + 1) & 3) * CH16;
      DC_TLOAD(n + 3, tch, ta);
      DC_PLOAD(n + 3, ta);
      DC_ADVANCE();
#pragma unroll
      for (int i = 0; i < 4; ++i) part[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int gp = 0; gp < 4; ++gp) {         // groups 2 gp (set A) and 2 gp + 1 (set B)
        DC_READ(fb1, fb2, ring, 2 * gp + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (2 * gp < 6) { DC_CORR(2 * gp); DC_CREAD(ring, 2 * gp + 1); }
        DC_MFMA(fa1, fa2, 2 * gp);
        if (gp < 3) DC_READ(fa1, fa2, ring, 2 * gp + 2)
        else DC_READ(fa1, fa2, ringn, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (2 * gp + 1 < 6) {
          DC_CORR(2 * gp + 1);
          if (2 * gp + 2 < 6) DC_CREAD(ring, 2 * gp + 2)
          else DC_CREAD(ringn, 0);
        }
        DC_MFMA(fb1, fb2, 2 * gp + 1);
      }
      __builtin_amdgcn_sched_barrier(0);       // the flush stays HERE (see bilinear_rows128_ring16c_kernel)
      {
        const float sg = (a & 1) ? -1.f : 1.f;   // part = (-1)^a M[n,a,:] (the prepared T alternates in sign)
        const float sga = sg * rs_a, sgb = sg * rs_b;
        const float pas_a = sga * pst[(n & 3) * PST], pas_b = sgb * pst[(n & 3) * PST + 16];
        float da = 0.f, db = 0.f;
#pragma unroll
        for (int cb2 = 0; cb2 < 2; ++cb2)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            acc[2 * cb2 + 0][t] = fmaf(pas_a, part[2 * cb2 + 0][t], acc[2 * cb2 + 0][t]);
            acc[2 * cb2 + 1][t] = fmaf(pas_b, part[2 * cb2 + 1][t], acc[2 * cb2 + 1][t]);
            da = fmaf(part[2 * cb2 + 0][t], zq[2 * cb2 + 0][t], da);
            db = fmaf(part[2 * cb2 + 1][t], zq[2 * cb2 + 1][t], db);
          }
        asm volatile("" : "+v"(da));             // keep the two sums out of v_pk_* (note in edgez.hip)
        asm volatile("" : "+v"(db));
        da += __shfl_xor(da, 16, 64); da += __shfl_xor(da, 32, 64);
        db += __shfl_xor(db, 16, 64); db += __shfl_xor(db, 32, 64);
        // one store for both row blocks: lanes 0..15 carry rows 0..15 (da), lanes 16..31 rows 16..31 (db)
        const float dv = (kg & 1) ? sgb * db : sga * da;
        if (kg < 2) dvc[(long)a * dv_ld + row_w + (lane & 31)] = dv;   // dv_ld >= the tile-padded row count
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(acc[i]));   // the flush is complete here (ring16c)
      }
      __builtin_amdgcn_sched_barrier(0);
      // chunk n + 2 (issued one iteration ago) must have landed.  Younger than it: the previous iteration's p load and
      // dv store, this iteration's three (four) T loads, p load and dv store.  (Anything issued between two phases --
      // output stores, zz / init loads -- sits in between and only makes the wait stricter.)  The p value of chunk n + 1
      // was requested two iterations ago, in front of chunk n + 2: it has landed too.
      DC_WAIT(7)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    // this phase's 32 output columns (the thread id is laundered: addresses computed here, not kept across the loop)
    {
      int tl = tid;
      asm volatile("" : "+v"(tl));
      const int e_row = tile * 256 + (tl >> 6) * 32 + (tl & 15), e_kg = (tl >> 4) & 3;
#pragma unroll
      for (int cb2 = 0; cb2 < 2; ++cb2)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const int row = e_row + 16 * nb;
          if (row < nrows) {
            float* op = out + (long)row * ldo + 32 * ch + 16 * cb2 + 4 * e_kg;
            const f32x4 v = acc[2 * cb2 + nb];
            if (vec_io) *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
            else { op[0] = v[0]; op[1] = v[1]; op[2] = v[2]; op[3] = v[3]; }
          }
        }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef DC_TLOAD
#undef DC_PLOAD
#undef DC_WAIT
#undef DC_ADVANCE
#undef DC_READ
#undef DC_CREAD
#undef DC_MFMA
#undef DC_CORR
}

// ---------------------------------------------------------------------------------------
// Two gradients from one contraction (hypernetwork backward, reference Hypernetworksmp.py:77-83 under autograd):
//     out1[n,c] = init1[n,c] + sum_a p[n,a] * M[n,a,c]              M[n,a,c] = sum_b q[n,b] T[a,b,c]
//     dv  [n,a] =              sum_c zz[n,c] * M[n,a,c]
// With T[a=i][b=o][c=k] = head_w[o,i,k], p = v (layer input), q = g (gradient of the layer's pre-norm output) and
// zz = z (trunk output), out1 is the gradient wrt z and dv the bilinear part of the gradient wrt v: the scale-after
// kernel already holds M[n,a,:] in its partial accumulators when it finishes an `a`, so the second gradient is one
// more multiply-add per accumulator register and a 4-lane reduction -- instead of a second 350-GFLOP launch.
// Layout differences from bilinear_rows128_ring16_kernel: zz must stay in registers (32 VGPRs per 64 columns), so
// a wave owns 32 rows x 64 columns and the eight waves of a workgroup are 4 row groups x 2 column halves (128
// rows); a ring slot holds one 32-deep k-step of BOTH column halves (two 12-KB pieces of the prepared T), every
// wave reads its own half.  dv comes out as two partial sums per row (one per column half) in dvp[half][n][a].
template <int PASSES>
__global__ __launch_bounds__(512, 2) void bilinear_rows128_dual_kernel(
    const float* __restrict__ p, long ldp, const float* __restrict__ q, long ldq, const float* __restrict__ zz, long ldz,
    const uint4* __restrict__ Tq, const float* __restrict__ init, long ldi, float* __restrict__ out, long ldo,
    float* __restrict__ dvp, int dv_ld, int nrows, int NA, int tiles, int asplit, long slab_stride, int vec_io,
    const float* __restrict__ tmax) {
  constexpr bool F16 = PASSES == 2;             // two fp16 planes, three passes; tmax = max |T|
  constexpr int NP = F16 ? 2 : 3;
  constexpr int HP = NP * 256;                  // 16-byte pieces of one (a, half, k-step) block of the prepared T
  constexpr int CH16 = 2 * HP;                  // per ring slot: 24 KB (16 KB): [half][plane][cb][lane]
  constexpr int PST = 8 * 64;
  __shared__ uint4 smem[4 * CH16 + 4 * PST / 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kg = lane >> 4;
  const int rg = wave & 3, hf = wave >> 2;      // row group, column half
  const int split = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const int a_beg = (int)((long)NA * split / asplit), a_end = (int)((long)NA * (split + 1) / asplit);
  const int row_w = tile * 128 + rg * 32;
  const int row_a = row_w + n16, row_b = row_w + 16 + n16;
  const long rowc_a = row_a < nrows ? row_a : nrows - 1, rowc_b = row_b < nrows ? row_b : nrows - 1;
  const int row_st = row_w + (lane & 31);
  const long rowc_st = row_st < nrows ? row_st : nrows - 1;
  if (asplit > 1) {
    out += (long)split * slab_stride;
    if (split > 0) init = nullptr;
  }
  const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const unsigned wave_p = __builtin_amdgcn_readfirstlane(sbase + 4 * CH16 * 16 + wave * 256);
  const bf16x8* ring = reinterpret_cast<const bf16x8*>(smem) + hf * HP + lane;
  const float* pst = reinterpret_cast<const float*>(smem + 4 * CH16) + wave * 64 + n16;
  p += (long)tile * 128 * ldp;
  const unsigned prow_off = (unsigned)((rowc_st - (long)tile * 128) * ldp * 4);
  const unsigned l_off = (unsigned)lane * 16;
  const long last_step = (long)a_end * 4 - 1;
  // the three 1-KB pieces this wave moves per k-step: piece index P = 64 wave + 512 i of the [half0 | half1] image
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // scalar copy: the LDS-DMA base pointers must be SGPRs
  const int P0 = 64 * wave_u, P1 = 64 * wave_u + 512, P2 = 64 * wave_u + 1024;
  const unsigned wave_t = __builtin_amdgcn_readfirstlane(sbase + (unsigned)P0 * 16);

  bf16x8 q1[8], q2[8], q3[F16 ? 1 : 8];
  float rs_a = 1.f, rs_b = 1.f;                 // F16: 1 / (scale of the lane's q row * scale of T)
  if constexpr (F16) {
    float qv[2][32];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4* qp = reinterpret_cast<const float4*>(q + (nb ? rowc_b : rowc_a) * ldq + 32 * s + 8 * kg);
        const float4 t0 = qp[0], t1 = qp[1];
        qv[nb][8 * s + 0] = t0.x; qv[nb][8 * s + 1] = t0.y; qv[nb][8 * s + 2] = t0.z; qv[nb][8 * s + 3] = t0.w;
        qv[nb][8 * s + 4] = t1.x; qv[nb][8 * s + 5] = t1.y; qv[nb][8 * s + 6] = t1.z; qv[nb][8 * s + 7] = t1.w;
      }
    float st, it;
    pow2_scale(tmax[0], st, it);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float m = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) m = fmaxf(m, fabsf(qv[nb][j]));
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      float sq, iq;
      pow2_scale(m, sq, iq);
      (nb ? rs_b : rs_a) = iq * it;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = qv[nb][8 * s + j] * sq;
        split2_x8_f16(v, q1[2 * s + nb], q2[2 * s + nb]);
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const float4* qp = reinterpret_cast<const float4*>(q + (nb ? rowc_b : rowc_a) * ldq + 32 * s + 8 * kg);
        const float4 t0 = qp[0], t1 = qp[1];
        const float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        split3_x8(v, q1[2 * s + nb], q2[2 * s + nb], q3[2 * s + nb]);
      }
  }
  // acc[2 cb + nb][j] = out[row(nb)][64 hf + 16 cb + 4 kg + j];  zr the same elements of zz
  f32x4 acc[8], zr[8];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int row = nb ? row_b : row_a;
      const long rowc = nb ? rowc_b : rowc_a;
      const int col = 64 * hf + 16 * cb + 4 * kg;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (init && row < nrows) {
        const float* ip = init + (long)row * ldi + col;
        if (vec_io) v = *reinterpret_cast<const float4*>(ip);
        else v = make_float4(ip[0], ip[1], ip[2], ip[3]);
      }
      acc[2 * cb + nb] = f32x4{v.x, v.y, v.z, v.w};
      const float* zp = zz + rowc * ldz + col;
      zr[2 * cb + nb] = f32x4{zp[0], zp[1], zp[2], zp[3]};
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#define DU_TLOAD(gi_)                                                                          \
  {                                                                                            \
    const long gi = (gi_) < last_step ? (gi_) : last_step;                                     \
    const long a_ = gi >> 2, s_ = gi & 3;                                                      \
    const uint4* h0 = Tq + ((a_ * 2 + 0) * 4 + s_) * HP;                                       \
    const uint4* h1 = Tq + ((a_ * 2 + 1) * 4 + s_) * HP;                                       \
    const unsigned dst = wave_t + (unsigned)((gi_) & 3) * (CH16 * 16);                         \
    glds_b128(h0 + P0, l_off, dst);                                                            \
    glds_b128(P1 < HP ? h0 + P1 : h1 + (P1 - HP), l_off, dst + 8192);                          \
    if (NP == 3) glds_b128(h1 + (P2 - HP), l_off, dst + 16384);                                \
  }
#define DU_PLOAD(a_)                                                                           \
  {                                                                                            \
    const int aa = (a_) < a_end ? (a_) : a_end - 1;                                            \
    glds_b32(p + aa, prow_off, wave_p + (unsigned)((a_) & 3) * (PST * 4));                     \
  }
  DU_PLOAD(a_beg);
  DU_PLOAD(a_beg + 1);
  DU_TLOAD((long)a_beg * 4 + 0);
  DU_TLOAD((long)a_beg * 4 + 1);
  DU_TLOAD((long)a_beg * 4 + 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  bf16x8 fa1, fa2, fa3, fb1, fb2, fb3;
#define DU_READ(F1_, F2_, F3_, slot_, cb_)                                                     \
  {                                                                                            \
    const bf16x8* fp = ring + (slot_) * (CH16) + (cb_) * 64;                                   \
    F1_ = fp[0];                                                                               \
    F2_ = fp[4 * 64];                                                                          \
    if (PASSES >= 6) F3_ = fp[8 * 64];                                                         \
  }
#define DU_MFMA1(F1_, F2_, F3_, qi_, P_)                                                       \
  {                                                                                            \
    if (PASSES >= 6) {                                                                         \
      P_ = mma16<F16>(F3_, q1[qi_], P_);                                                       \
      P_ = mma16<F16>(F1_, q3[qi_], P_);                                                       \
      P_ = mma16<F16>(F2_, q2[qi_], P_);                                                       \
    }                                                                                          \
    P_ = mma16<F16>(F2_, q1[qi_], P_);                                                         \
    P_ = mma16<F16>(F1_, q2[qi_], P_);                                                         \
    P_ = mma16<F16>(F1_, q1[qi_], P_);                                                         \
  }
#define DU_MFMA(F1_, F2_, F3_, s_, cb_)                                                        \
  {                                                                                            \
    DU_MFMA1(F1_, F2_, F3_, 2 * (s_) + 0, part[2 * (cb_) + 0])                                 \
    DU_MFMA1(F1_, F2_, F3_, 2 * (s_) + 1, part[2 * (cb_) + 1])                                 \
  }
  DU_READ(fa1, fa2, fa3, 0, 0);
  f32x4 part[8];
  // dvp[half][a][row]: the 32 rows of a wave are 128 contiguous bytes per `a` (row-major [row][a] would be 4-byte
  // stores at a 512-byte stride: 12x write amplification, measured with WRITE_SIZE).  32-bit offsets: the launcher
  // checks 2 * dv_ld * NA < 2^31
  const int dv_a = hf * NA * dv_ld + row_a, dv_b = hf * NA * dv_ld + row_b;
  for (int a = a_beg; a < a_end; ++a) {
#pragma unroll
    for (int i = 0; i < 8; ++i) part[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {                // k-step (a, s) sits in ring slot s
      DU_TLOAD((long)a * 4 + s + 3);
      if (s == 0) DU_PLOAD(a + 2);             // AFTER the T loads: see the wait below
#pragma unroll
      for (int cbp = 0; cbp < 2; ++cbp) {
        DU_READ(fb1, fb2, fb3, s, 2 * cbp + 1);
        __builtin_amdgcn_sched_barrier(0);
        DU_MFMA(fa1, fa2, fa3, s, 2 * cbp);
        if (cbp == 0) DU_READ(fa1, fa2, fa3, s, 2)
        else DU_READ(fa1, fa2, fa3, (s + 1) & 3, 0);
        __builtin_amdgcn_sched_barrier(0);
        DU_MFMA(fb1, fb2, fb3, s, 2 * cbp + 1);
      }
      if (s == 3) {
        // part = (-1)^a M[n,a,:] (the prepared T alternates in sign): scale-after flush and the second gradient
        const float pva = pst[(a & 3) * PST], pvb = pst[(a & 3) * PST + 16];
        const float sg = (a & 1) ? -1.f : 1.f;
        const float sga = F16 ? sg * rs_a : sg, sgb = F16 ? sg * rs_b : sg;   // F16: part carries the operand scales
        const float pas_a = sga * pva, pas_b = sgb * pvb;
        float da = 0.f, db = 0.f;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            acc[2 * cb + 0][t] = fmaf(pas_a, part[2 * cb + 0][t], acc[2 * cb + 0][t]);
            acc[2 * cb + 1][t] = fmaf(pas_b, part[2 * cb + 1][t], acc[2 * cb + 1][t]);
            da = fmaf(part[2 * cb + 0][t], zr[2 * cb + 0][t], da);
            db = fmaf(part[2 * cb + 1][t], zr[2 * cb + 1][t], db);
          }
        asm volatile("" : "+v"(da));             // keep the two sums out of v_pk_* (note in edgez.hip)
        asm volatile("" : "+v"(db));
        da += __shfl_xor(da, 16, 64); da += __shfl_xor(da, 32, 64);
        db += __shfl_xor(db, 16, 64); db += __shfl_xor(db, 32, 64);
        if (kg == 0) {
          dvp[dv_a + a * dv_ld] = sga * da;        // dv_ld >= the tile-padded row count: no bounds check needed
          dvp[dv_b + a * dv_ld] = sgb * db;
        }
      }
      // chunk i + 2 (issued one iteration ago) must have landed; everything issued after it may stay in flight
      // (in-order vmcnt, stores included): the previous `a`'s two dv stores + this step's T loads + the p load (s = 0),
      // the p load + T loads (s = 1), T loads (s = 2), T loads + this `a`'s dv stores (s = 3).  With vmcnt(NP)
      // everywhere the s = 3 wait, whose two youngest operations are the stores, drained the T loads issued a
      // quarter of a microsecond earlier.
      if (s == 0) wait_vmcnt<NP + 3>();
      else if (s == 1) wait_vmcnt<NP + 1>();
      else if (s == 2) wait_vmcnt<NP>();
      else wait_vmcnt<NP + 2>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef DU_TLOAD
#undef DU_PLOAD
#undef DU_READ
#undef DU_MFMA1
#undef DU_MFMA
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int row = nb ? row_b : row_a;
      if (row < nrows) {
        float* op = out + (long)row * ldo + 64 * hf + 16 * cb + 4 * kg;
        const f32x4 v = acc[2 * cb + nb];
        if (vec_io) *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
        else { op[0] = v[0]; op[1] = v[1]; op[2] = v[2]; op[3] = v[3]; }
      }
    }
}

// out1 = sum of the a-split slabs (if any); out2[n,a] = init2[n,a] + sum_h dvp[h][a][n] (nhalf partial sums: 2 column
// halves from the dual kernel, 4 column phases from the f16x3c one).
// One workgroup per 32 rows: the [128 a][32 n] pieces of dvp are read coalesced and transposed through LDS.
__global__ __launch_bounds__(256) void dual_finish_kernel(const float* __restrict__ slab, int splits, long slab_stride,
                                                          int nrows, float* __restrict__ out1, long ldo1,
                                                          const float* __restrict__ dvp, int dv_ld, int NA,
                                                          const float* __restrict__ init2, long ldi2,
                                                          float* __restrict__ out2, long ldo2, int nhalf) {
  __shared__ float tile[128][33];
  const int n0 = blockIdx.x * 32, tid = threadIdx.x;
  for (int idx = tid; idx < 128 * 32; idx += 256) {   // idx = a * 32 + n
    const int a = idx >> 5, n = idx & 31;
    const long o = (long)a * dv_ld + n0 + n;
    float t = dvp[o];
    for (int h = 1; h < nhalf; ++h) t += dvp[(long)h * NA * dv_ld + o];   // partial sums in a fixed order
    tile[a][n] = t;
  }
  __syncthreads();
  // A thread owns 16 outputs (n = k * 2 + tid / 128, c = tid % 128).  The slab loop is OUTERMOST so that the 16 loads
  // of a slab are independent (with it innermost every output paid `splits` dependent round trips: 104 us for this
  // kernel at 25 slabs of 1 280 rows); each output still adds its slabs in slab order.
  const int c = tid & 127, nh = tid >> 7;
  if (splits > 1) {
    float s[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) s[k] = 0.f;
    for (int z = 0; z < splits; ++z) {
      const float* sl = slab + (long)z * slab_stride + c;
      float v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const long row = n0 + 2 * k + nh;
        v[k] = sl[(row < nrows ? row : nrows - 1) * 128];
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) s[k] += v[k];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const long row = n0 + 2 * k + nh;
      if (row < nrows) out1[row * ldo1 + c] = s[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int n = 2 * k + nh;
    const long row = n0 + n;
    if (row < nrows) out2[row * ldo2 + c] = (init2 ? init2[row * ldi2 + c] : 0.f) + tile[c][n];
  }
}

// The same at a few hundred rows (the harness' shipped batch: 1 280 atoms, up to 32 a-splits): one thread per output,
// the slab loads of an output issued eight at a time -- the 32-row workgroups above are 40 workgroups whose threads each
// walk 32 slabs for 16 outputs (35 us per launch at 1 280 rows, sixteen launches per step); same summation order.
__global__ __launch_bounds__(256) void dual_finish_small_kernel(const float* __restrict__ slab, int splits,
                                                                long slab_stride, int nrows, float* __restrict__ out1,
                                                                long ldo1, const float* __restrict__ dvp, int dv_ld, int NA,
                                                                const float* __restrict__ init2, long ldi2,
                                                                float* __restrict__ out2, long ldo2, int nhalf) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long row = i >> 7;
  const int c = (int)(i & 127);
  if (row >= nrows) return;
  if (splits > 1) {
    const float* sl = slab + row * 128 + c;
    float s = 0.f;
    int z = 0;
    for (; z + 8 <= splits; z += 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = sl[(long)(z + k) * slab_stride];
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; z < splits; ++z) s += sl[(long)z * slab_stride];
    out1[row * ldo1 + c] = s;
  }
  const long o = (long)c * dv_ld + row;
  float t = dvp[o];
  for (int h = 1; h < nhalf; ++h) t += dvp[(long)h * NA * dv_ld + o];
  out2[row * ldo2 + c] = (init2 ? init2[row * ldi2 + c] : 0.f) + t;
}

// sgn(a) T[a] (sgn = (-1)^a if alternate, else 1) split into three bf16 planes in the ring kernels' fragment order
// (layout in the header above); element (a, b, c) of the [NA,128,128] operand is src[a*sa + b*sb + c*sc].
// F16: two fp16 planes of 2^k sgn(a) T[a], 2^k from tmax[0] = max |T| (pow2_scale), same order with 2 planes per k-step
// blockIdx.y = head of a multi-head layer: source + head * s_head, image + head * image_elems (0, 0: one operand)
template <bool F16>
__global__ void prepare_T_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int NA, long sa, long sb,
                                      long sc, int alternate, const float* __restrict__ tmax, long s_head = 0,
                                      long image_elems = 0) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)NA * 128 * 128) return;
  src += (long)blockIdx.y * s_head;
  dst += (long)blockIdx.y * image_elems;
  // thread order follows the fastest source stride so that reads coalesce
  int a = (int)(i >> 14), b, c;
  if (sc == 1) { b = (int)((i >> 7) & 127); c = (int)(i & 127); }
  else { c = (int)((i >> 7) & 127); b = (int)(i & 127); }
  float v = src[a * sa + b * sb + c * sc];
  if (alternate && (a & 1)) v = -v;
  const int half = c >> 6, cb = (c & 63) >> 4, i16 = c & 15;
  const int kh = b >> 6, s2 = (b >> 5) & 1, kg = (b & 31) >> 3, j = b & 7;
  constexpr int NP = F16 ? 2 : 3;
  const long blk = ((((long)a * 2 + half) * 2 + kh) * 2 + s2) * NP;  // planes of one k-step, each [cb][kg][i][j]
  const long in = (((long)cb * 4 + kg) * 16 + i16) * 8 + j;
  if constexpr (F16) {
    float st, it;
    pow2_scale(tmax[0], st, it);
    v *= st;
    const _Float16 h = (_Float16)v, l = (_Float16)(v - (float)h);
    _Float16* d16 = reinterpret_cast<_Float16*>(dst);
    d16[(blk + 0) * 2048 + in] = h;
    d16[(blk + 1) * 2048 + in] = l;
  } else {
    __bf16 x1, x2, x3;
    split3_bf16(v, x1, x2, x3);
    dst[(blk + 0) * 2048 + in] = x1;
    dst[(blk + 1) * 2048 + in] = x2;
    dst[(blk + 2) * 2048 + in] = x3;
  }
}

// The same three-plane image (NA = 1, no sign alternation) for SEVERAL 128 x 128 weights in one launch: the operands of the
// dense-layer kernel (edgez.hip, linear128_launch) in the 24-bit modes -- the hypernetwork's linear terms prepared their
// weight per product: 12 launches of 4.5 us per predicted-layer block and direction (round 5: one launch).
// Element (k, o) of item i is src[i][o * sc[i] + k * sb[i]]; image i at dst + i * 24576 floats.
__global__ void prepare_T_bf16_batch_kernel(WPrepBatch b, __bf16* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;            // 16384 elements per item
  const float* src = b.src[blockIdx.y];
  const long sb = b.sb[blockIdx.y], sc = b.sc[blockIdx.y];
  int bb, c;
  if (sc == 1) { bb = (i >> 7) & 127; c = i & 127; }              // thread order follows the fastest source stride
  else { c = (i >> 7) & 127; bb = i & 127; }
  const float v = src[(long)c * sc + (long)bb * sb];
  const int half = c >> 6, cb = (c & 63) >> 4, i16 = c & 15;
  const int kh = bb >> 6, s2 = (bb >> 5) & 1, kg = (bb & 31) >> 3, j = bb & 7;
  const long blk = ((((long)half) * 2 + kh) * 2 + s2) * 3;
  const long in = (((long)cb * 4 + kg) * 16 + i16) * 8 + j;
  __bf16 x1, x2, x3;
  split3_bf16(v, x1, x2, x3);
  __bf16* d = dst + (size_t)blockIdx.y * 49152;                   // 24576 floats = 49152 bf16
  d[(blk + 0) * 2048 + in] = x1;
  d[(blk + 1) * 2048 + in] = x2;
  d[(blk + 2) * 2048 + in] = x3;
}
int prepare_T_bf16_batch_launch(const WPrepBatch& b, float* dst, hipStream_t stream) {
  if (b.n <= 0) return CGAT_OK;
  hipLaunchKernelGGL(prepare_T_bf16_batch_kernel, dim3(64, b.n), dim3(256), 0, stream, b, (__bf16*)dst);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// f16x3c (mfma_bf16.h): the prepared T of the contraction kernels' 24-bit form.  One chunk = (a, column half, pair of
// 16-column blocks) = everything the forward kernel needs for 32 output columns of one `a`, 25 KB contiguous:
//   [k-step s = b/32 (4)][plane: h, l (2)][cb2 (2)][lane 64 x 16 B]          two fp16 planes of 2^k sgn(a) T[a]  (16 KB)
//   [cb2 (2)][term: 0 t6, 1 h6, 2 l6 (3)][lane 64 x 16 B | lane 64 x 8 B]    6-bit images for the correction terms (9 KB)
// lane = 16 kg + c % 16 holds, per plane fragment, b = 32 s + 8 kg + j (j = 0..7) and, per 6-bit fragment, all 32 values
// b = 32 s + 8 kg + j <-> element 8 s + j: the order in which the contraction kernels hold their row operand.
// One thread per (a, column c, k-group kg).  max |T| (tmax) lies behind the last chunk.
__device__ __forceinline__ void prepare_T_f16c_item(const float* __restrict__ src, uint4* __restrict__ dst, long i, long sa,
                                                    long sb, long sc, int alternate, float tm) {
  // thread order follows the fastest source stride where it can: c fastest when sc == 1
  int a = (int)(i >> 9), c, kg;
  if (sc == 1) { c = (int)(i & 127); kg = (int)((i >> 7) & 3); }
  else { kg = (int)(i & 3); c = (int)((i >> 2) & 127); }
  float st, it;
  pow2_scale(tm, st, it);
  if (alternate && (a & 1)) st = -st;
  float v[32];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) v[8 * s + j] = src[a * sa + (long)(32 * s + 8 * kg + j) * sb + c * sc] * st;
  const int half = c >> 6, cbp = (c & 63) >> 5, cb2 = (c >> 4) & 1, lane = 16 * kg + (c & 15);
  uint4* chunk = dst + (((long)a * 2 + half) * 2 + cbp) * F16C_CHUNK16;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = v[8 * s + j];
    bf16x8 h, l;
    split2_x8_f16(w, h, l);
    chunk[((s * 2 + 0) * 2 + cb2) * 64 + lane] = __builtin_bit_cast(uint4, h);
    chunk[((s * 2 + 1) * 2 + cb2) * 64 + lane] = __builtin_bit_cast(uint4, l);
  }
  frag6 l6, h6, t6;
  f16c_pack32(v, l6, h6, t6);
  unsigned* blk = reinterpret_cast<unsigned*>(chunk + 1024) + cb2 * 1152;
#pragma unroll
  for (int term = 0; term < 3; ++term) {
    const frag6& f = term == 0 ? t6 : (term == 1 ? h6 : l6);
    *reinterpret_cast<uint4*>(blk + term * 384 + lane * 4) = make_uint4(f.w[0], f.w[1], f.w[2], f.w[3]);
    *reinterpret_cast<uint2*>(blk + term * 384 + 256 + lane * 2) = make_uint2(f.w[4], f.w[5]);
  }
}
__global__ void prepare_T_f16c_kernel(const float* __restrict__ src, uint4* __restrict__ dst, int NA, long sa, long sb,
                                      long sc, int alternate, const float* __restrict__ tmax, int per_a = 0) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)NA * 512) return;
  prepare_T_f16c_item(src, dst, i, sa, sb, sc, alternate, tmax[per_a ? (int)(i >> 9) : 0]);   // per_a: one scale per block a
}
// out[a] = max |W[(128 a + c) * ldw + b]|, b, c < 128: one workgroup per block
__global__ __launch_bounds__(256) void absmax_blocks128_kernel(const float* __restrict__ W, long ldw, float* __restrict__ out) {
  const float* blk = W + (long)blockIdx.x * 128 * ldw;
  float m = 0.f;
  for (int i = threadIdx.x; i < 128 * 32; i += 256) {
    const float4 v = *reinterpret_cast<const float4*>(blk + (long)(i >> 5) * ldw + 4 * (i & 31));
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  __shared__ float wm[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
}

// Weight operands of the edge / dense kernels in the fp16 form: one workgroup per 128 x 128 block `a` keeps the block
// in registers, takes its largest magnitude, and writes the two planes of 2^k(a) W[a] in the bf16 kernel's order with
// two planes per k-step; max |W[a]| goes to wmax[a] behind the planes (the consumer undoes 2^k(a) per column block).
// No atomics, no second pass, one launch.
// (blockIdx.y = head of a batch of weights: per-head source and image offsets, see prepare_W_f16_heads_launch)
__global__ __launch_bounds__(256) void prepare_W_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst,
                                                            long sa, long sb, long sc, float* __restrict__ wmax,
                                                            long s_head, long image_floats) {
  __shared__ float wm[4];
  const int a = blockIdx.x, tid = threadIdx.x;
  src += (long)blockIdx.y * s_head;
  dst += (long)blockIdx.y * image_floats * 2;
  wmax += (long)blockIdx.y * image_floats;
  float v[64];
  float m = 0.f;
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    const int i = r * 256 + tid;                 // thread order follows the fastest source stride
    int b, c;
    if (sc == 1) { b = i >> 7; c = i & 127; }
    else { c = i >> 7; b = i & 127; }
    v[r] = src[a * sa + b * sb + c * sc];
    m = fmaxf(m, fabsf(v[r]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((tid & 63) == 0) wm[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
  if (tid == 0) wmax[a] = m;
  float st, it;
  pow2_scale(m, st, it);
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    const int i = r * 256 + tid;
    int b, c;
    if (sc == 1) { b = i >> 7; c = i & 127; }
    else { c = i >> 7; b = i & 127; }
    const float x = v[r] * st;
    const _Float16 h = (_Float16)x, l = (_Float16)(x - (float)h);
    const int half = c >> 6, cb = (c & 63) >> 4, i16 = c & 15;
    const int kh = b >> 6, s2 = (b >> 5) & 1, kg = (b & 31) >> 3, j = b & 7;
    const long blk = ((((long)a * 2 + half) * 2 + kh) * 2 + s2) * 2;
    const long in = (((long)cb * 4 + kg) * 16 + i16) * 8 + j;
    dst[(blk + 0) * 2048 + in] = h;
    dst[(blk + 1) * 2048 + in] = l;
  }
}
// Many 128 x 128 weights in ONE launch (a dense layer's own prepare is a single workgroup: 11 us of latency per layer,
// 48 layers per hypernetwork step): item i = (src, sb, sc) goes to dst + i * WPREP_IMAGE_FLOATS, wmax behind its planes.
__global__ __launch_bounds__(256) void prepare_W_f16_batch_kernel(WPrepBatch b, float* __restrict__ dst) {
  __shared__ float wm[4];
  const int it = blockIdx.x, tid = threadIdx.x;
  const float* src = b.src[it];
  const long sb = b.sb[it], sc = b.sc[it];
  float* img = dst + (size_t)it * WPREP_IMAGE_FLOATS;
  float v[64];
  float m = 0.f;
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    const int i = r * 256 + tid;
    int bb, c;
    if (sc == 1) { bb = i >> 7; c = i & 127; }
    else { c = i >> 7; bb = i & 127; }
    v[r] = src[bb * sb + c * sc];
    m = fmaxf(m, fabsf(v[r]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((tid & 63) == 0) wm[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
  if (tid == 0) img[16384] = m;
  float st, iv;
  pow2_scale(m, st, iv);
  _Float16* d16 = reinterpret_cast<_Float16*>(img);
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    const int i = r * 256 + tid;
    int bb, c;
    if (sc == 1) { bb = i >> 7; c = i & 127; }
    else { c = i >> 7; bb = i & 127; }
    const float x = v[r] * st;
    const _Float16 h = (_Float16)x, l = (_Float16)(x - (float)h);
    const int half = c >> 6, cb = (c & 63) >> 4, i16 = c & 15;
    const int kh = bb >> 6, s2 = (bb >> 5) & 1, kg = (bb & 31) >> 3, j = bb & 7;
    const long blk = (((long)half * 2 + kh) * 2 + s2) * 2;
    const long in = (((long)cb * 4 + kg) * 16 + i16) * 8 + j;
    d16[(blk + 0) * 2048 + in] = h;
    d16[(blk + 1) * 2048 + in] = l;
  }
}
int prepare_W_f16_batch_launch(const WPrepBatch& b, float* dst, hipStream_t stream) {
  if (b.n <= 0) return CGAT_OK;
  hipLaunchKernelGGL(prepare_W_f16_batch_kernel, dim3(b.n), dim3(256), 0, stream, b, dst);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}
int prepare_W_f16_launch(const float* src, void* dst, int NA, long sa, long sb, long sc, hipStream_t stream) {
  if (NA <= 0) return CGAT_OK;
  hipLaunchKernelGGL(prepare_W_f16_kernel, dim3(NA), dim3(256), 0, stream, src, (_Float16*)dst, sa, sb, sc,
                     (float*)dst + (size_t)NA * 16384, 0l, 0l);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}
int prepare_W_f16_heads_launch(const float* src, void* dst, int NA, long sa, long sb, long sc, int heads, long s_head,
                               long image_floats, hipStream_t stream) {
  if (NA <= 0 || heads <= 0) return CGAT_OK;
  hipLaunchKernelGGL(prepare_W_f16_kernel, dim3(NA, heads), dim3(256), 0, stream, src, (_Float16*)dst, sa, sb, sc,
                     (float*)dst + (size_t)NA * 16384, s_head, image_floats);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// out[0] = max |src[i]| (out[0] zeroed before; non-negative floats order like their bit patterns, and a maximum does
// not depend on the order it is taken in: deterministic)
__global__ void absmax_kernel(const float* __restrict__ src, long n, float* __restrict__ out) {
  float m = 0.f;
  const long n4 = n >> 2;
  const float4* s4 = reinterpret_cast<const float4*>(src);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = s4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(src[(n4 << 2) + threadIdx.x]));
  block_absmax_commit(m, out);
}
int absmax_launch(const float* src, long n, float* out, hipStream_t stream) {
  CGAT_TRY(fill_launch(out, 0.f, 1, stream));   // (a kernel, not hipMemsetAsync: see fill_launch in rowops.hip)
  if (n <= 0) return CGAT_OK;
  CGAT_CHECK_ARG((((uintptr_t)src) & 15) == 0, "absmax: source must be 16-byte aligned");
  const int blocks = (int)(cdiv(n, 4 * 256) < 512 ? cdiv(n, 4 * 256) : 512);
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, stream, src, n, out);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// Row-gathered variant for operands whose k index is a row number: element (a, b, c) = rows[gather[128 a + b]][c]
// for 128 a + b < nrows, zero beyond (the last block is padded).
// F16: two fp16 planes of 2^k rows, 2^k from emax[0] = max |rows|
template <bool F16>
__global__ void prepare_T_bf16_rows_kernel(const float* __restrict__ rows, long ld, const int* __restrict__ gather,
                                           int nrows, uint4* __restrict__ dst, int NA, const float* __restrict__ emax) {
  // one thread per 16-byte fragment piece: (a, k-step s, kg, column c) -> the 8 rows t = 128 a + 32 s + 8 kg + j of
  // column c; lanes run over c, so the eight row reads are coalesced and the three stores are 16 B at 16-B pitch
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)NA * 16 * 128) return;
  const int c = (int)(i & 127), kg = (int)((i >> 7) & 3), s = (int)((i >> 9) & 3);
  const long a = i >> 11;
  const long t0 = a * 128 + 32 * s + 8 * kg;
  bf16x8 x1, x2, x3;
  float vv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const long t = t0 + j;
    vv[j] = t < nrows ? rows[(gather ? (long)gather[t] : t) * ld + c] : 0.f;
  }
  if constexpr (F16) {
    float se, ie;
    pow2_scale(emax[0], se, ie);
#pragma unroll
    for (int j = 0; j < 8; ++j) vv[j] *= se;
    split2_x8_f16(vv, x1, x2);
  } else {
    split3_x8(vv, x1, x2, x3);
  }
  constexpr int NP = F16 ? 2 : 3;
  const int half = c >> 6, cb = (c & 63) >> 4, i16 = c & 15;
  const long blk = ((a * 2 + half) * 4 + s) * NP;         // planes of one k-step, each [cb][kg][i] x 16 bytes
  const long in = ((long)cb * 4 + kg) * 16 + i16;
  dst[(blk + 0) * 256 + in] = __builtin_bit_cast(uint4, x1);
  dst[(blk + 1) * 256 + in] = __builtin_bit_cast(uint4, x2);
  if constexpr (!F16) dst[(blk + 2) * 256 + in] = __builtin_bit_cast(uint4, x3);
}

// emax != null: the fp16 form (two planes of 2^k rows, 2^k from emax[0])
int prepare_T_bf16_rows_launch(const float* rows, long ld, const int* gather, int nrows, void* dst, int NA,
                               hipStream_t stream, const float* emax) {
  long total = (long)NA * 16 * 128;
  if (total <= 0) return CGAT_OK;
  if (emax)
    hipLaunchKernelGGL(prepare_T_bf16_rows_kernel<true>, dim3(cdiv(total, 256)), dim3(256), 0, stream, rows, ld, gather,
                       nrows, (uint4*)dst, NA, emax);
  else
    hipLaunchKernelGGL(prepare_T_bf16_rows_kernel<false>, dim3(cdiv(total, 256)), dim3(256), 0, stream, rows, ld, gather, nrows,
                     (uint4*)dst, NA, emax);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

int prepare_T_bf16_launch(const float* src, void* dst, int NA, long sa, long sb, long sc, int alternate,
                          hipStream_t stream) {
  long total = (long)NA * 128 * 128;
  if (total <= 0) return CGAT_OK;
  hipLaunchKernelGGL(prepare_T_bf16_kernel<false>, dim3(cdiv(total, 256)), dim3(256), 0, stream, src, (__bf16*)dst, NA,
                     sa, sb, sc, alternate, (const float*)nullptr);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}
// the three-plane image of `heads` operands in one launch: head h reads src + h * s_head and writes dst + h * image_floats
int prepare_T_bf16_heads_launch(const float* src, void* dst, int NA, long sa, long sb, long sc, int alternate, int heads,
                                long s_head, long image_floats, hipStream_t stream) {
  long total = (long)NA * 128 * 128;
  if (total <= 0 || heads <= 0) return CGAT_OK;
  hipLaunchKernelGGL(prepare_T_bf16_kernel<false>, dim3(cdiv(total, 256), heads), dim3(256), 0, stream, src, (__bf16*)dst,
                     NA, sa, sb, sc, alternate, (const float*)nullptr, s_head, image_floats * 2);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}
// fp16 form with the maximum already known (tmax[0], device memory): strided sources
int prepare_T_f16_scaled_launch(const float* src, void* dst, int NA, long sa, long sb, long sc, const float* tmax,
                                hipStream_t stream, int alternate) {
  long total = (long)NA * 128 * 128;
  if (total <= 0) return CGAT_OK;
  hipLaunchKernelGGL(prepare_T_bf16_kernel<true>, dim3(cdiv(total, 256)), dim3(256), 0, stream, src, (__bf16*)dst, NA,
                     sa, sb, sc, alternate, tmax);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}
// fp16 form: the whole [NA,128,128] source is contiguous (any index order); max |T| goes behind the planes
int prepare_T_f16_launch(const float* src, void* dst, int NA, long sa, long sb, long sc, int alternate,
                         hipStream_t stream) {
  long total = (long)NA * 128 * 128;
  if (total <= 0) return CGAT_OK;
  float* tmax = (float*)dst + total;
  CGAT_TRY(absmax_launch(src, total, tmax, stream));
  hipLaunchKernelGGL(prepare_T_bf16_kernel<true>, dim3(cdiv(total, 256)), dim3(256), 0, stream, src, (__bf16*)dst, NA,
                     sa, sb, sc, alternate, (const float*)tmax);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// f16x3c form (layout at prepare_T_f16c_kernel): NA * F16C_A_FLOATS floats, max |T| behind them; contiguous source
int prepare_T_f16c_launch(const float* src, void* dst, int NA, long sa, long sb, long sc, int alternate,
                          hipStream_t stream) {
  const long total = (long)NA * 128 * 128;
  if (total <= 0) return CGAT_OK;
  float* tmax = (float*)dst + (size_t)NA * F16C_A_FLOATS;
  CGAT_TRY(absmax_launch(src, total, tmax, stream));
  hipLaunchKernelGGL(prepare_T_f16c_kernel, dim3(cdiv((long)NA * 512, 256)), dim3(256), 0, stream, src, (uint4*)dst, NA,
                     sa, sb, sc, alternate, (const float*)tmax);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// The same image for a dense-layer weight given as W2 output rows of 128 contiguous inputs (row stride ldw; block a = rows
// 128 a .. 128 a + 127), e.g. a column slice of a stacked weight: the maximum is taken over exactly those elements
// (edgez.hip, edge_zc_kernel) -- and PER BLOCK a: an output block whose weights are small beside the tensor's largest keeps
// its own 24 bits.  NA * F16C_A_FLOATS floats + the NA maxima.
size_t prepare_W_f16c_rows_floats(int W2) { return (size_t)(W2 / 128) * F16C_A_FLOATS + (size_t)(W2 / 128) + 4; }
int prepare_W_f16c_rows_launch(const float* W, long ldw, int W2, void* dst, hipStream_t stream) {
  const int NA = W2 / 128;
  if (NA <= 0) return CGAT_OK;
  float* tmax = (float*)dst + (size_t)NA * F16C_A_FLOATS;
  hipLaunchKernelGGL(absmax_blocks128_kernel, dim3(NA), dim3(256), 0, stream, W, ldw, tmax);
  CGAT_LAUNCH_CHECK();
  hipLaunchKernelGGL(prepare_T_f16c_kernel, dim3(cdiv((long)NA * 512, 256)), dim3(256), 0, stream, W, (uint4*)dst, NA,
                     (long)128 * ldw, 1l, ldw, 0, (const float*)tmax, 1);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// ---- the same for several [NA,128,128] tensors at once (the predicted layers of a hypernetwork: 4 x (memset + absmax +
// prepare) = 12 launches of ~8 us each with a dispatch gap between every pair -> 2 launches).  Maxima without atomics:
// stage 1 writes one partial maximum per workgroup, every workgroup of stage 2 folds the 64 partials of its tensor.
#define TPREP_PARTS 64
__global__ void absmax_partial_batch_kernel(TPrepBatch b, long total, float* __restrict__ part) {
  const float* src = b.src[blockIdx.y];
  float m = 0.f;
  const long n4 = total >> 2;                        // total = NA * 16384: a multiple of 4
  const float4* s4 = reinterpret_cast<const float4*>(src);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = s4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  __shared__ float wm[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.y * TPREP_PARTS + blockIdx.x] = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
}
__global__ void prepare_T_f16_batch_kernel(TPrepBatch b, int NA, long sa, long sb, long sc, int alternate,
                                           const float* __restrict__ part) {
  const float* src = b.src[blockIdx.y];
  _Float16* d16 = reinterpret_cast<_Float16*>(b.dst[blockIdx.y]);
  float tm = part[blockIdx.y * TPREP_PARTS + (threadIdx.x & 63)];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tm = fmaxf(tm, __shfl_xor(tm, o, 64));
  const long total = (long)NA * 128 * 128;
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<float*>(b.dst[blockIdx.y])[total] = tm;   // behind the planes
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int a = (int)(i >> 14), bb, c;
  if (sc == 1) { bb = (int)((i >> 7) & 127); c = (int)(i & 127); }
  else { c = (int)((i >> 7) & 127); bb = (int)(i & 127); }
  float v = src[a * sa + bb * sb + c * sc];
  if (alternate && (a & 1)) v = -v;
  const int half = c >> 6, cb = (c & 63) >> 4, i16 = c & 15;
  const int kh = bb >> 6, s2 = (bb >> 5) & 1, kg = (bb & 31) >> 3, j = bb & 7;
  const long blk = ((((long)a * 2 + half) * 2 + kh) * 2 + s2) * 2;   // two planes per k-step, each [cb][kg][i][j]
  const long in = (((long)cb * 4 + kg) * 16 + i16) * 8 + j;
  float st, it;
  pow2_scale(tm, st, it);
  v *= st;
  const _Float16 h = (_Float16)v, l = (_Float16)(v - (float)h);
  d16[(blk + 0) * 2048 + in] = h;
  d16[(blk + 1) * 2048 + in] = l;
}
// the f16x3c image (prepare_T_f16c_kernel) of several tensors: blockIdx.y = tensor, its maximum folded from the partials
__global__ void prepare_T_f16c_batch_kernel(TPrepBatch b, int NA, long sa, long sb, long sc, int alternate,
                                            const float* __restrict__ part) {
  float tm = part[blockIdx.y * TPREP_PARTS + (threadIdx.x & 63)];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tm = fmaxf(tm, __shfl_xor(tm, o, 64));
  float* img = reinterpret_cast<float*>(b.dst[blockIdx.y]);
  if (blockIdx.x == 0 && threadIdx.x == 0) img[(size_t)NA * F16C_A_FLOATS] = tm;   // behind the last chunk
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)NA * 512) return;
  prepare_T_f16c_item(b.src[blockIdx.y], reinterpret_cast<uint4*>(img), i, sa, sb, sc, alternate, tm);
}
size_t bilinear_prepare_T_batch_ws_floats(int n) { return (size_t)(n > 0 ? n : 1) * TPREP_PARTS; }
// f16x3 / f16x3c modes, 128-wide interleaved layout only (returns CGAT_ERR_UNSUPPORTED otherwise: prepare one by one);
// dst[i]: bilinear_T_floats(...) floats each; part: bilinear_prepare_T_batch_ws_floats(n) floats
int bilinear_prepare_T_batch(int n, const float* const* src, float* const* dst, int n0, int n1, int n2, int perm0,
                             int perm1, int perm2, float* part, hipStream_t stream, int alternate) {
  int dims[3] = {n0, n1, n2};
  const int mode = bilinear_mode();
  static int off = -1;   // CGAT_NO_TPREP_BATCH=1 (debug): prepare the operands one by one
  if (off < 0) { const char* e = getenv("CGAT_NO_TPREP_BATCH"); off = (e && e[0] == '1') ? 1 : 0; }
  if (off || n < 1 || n > TPREP_MAX || (mode != 2 && mode != 4) || !bilinear_T_interleaved(dims[perm1], dims[perm2]))
    return CGAT_ERR_UNSUPPORTED;
  const long st[3] = {(long)n1 * n2, (long)n2, 1};
  const int NA = dims[perm0];
  const long total = (long)NA * 128 * 128;
  TPrepBatch b;
  b.n = n;
  for (int i = 0; i < n; ++i) {
    if ((((uintptr_t)src[i]) & 15) != 0) return CGAT_ERR_UNSUPPORTED;
    b.src[i] = src[i]; b.dst[i] = dst[i];
  }
  hipLaunchKernelGGL(absmax_partial_batch_kernel, dim3(TPREP_PARTS, n), dim3(256), 0, stream, b, total, part);
  CGAT_LAUNCH_CHECK();
  if (mode == 4)
    hipLaunchKernelGGL(prepare_T_f16c_batch_kernel, dim3(cdiv((long)NA * 512, 256), n), dim3(256), 0, stream, b, NA,
                       st[perm0], st[perm1], st[perm2], alternate, (const float*)part);
  else
    hipLaunchKernelGGL(prepare_T_f16_batch_kernel, dim3(cdiv(total, 256), n), dim3(256), 0, stream, b, NA, st[perm0],
                       st[perm1], st[perm2], alternate, (const float*)part);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// out[n, c] = sum_s slab[s][n][c]   (fixed order)
__global__ void slab_sum_rows_kernel(const float* __restrict__ slab, int splits, long slab_stride, int nrows,
                                     float* __restrict__ out, long ldo) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)nrows * 128) return;
  float s = 0.f;
  int z = 0;
  for (; z + 4 <= splits; z += 4) {                // four independent loads per round trip, added in slab order
    const float v0 = slab[(long)z * slab_stride + i], v1 = slab[(long)(z + 1) * slab_stride + i];
    const float v2 = slab[(long)(z + 2) * slab_stride + i], v3 = slab[(long)(z + 3) * slab_stride + i];
    s += v0; s += v1; s += v2; s += v3;
  }
  for (; z < splits; ++z) s += slab[(long)z * slab_stride + i];
  out[(i >> 7) * ldo + (i & 127)] = s;
}

// The same sum with the hypernetwork's LayerNorm(no affine) + tanh behind it (reference Hypernetworksmp.py:205-209): one
// wave per row sums the slabs into u (kept: backward needs the pre-norm values) and normalises what it holds in
// registers -- the arithmetic of layernorm_tanh_fwd_kernel (rowops.hip) on the same values, one launch and one read of
// u less per predicted layer.
__global__ void slab_sum_ln_tanh_kernel(const float* __restrict__ slab, int splits, long slab_stride, int nrows,
                                        float* __restrict__ out, long ldo, float* __restrict__ y, float eps) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= nrows) return;
  float x0 = 0.f, x1 = 0.f;
  int z = 0;
  for (; z + 4 <= splits; z += 4) {                // eight independent loads per round trip, added in slab order
    float a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = slab[(long)(z + u) * slab_stride + (long)row * 128 + lane];
      b[u] = slab[(long)(z + u) * slab_stride + (long)row * 128 + 64 + lane];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { x0 += a[u]; x1 += b[u]; }
  }
  for (; z < splits; ++z) {
    x0 += slab[(long)z * slab_stride + (long)row * 128 + lane];
    x1 += slab[(long)z * slab_stride + (long)row * 128 + 64 + lane];
  }
  out[(long)row * ldo + lane] = x0;
  out[(long)row * ldo + 64 + lane] = x1;
  float s = 0.f;
  s += x0; s += x1;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / 128;
  float v = 0.f;
  { const float d0 = x0 - mean; v += d0 * d0; const float d1 = x1 - mean; v += d1 * d1; }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const float rstd = rsqrtf(v / 128 + eps);
  y[(long)row * 128 + lane] = tanhf((x0 - mean) * rstd);
  y[(long)row * 128 + 64 + lane] = tanhf((x1 - mean) * rstd);
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
static int rows_asplit(int nrows, int rows_wg) {
  const int tiles = cdiv(nrows, rows_wg);
  // Few rows (the reference's shipped batch of 64 crystals is 1 280 atoms = 10 tiles): every workgroup streams its share
  // of T through its LDS ring whatever the row count, so the launch takes as long as ONE workgroup needs for 128 / sp
  // slices of T -- with sp <= 4 that left 216 of the 256 CUs idle and ~60-100 us per contraction.  Up to 32 ways then
  // (>= 4 values of `a` per workgroup); the slab sum over sp x [nrows, 128] floats is negligible at these sizes.
  if (tiles * 4 <= 128) {
    int sp = 256 / tiles;
    return sp > 32 ? 32 : (sp < 4 ? 4 : sp);
  }
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

int bilinear_mode();
// rows per workgroup of the kernel that will run (the split-bf16 ring kernel uses 8 waves = 256 rows)
static int rows_per_wg() { return bilinear_mode() == 0 ? 128 : 256; }

bool bilinear_T_interleaved(int NB, int NC) { return NB == 128 && NC == 128 && !force_generic(); }

// 0 = f32-input MFMA (exact fp32), 6 = 3-way bf16 split with 6 MFMA passes (24-bit operands), 3 = 3 passes,
// 2 = 2-way fp16 split with 3 passes (22-bit operands, scaled per row / per tensor),
// 4 = "f16x3c" (DEFAULT): 24-bit operands -- the fp16 passes of mode 2 plus the three 6-bit correction terms
//     (mfma_bf16.h) in the kernels that have that form (the hypernetwork contractions); every other matrix-core kernel
//     runs its six-pass bf16 form, exactly as in mode 6 (tests on `bilinear_mode() == 2` are false, `!= 0` / `!= 3` true)
static int g_bilinear_mode = -1;
int bilinear_mode() {
  if (g_bilinear_mode < 0) {
    const char* e = getenv("CGAT_BILINEAR_MODE");   // f32 | bf16x6 | bf16x3 | f16x3 | f16x3c (default)
    g_bilinear_mode = 4;
    if (e && !strcmp(e, "f32")) g_bilinear_mode = 0;
    if (e && !strcmp(e, "bf16x3")) g_bilinear_mode = 3;
    if (e && !strcmp(e, "bf16x6")) g_bilinear_mode = 6;
    if (e && !strcmp(e, "f16x3")) g_bilinear_mode = 2;
  }
  return g_bilinear_mode;
}
void bilinear_set_mode(int m) { g_bilinear_mode = (m == 6 || m == 3 || m == 2 || m == 4) ? m : 0; }
// floats of workspace the prepared T occupies (the bf16 form stores three 2-byte planes, the fp16 form two and its scale,
// the f16x3c form the fp16 form + 18 bits per element of 6-bit images)
size_t bilinear_T_floats(int NA, int NB, int NC) {
  size_t n = (size_t)NA * NB * NC;
  if (!bilinear_T_interleaved(NB, NC) || bilinear_mode() == 0) return n;
  if (bilinear_mode() == 4) return (size_t)NA * F16C_A_FLOATS + 4;
  return bilinear_mode() == 2 ? n + 4 : (n * 3 + 1) / 2;
}

size_t bilinear_T_floats_max(int NA, int NB, int NC) {   // the mode may change between a size query and the call
  const size_t n = (size_t)NA * NB * NC, a = n * 3 / 2 + 4, b = (size_t)NA * F16C_A_FLOATS + 4;
  return (bilinear_T_interleaved(NB, NC) && b > a) ? b : a;
}

size_t bilinear_rows_ws_bytes(int nrows, int NA, int NB, int NC) {
  if (!bilinear_T_interleaved(NB, NC)) return 0;
  int sp = rows_asplit(nrows, rows_per_wg());
  return sp > 1 ? ws_round((size_t)sp * nrows * 128, 4) : 0;
}

// T must come from bilinear_prepare_T (interleaved columns iff bilinear_T_interleaved(NB, NC))
// ---- fused pair of contractions (see bilinear_rows128_dual_kernel); widths 128, split-bf16 modes only ----
bool bilinear_dual_fast(int NA, int NB, int NC) {
  return bilinear_mode() != 0 && NA == 128 && NB == 128 && NC == 128 && !force_generic();
}
static int dual_dv_ld(int nrows) { return cdiv(nrows, 256) * 256; }   // rows padded to whole tiles (128 or 256 rows)
static int dual_asplit_max(int nrows) {   // the f16x3c form runs 256-row workgroups, the others 128-row ones
  const int a = rows_asplit(nrows, 128), b = rows_asplit(nrows, 256);
  return a > b ? a : b;
}
size_t bilinear_dual_ws_bytes(int nrows) {
  const int sp = dual_asplit_max(nrows);
  return ws_round((size_t)4 * dual_dv_ld(nrows) * 128 + (sp > 1 ? (size_t)sp * nrows * 128 : 0), 4);   // <= 4 dv partials
}
// T: bilinear_prepare_T of the [NA,128,128] operand.  out1 = init1 + sum_a p[:,a] M[:,a,:], out2 = init2 + M . zz
int bilinear_dual_launch(const float* p, long ldp, const float* q, long ldq, const float* zz, long ldz, const float* T,
                         const float* init1, long ldi1, float* out1, long ldo1, const float* init2, long ldi2,
                         float* out2, long ldo2, int nrows, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (nrows <= 0) return CGAT_OK;
  CGAT_CHECK_ARG((ldq % 4) == 0 && (((uintptr_t)q) & 15) == 0 && (((uintptr_t)T) & 15) == 0 && ldp < (1l << 22) &&
                     (long)dual_dv_ld(nrows) * 512 < (1l << 31),
                 "bilinear_dual: q and T must be 16-byte aligned with ldq %% 4 == 0, nrows < 2^23");
  const bool c256 = bilinear_mode() == 4;   // the f16x3c form: 256-row workgroups, dv complete per row
  const int tiles = cdiv(nrows, c256 ? 256 : 128);
  const int sp = rows_asplit(nrows, c256 ? 256 : 128);
  if (!ws || ws_bytes < bilinear_dual_ws_bytes(nrows)) {
    cgat_set_error("bilinear_dual: workspace too small (%zu < %zu)", ws_bytes, bilinear_dual_ws_bytes(nrows));
    return CGAT_ERR_WORKSPACE;
  }
  const int dv_ld = dual_dv_ld(nrows);
  float* dvp = (float*)ws;
  float* slab = dvp + (size_t)4 * dv_ld * 128;
  float* dst = sp > 1 ? slab : out1;
  const long dld = sp > 1 ? 128 : ldo1, stride = sp > 1 ? (long)nrows * 128 : 0;
  const int vec_io = ((dld % 4) == 0 && (((uintptr_t)dst) & 15) == 0 &&
                      (!init1 || ((ldi1 % 4) == 0 && (((uintptr_t)init1) & 15) == 0))) ? 1 : 0;
  {
    CGAT_PROF("bilinear_dual", stream);
    const float* tmax = T + (size_t)128 * 128 * 128;   // f16x3: max |T| behind the two planes
    if (bilinear_mode() == 6)
      hipLaunchKernelGGL((bilinear_rows128_dual_kernel<6>), dim3(tiles * sp), dim3(512), 0, stream, p, ldp, q, ldq, zz, ldz,
                         (const uint4*)T, init1, ldi1, dst, dld, dvp, dv_ld, nrows, 128, tiles, sp, stride, vec_io, tmax);
    else if (c256) {   // prepare_T_f16c_kernel's image, max |T| behind it
      CGAT_CHECK_ARG((ldz % 4) == 0 && (((uintptr_t)zz) & 15) == 0 && ldz < (1l << 22),
                     "bilinear_dual: zz must be 16-byte aligned with ldz %% 4 == 0");
      hipLaunchKernelGGL(bilinear_rows128_dualc_kernel, dim3(tiles * sp), dim3(512), 0, stream, p, ldp, q, ldq, zz, ldz,
                         (const uint4*)T, init1, ldi1, dst, dld, dvp, dv_ld, nrows, 128, tiles, sp, stride, vec_io,
                         T + (size_t)128 * F16C_A_FLOATS);
    } else if (bilinear_mode() == 2)
      hipLaunchKernelGGL((bilinear_rows128_dual_kernel<2>), dim3(tiles * sp), dim3(512), 0, stream, p, ldp, q, ldq, zz, ldz,
                         (const uint4*)T, init1, ldi1, dst, dld, dvp, dv_ld, nrows, 128, tiles, sp, stride, vec_io, tmax);
    else
      hipLaunchKernelGGL((bilinear_rows128_dual_kernel<3>), dim3(tiles * sp), dim3(512), 0, stream, p, ldp, q, ldq, zz, ldz,
                         (const uint4*)T, init1, ldi1, dst, dld, dvp, dv_ld, nrows, 128, tiles, sp, stride, vec_io, tmax);
    CGAT_LAUNCH_CHECK();
  }
  if (nrows <= 8192)
    hipLaunchKernelGGL(dual_finish_small_kernel, dim3(cdiv((long)nrows * 128, 256)), dim3(256), 0, stream, slab, sp, stride,
                       nrows, out1, ldo1, dvp, dv_ld, 128, init2, ldi2, out2, ldo2, c256 ? 4 : 2);
  else
    hipLaunchKernelGGL(dual_finish_kernel, dim3(cdiv(nrows, 32)), dim3(256), 0, stream, slab, sp, stride, nrows, out1, ldo1,
                       dvp, dv_ld, 128, init2, ldi2, out2, ldo2, c256 ? 4 : 2);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// ln_out (optional, NC = 128): y = tanh(LayerNorm(out)) [nrows,128] contiguous, fused into the slab sum when there is one
int bilinear_rows_launch(const float* p, long ldp, const float* q, long ldq, const float* T, const float* init,
                         long ldi, float* out, long ldo, int nrows, int NA, int NB, int NC, void* ws, size_t ws_bytes,
                         hipStream_t stream, float* ln_out, float ln_eps) {
  if (nrows <= 0) return CGAT_OK;
  bool ln_done = false;
  if (bilinear_T_interleaved(NB, NC)) {
    if (!rows_fast(q, ldq, NB, NC) || (((uintptr_t)T) & 15) != 0) {
      cgat_set_error("bilinear_rows: q and T must be 16-byte aligned with ldq %% 4 == 0 at width 128");
      return CGAT_ERR_ARG;
    }
    const int tiles = cdiv(nrows, 128);
    const int sp = rows_asplit(nrows, rows_per_wg());
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
    if (bilinear_mode() != 0) {
      CGAT_PROF("bilinear_rows", stream);
      const int tiles2 = cdiv(nrows, 256);
      const int vec_io = ((ldo % 4) == 0 && (dld % 4) == 0 && (((uintptr_t)dst) & 15) == 0 &&
                          (!init || ((ldi % 4) == 0 && (((uintptr_t)init) & 15) == 0))) ? 1 : 0;
      if (ldp >= (1l << 22)) {   // the kernel addresses p with 32-bit lane offsets inside a 256-row tile
        cgat_set_error("bilinear_rows: ldp %ld too large", ldp);
        return CGAT_ERR_ARG;
      }
      const float* tmax = T + (size_t)NA * 128 * 128;   // f16x3: max |T| behind the two planes (f16x3c: behind its image)
      if (bilinear_mode() == 6)
        hipLaunchKernelGGL((bilinear_rows128_ring16_kernel<6>), dim3(tiles2 * sp), dim3(512), 0, stream, p, ldp, q, ldq,
                           (const uint4*)T, init, ldi, dst, dld, nrows, NA, tiles2, sp, stride, vec_io, tmax);
      else if (bilinear_mode() == 4)
        hipLaunchKernelGGL(bilinear_rows128_ring16c_kernel, dim3(tiles2 * sp), dim3(512), 0, stream, p, ldp, q, ldq,
                           (const uint4*)T, init, ldi, dst, dld, nrows, NA, tiles2, sp, stride, vec_io,
                           T + (size_t)NA * F16C_A_FLOATS);
#ifdef CGAT_DEV_ABLATIONS   // timing-only variants (wrong results): only in builds made for tools/ring_ablation.py
      else if (bilinear_mode() == 2 && getenv("CGAT_RING_ABL")) {
#define RG_ABL(A_) hipLaunchKernelGGL((bilinear_rows128_ring16_kernel<2, A_>), dim3(tiles2 * sp), dim3(512), 0, stream, p, ldp, q, ldq, (const uint4*)T, init, ldi, dst, dld, nrows, NA, tiles2, sp, stride, vec_io, tmax)
        switch (atoi(getenv("CGAT_RING_ABL"))) {
          case 1: RG_ABL(1); break; case 2: RG_ABL(2); break; case 3: RG_ABL(3); break; case 4: RG_ABL(4); break;
          case 7: RG_ABL(7); break; case 8: RG_ABL(8); break; case 15: RG_ABL(15); break; default: RG_ABL(0); break;
        }
#undef RG_ABL
      }
#endif
      else if (bilinear_mode() == 2)
        hipLaunchKernelGGL((bilinear_rows128_ring16_kernel<2>), dim3(tiles2 * sp), dim3(512), 0, stream, p, ldp, q, ldq,
                           (const uint4*)T, init, ldi, dst, dld, nrows, NA, tiles2, sp, stride, vec_io, tmax);
      else
        hipLaunchKernelGGL((bilinear_rows128_ring16_kernel<3>), dim3(tiles2 * sp), dim3(512), 0, stream, p, ldp, q, ldq,
                           (const uint4*)T, init, ldi, dst, dld, nrows, NA, tiles2, sp, stride, vec_io, tmax);
    } else {
      CGAT_PROF("bilinear_rows", stream);
      static int variant = -1;  // dev knob: CGAT_BIL_VARIANT = <JS><FLUSH>, e.g. 161, 162, 322, 324
      if (variant < 0) {
        const char* ev = getenv("CGAT_BIL_VARIANT");
        variant = ev ? atoi(ev) : 162;
      }
#ifdef CGAT_DEV_ABLATIONS
      const char* ev2 = getenv("CGAT_BIL_VARIANT_LIVE");  // re-read on every call (A/B in one process)
      const int v = ev2 ? atoi(ev2) : variant;
#else
      const int v = variant >= 900 ? 162 : variant;       // 90x = timing-only ablations: dev builds only
#endif
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
    if (sp > 1 && ln_out) {
      hipLaunchKernelGGL(slab_sum_ln_tanh_kernel, dim3(cdiv(nrows, 4)), dim3(256), 0, stream, (const float*)ws, sp, stride,
                         nrows, out, ldo, ln_out, ln_eps);
      CGAT_LAUNCH_CHECK();
      ln_done = true;
    } else if (sp > 1) {
      hipLaunchKernelGGL(slab_sum_rows_kernel, dim3(cdiv((long)nrows * 128, 256)), dim3(256), 0, stream,
                         (const float*)ws, sp, stride, nrows, out, ldo);
      CGAT_LAUNCH_CHECK();
    }
  } else if (!force_generic()) {
    // Widths other than 128: out = init + (p (x) q) T, the row-wise outer product [nrows, NA * NB] formed in the operand
    // loader of the fp32 engine (gemm.hip) and T [NA * NB, NC] as it lies: 0.6 ms at 83 340 rows of width 64 where the
    // one-thread-per-output kernel below took 5.4 (and 850 ms at width 256)
    if (init && (init != out || ldi != ldo)) CGAT_TRY(copy2d_launch(init, ldi, out, ldo, nrows, NC, stream));
    GemmParams g = gemm_params(nrows, NC, NA * NB, q, ldq, T, NC, out, ldo);
    g.b_kmajor = 1;
    g.a_outer = p; g.ld_a_outer = ldp; g.outer_n = NB;
    g.beta = init ? 1.f : 0.f;
    CGAT_TRY(gemm_launch(g, nullptr, 0, stream));
  } else {
    CGAT_PROF("bilinear_rows_generic", stream);
    hipLaunchKernelGGL(bilinear_rows_generic_kernel, dim3(cdiv((long)nrows * NC, 256)), dim3(256), 0, stream, p, ldp,
                       q, ldq, T, init, ldi, out, ldo, nrows, NA, NB, NC);
    CGAT_LAUNCH_CHECK();
  }
  if (ln_out && !ln_done) {
    if (NC != 128 || ldo != 128) {
      cgat_set_error("bilinear_rows: the LayerNorm epilogue needs 128 contiguous columns");
      return CGAT_ERR_ARG;
    }
    return layernorm_tanh_fwd_launch(out, ln_out, nrows, 128, ln_eps, stream);
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

// ---------------------------------------------------------------------------------------
// Split-bf16 weight gradient:  out[a,b,c] = sum_n p[n,a] q[n,b] r[n,c]  with the contraction
// index n on the MFMA k axis.  Pre-passes (once per call, ~0.1 ms at N = 83k):
//   pT, qT [128][Np]   transposes (Np = N rounded up to 32, zero padded): an A fragment needs 8
//                      consecutive n for one b
//   Rq [Np/16][piece][cb][h][r][j]   r split into three bf16 planes in B-fragment order
// Workgroup = 8 waves = two `a` values (waves 0-3 / 4-7) x 128 b x 128 c; wave = 32 b x 128 c.
// The A fragment (p*q, 8 values per lane) is split on the fly; six MFMA passes, smallest first.
// ---------------------------------------------------------------------------------------
// mx (optional): max |in| is folded into it (zeroed before; f16x3 mode)
__global__ void transpose_pad_kernel(const float* __restrict__ in, long ld, int rows, int cols, int rows_pad,
                                     float* __restrict__ out, float* __restrict__ mx) {  // out[c][n] = in[n][c], n < rows_pad (zeros beyond rows)
  __shared__ float t[32][33];
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 8 rows per pass
  float m = 0.f;
  for (int i = ty; i < 32; i += 8) {
    int n = n0 + i, c = c0 + tx;
    const float v = (n < rows && c < cols) ? in[(long)n * ld + c] : 0.f;
    t[i][tx] = v;
    m = fmaxf(m, fabsf(v));
  }
  if (mx) block_absmax_commit(m, mx);
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    int c = c0 + i, n = n0 + tx;
    if (c < cols && n < rows_pad) out[(long)c * rows_pad + n] = t[tx][i];
  }
}

// F16: two fp16 planes of 2^k r, 2^k from mx[2] = max |r| (f16x3 mode; mx = {max|p|, max|q|, max|r|})
template <bool F16>
__global__ void split_rows_bf16_kernel(const float* __restrict__ r, long ldr, int rows, int rows_pad,
                                       __bf16* __restrict__ dst, const float* __restrict__ mx) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows_pad * 128) return;
  const int n = (int)(i >> 7), c = (int)(i & 127);
  float v = n < rows ? r[(long)n * ldr + c] : 0.f;
  const int s = n >> 4, h = (n >> 3) & 1, j = n & 7, cb = c >> 5, rr = c & 31;
  constexpr int NP = F16 ? 2 : 3;
  const long base = (long)s * NP * 4;
  if constexpr (F16) {
    float sr, ir;
    pow2_scale(mx[2], sr, ir);
    v *= sr;
    const _Float16 x1 = (_Float16)v, x2 = (_Float16)(v - (float)x1);
    _Float16* d16 = reinterpret_cast<_Float16*>(dst);
    d16[((((base + 0 * 4 + cb) * 2 + h) * 32 + rr) * 8) + j] = x1;
    d16[((((base + 1 * 4 + cb) * 2 + h) * 32 + rr) * 8) + j] = x2;
  } else {
    __bf16 x1, x2, x3;
    split3_bf16(v, x1, x2, x3);
    dst[((((base + 0 * 4 + cb) * 2 + h) * 32 + rr) * 8) + j] = x1;
    dst[((((base + 1 * 4 + cb) * 2 + h) * 32 + rr) * 8) + j] = x2;
    dst[((((base + 2 * 4 + cb) * 2 + h) * 32 + rr) * 8) + j] = x3;
  }
}

// out[0] = max |t[n, c]| over a [rows, 128] view with row stride ld (16-byte aligned rows; out zeroed before)
__global__ void absmax_rows128_kernel(const float* __restrict__ t, long ld, int rows, float* __restrict__ out) {
  const int c4 = threadIdx.x & 31, r0 = threadIdx.x >> 5;      // 8 rows of 32 float4 per workgroup pass
  float m = 0.f;
  for (long n = (long)blockIdx.x * 8 + r0; n < rows; n += (long)gridDim.x * 8) {
    const float4 v = *reinterpret_cast<const float4*>(t + n * ld + 4 * c4);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  block_absmax_commit(m, out);
}
// max |t[n, 0..127]| over rows of stride ld folded into out[0] (NOT zeroed here)
int absmax_rows128_launch(const float* t, long ld, int rows, float* out, hipStream_t stream) {
  if (rows <= 0) return CGAT_OK;
  CGAT_CHECK_ARG((ld % 4) == 0 && (((uintptr_t)t) & 15) == 0, "absmax_rows128: rows must be 16-byte aligned");
  hipLaunchKernelGGL(absmax_rows128_kernel, dim3(rows < 8192 ? (rows + 7) / 8 : 1024), dim3(256), 0, stream, t, ld, rows, out);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// Tried and dropped (round 1): the same kernel on v_mfma_f32_16x16x32_bf16 with the product split of the next k-step
// interleaved between the MFMAs and the flush through slab tiles -- 1.74 ms vs 1.66 ms for this form.  The kernel is
// bound by the SIMD's vector ISSUE port, not by the matrix pipe: per 32-row step a wave issues ~170 VALU instructions
// for the 16 product splits (4 cycles each) and its MFMAs hold the port for 8 cycles apiece; 96 16x16x32 MFMAs
// (768 cycles of issue) leave less room beside them than 48 32x32x16 ones (384), so here the 32x32x16 shape wins
// although it clocks lower.  Fewer VALU instructions per split is the remaining lever.
template <int PASSES, int ABL = 0>   // ABL (timing only, wrong results): 1 no global loads, 2 no split, 4 no barrier
__global__ __launch_bounds__(512, 2) void bilinear_wgrad128_bf16_kernel(const float* __restrict__ pT,
                                                                        const float* __restrict__ qT,
                                                                        const uint4* __restrict__ Rq,
                                                                        float* __restrict__ slab, int rows_pad,
                                                                        int rows_per_split, int NA,
                                                                        const float* __restrict__ mx) {
  constexpr bool F16 = PASSES == 2;        // two fp16 planes, three passes; mx = {max|p|, max|q|, max|r|}
  constexpr int NP = F16 ? 2 : 3;
  constexpr int KS = 2;                    // k-steps (16 rows each) per chunk
  constexpr int RCH = KS * NP * 256;       // 16-byte pieces of Rq per chunk
  constexpr int QP = 36;                   // pitch (floats) of the q^T tile: conflict-free 16-byte reads
  __shared__ uint4 Rs[2][RCH];
  __shared__ __attribute__((aligned(16))) float Qs[2][128 * QP];
  __shared__ __attribute__((aligned(16))) float Ps[2][2 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hi = lane >> 5;
  const int grp = wave >> 2, wb = wave & 3;
  // XCD-aware placement: workgroups are dealt to the 8 XCDs round-robin by linear id, and every workgroup of a row
  // split streams the same q^T / r tiles.  With (x, y) = (a pair, split) in natural order each XCD's L2 would serve
  // all splits' streams at once (27 MB each, 4 MB of L2); remapped, the workgroups sharing an XCD share ONE stream
  // and run in near lockstep, so the tiles are fetched into that L2 once instead of once per workgroup.
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int nx = gridDim.x, ny = gridDim.y, total = nx * ny;
    if (total % 8 == 0 && 8 % ny == 0) {
      const int lin = by * nx + bx, xcd = lin & 7, w = lin >> 3;   // w-th workgroup of its XCD
      const int xps = 8 / ny;                                      // XCDs per split
      by = xcd / xps;
      bx = (xcd % xps) * (total / 8) + w;                          // a-pair index inside the split
      if (bx >= nx) { bx = blockIdx.x; by = blockIdx.y; }          // irregular grid: natural order
    }
  }
  const int a0 = bx * 2, z = by;
  const int nbeg = z * rows_per_split;
  const int nend = min(rows_pad, nbeg + rows_per_split);
  const int nchunks = (nend - nbeg) / 32;   // rows_per_split and rows_pad are multiples of 32

  f32x16 acc[4], tot[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int t = 0; t < 16; ++t) { acc[cb][t] = 0.f; tot[cb][t] = 0.f; }

  // F16: the products p*q are brought into fp16 range by 2^k from max|p| max|q| (folded into the staged p); the sums
  // come out scaled by that and by r's scale
  float spq = 1.f, inv_all = 1.f;
  if constexpr (F16) {
    float ipq, sr, ir;
    pow2_scale(mx[0] * mx[1], spq, ipq);
    pow2_scale(mx[2], sr, ir);
    inv_all = ipq * ir;
  }
  uint4 pr0, pr1, pr2;
  float4 pq0, pq1;
  float pp = 0.f;
  const int qb0 = tid >> 3, qn4 = tid & 7;                 // q^T pieces: rows qb0 and qb0 + 64
#define WG_GLOAD(n0_)                                                                   \
  {                                                                                     \
    const uint4* rb = Rq + (long)((n0_) >> 4) * (NP * 256) + tid;                       \
    pr0 = rb[0]; pr1 = rb[512];                                                         \
    if (NP == 3) pr2 = rb[1024];                                                        \
    pq0 = *reinterpret_cast<const float4*>(qT + (long)qb0 * rows_pad + (n0_) + 4 * qn4);        \
    pq1 = *reinterpret_cast<const float4*>(qT + (long)(qb0 + 64) * rows_pad + (n0_) + 4 * qn4); \
    if (tid < 64) {                                                                     \
      const int aa = a0 + (tid >> 5);                                                   \
      pp = aa < NA ? pT[(long)aa * rows_pad + (n0_) + (tid & 31)] : 0.f;                \
      if (F16) pp *= spq;                                                               \
      if (((((n0_) - nbeg) >> 5) >> 4) & 1) pp = -pp; /* odd flush groups accumulate -p*q*r */ \
    }                                                                                   \
  }
#define WG_LSTORE(buf_)                                                                 \
  {                                                                                     \
    uint4* lb = &Rs[buf_][tid];                                                         \
    lb[0] = pr0; lb[512] = pr1;                                                         \
    if (NP == 3) lb[1024] = pr2;                                                        \
    *reinterpret_cast<float4*>(&Qs[buf_][qb0 * QP + 4 * qn4]) = pq0;                    \
    *reinterpret_cast<float4*>(&Qs[buf_][(qb0 + 64) * QP + 4 * qn4]) = pq1;             \
    if (tid < 64) Ps[buf_][tid] = pp;                                                   \
  }
  if (nchunks > 0) {
    WG_GLOAD(nbeg);
    WG_LSTORE(0);
  }
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int cur = c & 1;
    if constexpr (!(ABL & 1)) { if (c + 1 < nchunks) WG_GLOAD(nbeg + (c + 1) * 32); }
    if ((c & 15) == 0 && c > 0) {   // two-level summation over the long row dimension (512-row partials);
      // groups alternate in sign (see bilinear_rows128_bf16_kernel: cancels the bf16 MFMA's floor bias)
      const float sg = (((c >> 4) - 1) & 1) ? -1.f : 1.f;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        tot[cb] += acc[cb] * sg;
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[cb][t] = 0.f;
      }
    }
    const bf16x8* bs = reinterpret_cast<const bf16x8*>(&Rs[cur][hi * 32 + r]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4* q4 = reinterpret_cast<const float4*>(&Qs[cur][(wb * 32 + r) * QP + ks * 16 + 8 * hi]);
      const float4* p4 = reinterpret_cast<const float4*>(&Ps[cur][grp * 32 + ks * 16 + 8 * hi]);
      const float4 qa = q4[0], qb = q4[1], pa = p4[0], pb = p4[1];
      const float av[8] = {pa.x * qa.x, pa.y * qa.y, pa.z * qa.z, pa.w * qa.w,
                           pb.x * qb.x, pb.y * qb.y, pb.z * qb.z, pb.w * qb.w};
      bf16x8 a1, a2v, a3;
      if constexpr (ABL & 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { a1[j] = (__bf16)av[j]; }
        a2v = a1; a3 = a1;
      } else if constexpr (F16) {
        split2_x8_f16(av, a1, a2v);
      } else {
        split3_x8(av, a1, a2v, a3);
      }
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const bf16x8 b1 = bs[((ks * NP + 0) * 4 + cb) * 64];
        const bf16x8 b2 = bs[((ks * NP + 1) * 4 + cb) * 64];
        if constexpr (F16) {
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a2v), __builtin_bit_cast(f16x8, b1), acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, b2), acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, b1), acc[cb], 0, 0, 0);
        } else {
          if (PASSES >= 6) {
            const bf16x8 b3 = bs[((ks * 3 + 2) * 4 + cb) * 64];
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2v, b2, acc[cb], 0, 0, 0);
          }
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2v, b1, acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[cb], 0, 0, 0);
        }
      }
    }
    if (c + 1 < nchunks) WG_LSTORE(cur ^ 1);
    if constexpr (!(ABL & 4)) __syncthreads();
  }
#undef WG_GLOAD
#undef WG_LSTORE
  const int a = a0 + grp;
  if (a >= NA) return;
  float* o = slab + ((long)z * NA + a) * 128 * 128;
  const float sg_last = (nchunks > 0 && (((nchunks - 1) >> 4) & 1)) ? -1.f : 1.f;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    acc[cb] = acc[cb] * sg_last + tot[cb];
    if constexpr (F16) acc[cb] = acc[cb] * inv_all;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int b = wb * 32 + (t & 3) + 8 * (t >> 2) + 4 * hi;
      o[(long)b * 128 + cb * 32 + r] = acc[cb][t];
    }
  }
}

// ---------------------------------------------------------------------------------------
// f16x3 weight gradient, BATCHED over predicted layers and software-pipelined (round 2).
//
// Same arithmetic and data layout as bilinear_wgrad128_bf16_kernel<2> above (pT, qT transposes, Rq = two fp16 planes of
// 2^k r in B-fragment order, products p*q split on the fly, 512-row partial sums with alternating sign), two changes:
//  * one launch covers every (layer, row split, a pair) unit: the four predicted layers of a hypernetwork give
//    4 x 64 = 256 units = one workgroup per CU with NO row split, so the slabs, their summation pass and three of
//    the four launches disappear (the per-layer launches of round 1 had to split the rows four ways to fill the chip,
//    or ran on half of it beside another stream).  A workgroup loops over units when the grid is smaller.
//  * the loop is a software pipeline in source order, pinned with sched_barrier: the round-1 kernel ran, per 16-row
//    step and wave, [4 LDS reads -> 24 VALU of product split -> 12 MFMAs] back to back, and because the two waves of
//    a SIMD leave the chunk barrier together they both sat in the read + split phase at the same time with the matrix
//    pipe idle (measured 0.50 of the MFMA issue rate).  Here the A fragments of step s+1 are produced in the issue
//    slots an MFMA leaves free (it holds the vector port for 8 of its 32 cycles) while the MFMAs of step s run, the
//    B fragments are double-buffered one column block ahead, and a three-slot LDS ring lets the fragments of the next
//    chunk be fetched BEFORE the chunk barrier, so no wave starts a chunk with an empty matrix pipe.
// ---------------------------------------------------------------------------------------
// (WgradBatchDesc / WgradPrepDesc: wgrad_batch.h)

// mx[4 * layer + which] = max |tensor|, which 0 / 1 / 2 = p / q / r  (mx zeroed before)
__global__ void absmax_rows_batch_kernel(WgradPrepDesc d, long ldp, long ldq, long ldr, int rows, int NA,
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
  block_absmax_commit(m, mx + 4 * layer + which);
}

// z = 2 * layer + which: which 0 -> pT [128][rows_pad] = (p * 2^k * sign(n))^T, 2^k from max|p| max|q| (the products
// p*q must fit fp16) and sign(n) = -1 in the odd 512-row groups of n's row split (the kernel's partial sums alternate
// in sign); which 1 -> qT = q^T.  Rows beyond `rows` and columns beyond NA are zero.
__global__ void transpose_pad_batch_kernel(WgradPrepDesc d, long ldp, long ldq, int rows, int NA, int rows_pad,
                                           int rows_per_split, float* __restrict__ pT, float* __restrict__ qT, long sT,
                                           const float* __restrict__ mx) {
  __shared__ float t[32][33];
  const int layer = blockIdx.z >> 1, which = blockIdx.z & 1;
  const float* in = which ? d.q[layer] : d.p[layer];
  const long ld = which ? ldq : ldp;
  const int cols = which ? 128 : NA;
  float* out = (which ? qT : pT) + (long)layer * sT;
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  float scale = 1.f;
  if (!which) {
    float ipq;
    pow2_scale(mx[4 * layer] * mx[4 * layer + 1], scale, ipq);
    if ((((n0 % rows_per_split) >> 5) >> 4) & 1) scale = -scale;   // a 32-row tile never straddles a 512-row group
  }
  for (int i = ty; i < 32; i += 8) {
    const int n = n0 + i, c = c0 + tx;
    t[i][tx] = (n < rows && c < cols) ? in[(long)n * ld + c] * scale : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, n = n0 + tx;
    if (c < 128 && n < rows_pad) out[(long)c * rows_pad + n] = t[tx][i];
  }
}
__global__ void split_rows_f16_batch_kernel(WgradPrepDesc d, long ldr, int rows, int rows_pad, _Float16* __restrict__ dst,
                                            long sR_halfs, const float* __restrict__ mx) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows_pad * 128) return;
  const int layer = blockIdx.y;
  const int n = (int)(i >> 7), c = (int)(i & 127);
  float v = n < rows ? d.r[layer][(long)n * ldr + c] : 0.f;
  const int s = n >> 4, h = (n >> 3) & 1, j = n & 7, cb = c >> 5, rr = c & 31;
  const long base = (long)s * 2 * 4;
  float sr, ir;
  pow2_scale(mx[4 * layer + 2], sr, ir);
  v *= sr;
  const _Float16 x1 = (_Float16)v, x2 = (_Float16)(v - (float)x1);
  _Float16* d16 = dst + (long)layer * sR_halfs;
  d16[((((base + 0 * 4 + cb) * 2 + h) * 32 + rr) * 8) + j] = x1;
  d16[((((base + 1 * 4 + cb) * 2 + h) * 32 + rr) * 8) + j] = x2;
}

// Ring slot (one 32-row chunk): Rs = two k-steps x two planes x four column blocks x 64 lanes x 16 B of r fragments;
// Qs = the q^T tile [128 b][32 n] with the eight 16-byte pieces of a row XOR-swizzled by (b >> 1) & 7 (LDS-DMA writes
// 1 KB per wave instruction linearly, so there is no room for a padded pitch: the swizzle is applied on the GLOBAL
// address each lane fetches, and makes the 16-byte fragment reads of 32 consecutive rows conflict-free); Ps = the
// two staged p rows.
#define WGP_RS_B 16384
#define WGP_QS_B 16384
#define WGP_PS_B 256
#define WGP_BUF_B (WGP_RS_B + WGP_QS_B + WGP_PS_B)
#define WGP_SLOTS 4
#define WGP_SB() __builtin_amdgcn_sched_barrier(0)

// one pair of products -> one 32-bit word of each fragment plane (6 VALU)
#define WGP_SPLIT(k_, pa_, pb_, qa_, qb_)                                    \
  {                                                                          \
    unsigned w1_, w2_;                                                       \
    split2_pair_f16((pa_) * (qa_), (pb_) * (qb_), w1_, w2_);                 \
    asm volatile("" : "+v"(w1_), "+v"(w2_)); /* packed words NOW: the conversions must not sink into the next step */ \
    nh[k_] = w1_; nl[k_] = w2_;                                              \
  }

__global__ __launch_bounds__(512, 2) void bilinear_wgrad128_f16p_kernel(const float* __restrict__ pT_,
                                                                        const float* __restrict__ qT_,
                                                                        const uint4* __restrict__ Rq_,
                                                                        const float* __restrict__ mx_,
                                                                        WgradBatchDesc u) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[WGP_SLOTS * WGP_BUF_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hi = lane >> 5;
  const int grp = wave >> 2, wb = wave & 3;
  const int total = u.n_layers * u.splits * u.npairs, streams = u.n_layers * u.splits;
  const bool xcd_map = total % 8 == 0 && streams <= 8 && 8 % streams == 0 && u.npairs % (8 / streams) == 0 &&
                       gridDim.x % 8 == 0;
  const int rows_pad = u.rows_pad;
  // ---- per-lane LDS read offsets inside a slot ----
  const unsigned rd_rs = lane * 16;
  const int rowb = wb * 32 + r;
  const int fsw = (rowb >> 1) & 7;
  unsigned rd_q[2][2];   // [k-step][first / second 16-byte piece of the lane's 8 values]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int e = 0; e < 2; ++e) rd_q[ks][e] = WGP_RS_B + rowb * 128 + (((ks * 4 + 2 * hi + e) ^ fsw) << 4);
  const unsigned rd_ps = WGP_RS_B + WGP_QS_B + (grp * 32 + 8 * hi) * 4;
  // ---- LDS-DMA: scalar LDS bases of this wave's pieces, per-lane global byte offsets ----
  const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const unsigned dma_w = __builtin_amdgcn_readfirstlane(sbase + wave * 1024);
  const unsigned voff_r = (unsigned)tid * 16;
  const unsigned voff_q = (unsigned)(((tid >> 3) * (long)rows_pad + 4 * ((tid & 7) ^ ((tid >> 4) & 7))) * 4);
  const unsigned voff_q2 = voff_q + (unsigned)((long)64 * rows_pad * 4);

  for (int v = blockIdx.x; v < total; v += gridDim.x) {
    // XCD-aware placement: workgroups are dealt to the 8 XCDs round-robin by linear id, and every workgroup of a
    // (layer, split) stream reads the same q^T / r tiles.  Mapped so that the workgroups sharing an XCD share ONE stream
    // and run in near lockstep, the tiles enter that L2 once instead of once per workgroup (speed only).
    int stream, pair;
    if (xcd_map) {
      const int xcd = v & 7, w = v >> 3, xps = 8 / streams;
      stream = xcd / xps;
      pair = (xcd % xps) * (total / 8) + w;
    } else {
      stream = v / u.npairs;
      pair = v % u.npairs;
    }
    stream = __builtin_amdgcn_readfirstlane(stream);   // uniform: keep the unit's addressing on the scalar unit
    pair = __builtin_amdgcn_readfirstlane(pair);
    const int layer = stream / u.splits, z = stream % u.splits;
    const int a0 = pair * 2;
    const int nbeg = z * u.rows_per_split;
    const int nend = min(rows_pad, nbeg + u.rows_per_split);
    const int nchunks = (nend - nbeg) / 32;   // rows_per_split and rows_pad are multiples of 32
    const char* pT = reinterpret_cast<const char*>(pT_ + (long)layer * u.sT + (long)a0 * rows_pad);
    const char* qT = reinterpret_cast<const char*>(qT_ + (long)layer * u.sT);
    const char* Rq = reinterpret_cast<const char*>(Rq_ + (long)layer * u.sR);
    const float* mx = mx_ + 4 * layer;
    const unsigned voff_p = (unsigned)(((lane >> 5) * (long)rows_pad + (lane & 31)) * 4);

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
      // chunk ci -> ring slot ci % 4; five LDS-DMA instructions per wave (the index is clamped: the last iterations
      // re-load the last chunk into a slot nobody reads, which keeps the vmcnt arithmetic uniform)
#define WGP_DMA(ci_)                                                                        \
  {                                                                                         \
    const int cc_ = (ci_) < nchunks ? (ci_) : nchunks - 1;                                  \
    const long n0_ = nbeg + (long)cc_ * 32;                                                 \
    const unsigned d_ = dma_w + (unsigned)((ci_) & 3) * WGP_BUF_B;                          \
    const char* rb_ = Rq + (n0_ >> 4) * 8192;                                               \
    glds_b128(rb_, voff_r, d_);                                                             \
    glds_b128(rb_ + 8192, voff_r, d_ + 8192);                                               \
    glds_b128(qT + n0_ * 4, voff_q, d_ + WGP_RS_B);                                         \
    glds_b128(qT + n0_ * 4, voff_q2, d_ + WGP_RS_B + 8192);                                 \
    glds_b32(pT + n0_ * 4, voff_p, sbase + (unsigned)((ci_) & 3) * WGP_BUF_B + WGP_RS_B + WGP_QS_B); \
  }
      WGP_DMA(0);
      WGP_DMA(1);
      WGP_DMA(2);
      wait_vmcnt<5>();                  // chunks 0 and 1 have landed (this wave's pieces) ...
      __builtin_amdgcn_s_barrier();     // ... and everybody else's
      asm volatile("" ::: "memory");
      // fragments of the first step
      unsigned nh[4], nl[4];
      bf16x8 B[2][2];
      {
        const float4 qa = *reinterpret_cast<const float4*>(smem + rd_q[0][0]);
        const float4 qb = *reinterpret_cast<const float4*>(smem + rd_q[0][1]);
        const float4 pa = *reinterpret_cast<const float4*>(smem + rd_ps);
        const float4 pb = *reinterpret_cast<const float4*>(smem + rd_ps + 16);
        WGP_SPLIT(0, pa.x, pa.y, qa.x, qa.y) WGP_SPLIT(1, pa.z, pa.w, qa.z, qa.w)
        WGP_SPLIT(2, pb.x, pb.y, qb.x, qb.y) WGP_SPLIT(3, pb.z, pb.w, qb.z, qb.w)
        B[0][0] = *reinterpret_cast<const bf16x8*>(smem + rd_rs);
        B[0][1] = *reinterpret_cast<const bf16x8*>(smem + rd_rs + 4096);
      }
      // One 16-row step: 12 MFMAs on the fragments (a1, a2) made during the previous step; meanwhile the p, q values
      // of the NEXT step (slot offset so_, k-step kn_) are read and split into (nh, nl), and the B fragments are
      // fetched one column block ahead (the last prefetch reads the next step's first block at bn_).
#define WGP_STEP(bc_, bn_, so_, kn_)                                                                               \
  {                                                                                                                \
    const bf16x8 a1 = __builtin_bit_cast(bf16x8, make_uint4(nh[0], nh[1], nh[2], nh[3]));                          \
    const bf16x8 a2 = __builtin_bit_cast(bf16x8, make_uint4(nl[0], nl[1], nl[2], nl[3]));                          \
    WGP_SB();                                                                                                      \
    /* ---- column block 0: issue the reads of the next step's p, q ---- */                                        \
    B[1][0] = *reinterpret_cast<const bf16x8*>(smem + (bc_) + 1024);                                               \
    B[1][1] = *reinterpret_cast<const bf16x8*>(smem + (bc_) + 4096 + 1024);                                        \
    const float4 qa = *reinterpret_cast<const float4*>(smem + (so_) + rd_q[kn_][0]);                               \
    const float4 pa = *reinterpret_cast<const float4*>(smem + (so_) + rd_ps + (kn_) * 64);                         \
    WGP_SB();                                                                                                      \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a2), __builtin_bit_cast(f16x8, B[0][0]), acc[0], 0, 0, 0); \
    WGP_SB();                                                                                                      \
    const float4 qb = *reinterpret_cast<const float4*>(smem + (so_) + rd_q[kn_][1]);                               \
    const float4 pb = *reinterpret_cast<const float4*>(smem + (so_) + rd_ps + (kn_) * 64 + 16);                    \
    WGP_SB();                                                                                                      \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, B[0][1]), acc[0], 0, 0, 0); \
    WGP_SB();                                                                                                      \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, B[0][0]), acc[0], 0, 0, 0); \
    WGP_SB();                                                                                                      \
    /* ---- column block 1: split pairs 0, 1 ---- */                                                               \
    B[0][0] = *reinterpret_cast<const bf16x8*>(smem + (bc_) + 2048);                                               \
    B[0][1] = *reinterpret_cast<const bf16x8*>(smem + (bc_) + 4096 + 2048);                                        \
    WGP_SB();                                                                                                      \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a2), __builtin_bit_cast(f16x8, B[1][0]), acc[1], 0, 0, 0); \
    WGP_SB();                                                                                                      \
    WGP_SPLIT(0, pa.x, pa.y, qa.x, qa.y)                                                                           \
    WGP_SB();                                                                                                      \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, B[1][1]), acc[1], 0, 0, 0); \
    WGP_SB();                                                                                                      \
    WGP_SPLIT(1, pa.z, pa.w, qa.z, qa.w)                                                                           \
    WGP_SB();                                                                                                      \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, B[1][0]), acc[1], 0, 0, 0); \
    WGP_SB();                                                                                                      \
    /* ---- column block 2: split pairs 2, 3 ---- */                                                               \
    B[1][0] = *reinterpret_cast<const bf16x8*>(smem + (bc_) + 3072);                                               \
    B[1][1] = *reinterpret_cast<const bf16x8*>(smem + (bc_) + 4096 + 3072);                                        \
    WGP_SB();                                                                                                      \
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a2), __builtin_bit_cast(f16x8, B[0][0]), acc[2], 0, 0, 0); \
    WGP_SB();                                                                                                      \
    WGP_SPLIT(2, pb.x, pb.y, qb.x, qb.y)                                                                           \
    WGP_SB();                                                                                                      \
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, B[0][1]), acc[2], 0, 0, 0); \
    WGP_SB();                                                                                                      \
    WGP_SPLIT(3, pb.z, pb.w, qb.z, qb.w)                                                                           \
    WGP_SB();                                                                                                      \
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, B[0][0]), acc[2], 0, 0, 0); \
    WGP_SB();                                                                                                      \
    /* ---- column block 3: first B block of the next step ---- */                                                 \
    B[0][0] = *reinterpret_cast<const bf16x8*>(smem + (bn_));                                                      \
    B[0][1] = *reinterpret_cast<const bf16x8*>(smem + (bn_) + 4096);                                               \
    WGP_SB();                                                                                                      \
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a2), __builtin_bit_cast(f16x8, B[1][0]), acc[3], 0, 0, 0); \
    WGP_SB();                                                                                                      \
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, B[1][1]), acc[3], 0, 0, 0); \
    WGP_SB();                                                                                                      \
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, B[1][0]), acc[3], 0, 0, 0); \
    WGP_SB();                                                                                                      \
  }
#pragma clang loop unroll(disable)
      for (int c = 0; c < nchunks; ++c) {
        // chunk c + 3 into the slot chunk c - 1 was computed from (its readers passed the barrier that ended c - 1)
        WGP_DMA(c + 3);
        if ((c & 15) == 0 && c > 0) {   // two-level summation over the long row dimension (512-row partials);
          // groups alternate in sign (cancels the MFMA accumulator's rounding bias, see bilinear_rows128_ring16_kernel)
          const float sg = (((c >> 4) - 1) & 1) ? -1.f : 1.f;
#pragma unroll
          for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
              tot[cb][t] = fmaf(acc[cb][t], sg, tot[cb][t]);
              acc[cb][t] = 0.f;
            }
        }
        const unsigned o0 = (unsigned)(c & 3) * WGP_BUF_B, o1 = (unsigned)((c + 1) & 3) * WGP_BUF_B;
        // step 0 of chunk c: next = step 1 of the same slot
        WGP_STEP(o0 + rd_rs, o0 + rd_rs + 8192, o0, 1)
        // step 1: next = step 0 of chunk c + 1 (landed and published by the barrier that ended chunk c - 1)
        WGP_STEP(o0 + rd_rs + 8192, o1 + rd_rs, o1, 0)
        // chunk c + 2 (issued one iteration ago) must have landed before the barrier publishes it; younger than it:
        // only this iteration's five loads
        wait_vmcnt<5>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped re-loads still in flight target this unit's ring
#undef WGP_DMA
#undef WGP_STEP
    }
    const int a = a0 + grp;
    if (a < u.NA) {
      float* o = u.splits == 1 ? u.out[layer] + (long)a * 128 * 128
                               : u.slab + (((long)layer * u.splits + z) * u.NA + a) * 128 * 128;
      const float sg_last = (nchunks > 0 && (((nchunks - 1) >> 4) & 1)) ? -1.f : 1.f;
      // the lane id is laundered so that the 64 per-lane store addresses are computed HERE, once per unit, instead of
      // being hoisted out of the unit loop and kept alive (= spilled) across the main loop
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
__global__ void slab_sum_batch_kernel(const float* __restrict__ slab, int splits, long n, WgradBatchDesc u) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* sl = slab + (long)blockIdx.y * splits * n;
  float s = 0.f;
  for (int z = 0; z < splits; ++z) s += sl[(long)z * n + i];
  u.out[blockIdx.y][i] = s;
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

static int wgrad_bf16_splits(int NA) { return cdiv(256, cdiv(NA, 2)); }   // one 512-thread workgroup per CU
static size_t wgrad_bf16_ws(int nrows, int NA, size_t* o_pT, size_t* o_qT, size_t* o_Rq, size_t* o_slab) {
  const size_t np = (size_t)cdiv(nrows, 32) * 32;
  size_t off = 0;
  *o_pT = off; off += ws_round(np * 128, 4);
  *o_qT = off; off += ws_round(np * 128, 4);
  *o_Rq = off; off += ws_round(np * 128 * 3, 2);
  *o_slab = off; off += ws_round((size_t)wgrad_bf16_splits(NA) * NA * 128 * 128, 4);
  off += 16;                              // f16x3: {max|p|, max|q|, max|r|} behind the slabs
  return off;
}

// ---- batched f16x3 launch (bilinear_wgrad128_f16p_kernel) ----
// row splits per layer: enough units to fill the chip once (more only adds slab traffic), at least 8 chunks per split
static int wgrad_batch_pick(int n_layers, int nrows, int NA, int* rps_out) {
  const int npairs = cdiv(NA, 2), np = cdiv(nrows, 32) * 32;
  int splits = 256 / (n_layers * npairs);
  if (splits > np / 256) splits = np / 256;
  if (splits < 1) splits = 1;
  const int rps = cdiv(np / 32, splits) * 32;
  if (rps_out) *rps_out = rps;
  return cdiv(np, rps);
}
static size_t wgrad_batch_ws(int n_layers, int nrows, int NA, int splits, size_t* o_pT, size_t* o_qT, size_t* o_Rq,
                             size_t* o_slab, size_t* o_mx) {
  const size_t np = (size_t)cdiv(nrows, 32) * 32;
  size_t off = 0;
  *o_pT = off; off += ws_round((size_t)n_layers * np * 128, 4);
  *o_qT = off; off += ws_round((size_t)n_layers * np * 128, 4);
  *o_Rq = off; off += ws_round((size_t)n_layers * np * 128 * 2, 2);
  *o_slab = off; if (splits > 1) off += ws_round((size_t)n_layers * splits * NA * 128 * 128, 4);
  *o_mx = off; off += 256;
  return off;
}
bool bilinear_wgrad_batch_fast(int n_layers, int NA, int NB, int NC, long ldq, long ldr) {
  return (bilinear_mode() == 2 || bilinear_mode() == 4) && n_layers >= 1 && n_layers <= WGB_MAX && NA >= 1 && NA <= 128 && NB == 128 && NC == 128 &&
         (ldr % 4) == 0 && !force_generic();
}
size_t bilinear_wgrad_batch_ws_bytes(int n_layers, int nrows, int NA, int NB, int NC) {
  const size_t single = bilinear_wgrad_ws_bytes(nrows, NA, NB, NC);
  if (NB != 128 || NC != 128 || NA > 128 || NA < 1 || n_layers > WGB_MAX || n_layers < 1 || nrows <= 0) return single;
  size_t a, b, c, d, e;
  size_t batch = wgrad_batch_ws(n_layers, nrows, NA, wgrad_batch_pick(n_layers, nrows, NA, nullptr), &a, &b, &c, &d, &e);
  const size_t batch_c = wgradc_ws_bytes(n_layers, nrows, NA);     // f16x3c form (wgradc.hip)
  if (batch_c > batch) batch = batch_c;
  return batch > single ? batch : single;
}
// out[l][a,b,c] = sum_n p[l][n,a] q[l][n,b] r[l][n,c] for l < n_layers in ONE launch (f16x3 mode; other modes: one
// launch per layer).  max_wgs: workgroups of the grid (0 = 256, one per CU; 128 = half of the chip for running beside
// an HBM-bound kernel on another stream -- every workgroup then walks two units)
// The operand preparation (maxima, scaled transposes, fp16 planes) of ONE layer -- slot `slot` of an `n_layers` batch --
// so that a caller whose layers become ready one after the other can issue each layer's share early, on any stream,
// and finish with bilinear_wgrad_batch_launch(..., prepared = true) on the same workspace.  CGAT_ERR_UNSUPPORTED when
// the batched f16x3 kernel would not take these operands (the caller then launches unprepared).
int bilinear_wgrad_batch_prep(int slot, int n_layers, const float* p, long ldp, const float* q, long ldq, const float* r,
                              long ldr, int nrows, int NA, int NB, int NC, void* ws, size_t ws_bytes,
                              hipStream_t stream) {
  if (slot < 0 || slot >= n_layers || ((((uintptr_t)q) | ((uintptr_t)r)) & 15) != 0 || nrows <= 0 || nrows > 8000000 ||
      !bilinear_wgrad_batch_fast(n_layers, NA, NB, NC, ldq, ldr))
    return CGAT_ERR_UNSUPPORTED;
  if (bilinear_mode() == 4)
    return wgradc_prep(slot, 1, n_layers, &p, ldp, &q, ldq, &r, ldr, nrows, NA, ws, ws_bytes, stream);
  const int np = cdiv(nrows, 32) * 32;
  int rps = 0;
  const int splits = wgrad_batch_pick(n_layers, nrows, NA, &rps);
  size_t o_pT, o_qT, o_Rq, o_slab, o_mx;
  const size_t need = wgrad_batch_ws(n_layers, nrows, NA, splits, &o_pT, &o_qT, &o_Rq, &o_slab, &o_mx);
  if (!ws || ws_bytes < need) {
    cgat_set_error("bilinear_wgrad_batch_prep: workspace too small (%zu < %zu)", ws_bytes, need);
    return CGAT_ERR_WORKSPACE;
  }
  const long sT = (long)np * 128, sR8 = (long)np * 128 * 2;   // per-layer strides: floats of pT / qT, halves of Rq
  float* pT = (float*)((char*)ws + o_pT) + (size_t)slot * sT;
  float* qT = (float*)((char*)ws + o_qT) + (size_t)slot * sT;
  _Float16* Rq = (_Float16*)((char*)ws + o_Rq) + (size_t)slot * sR8;
  float* mx = (float*)((char*)ws + o_mx) + 4 * slot;
  WgradPrepDesc pd;
  memset(&pd, 0, sizeof(pd));
  pd.p[0] = p; pd.q[0] = q; pd.r[0] = r;
  CGAT_TRY(fill_launch(mx, 0.f, 4, stream));
  hipLaunchKernelGGL(absmax_rows_batch_kernel, dim3(256, 3), dim3(256), 0, stream, pd, ldp, ldq, ldr, nrows, NA, mx);
  CGAT_LAUNCH_CHECK();
  hipLaunchKernelGGL(transpose_pad_batch_kernel, dim3(np / 32, 4, 2), dim3(256), 0, stream, pd, ldp, ldq, nrows, NA, np, rps,
                     pT, qT, sT, (const float*)mx);
  CGAT_LAUNCH_CHECK();
  hipLaunchKernelGGL(split_rows_f16_batch_kernel, dim3(cdiv((long)np * 128, 256), 1), dim3(256), 0, stream, pd, ldr, nrows, np,
                     Rq, sR8, (const float*)mx);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

int bilinear_wgrad_batch_launch(int n_layers, const float* const* p, long ldp, const float* const* q, long ldq,
                                const float* const* r, long ldr, float* const* out, int nrows, int NA, int NB, int NC,
                                void* ws, size_t ws_bytes, hipStream_t stream, int max_wgs, bool prepared) {
  if (n_layers <= 0) return CGAT_OK;
  bool aligned = true;
  for (int l = 0; l < n_layers && l < WGB_MAX; ++l)
    aligned = aligned && ((((uintptr_t)q[l]) | ((uintptr_t)r[l])) & 15) == 0;
  // (the q^T tile is fetched with 32-bit lane offsets: 128 rows of rows_pad floats must stay below 4 GB)
  if (!aligned || nrows <= 0 || nrows > 8000000 || !bilinear_wgrad_batch_fast(n_layers, NA, NB, NC, ldq, ldr)) {
    if (prepared) {
      cgat_set_error("bilinear_wgrad_batch: prepared operands but not the batched form");
      return CGAT_ERR_ARG;
    }
    for (int l = 0; l < n_layers; ++l)
      CGAT_TRY(bilinear_wgrad_launch(p[l], ldp, q[l], ldq, r[l], ldr, out[l], nrows, NA, NB, NC, ws, ws_bytes, stream,
                                     max_wgs > 0 && max_wgs < 256 ? max_wgs / cdiv(NA, 2) : 0));
    return CGAT_OK;
  }
  if (bilinear_mode() == 4)
    return wgradc_launch(n_layers, p, ldp, q, ldq, r, ldr, out, nrows, NA, ws, ws_bytes, stream, max_wgs, prepared);
  if (max_wgs <= 0 || max_wgs > 256) max_wgs = 256;
  const int npairs = cdiv(NA, 2);
  const int np = cdiv(nrows, 32) * 32;
  int rps = 0;
  const int splits = wgrad_batch_pick(n_layers, nrows, NA, &rps);
  size_t o_pT, o_qT, o_Rq, o_slab, o_mx;
  const size_t need = wgrad_batch_ws(n_layers, nrows, NA, splits, &o_pT, &o_qT, &o_Rq, &o_slab, &o_mx);
  if (!ws || ws_bytes < need) {
    cgat_set_error("bilinear_wgrad_batch: workspace too small (%zu < %zu)", ws_bytes, need);
    return CGAT_ERR_WORKSPACE;
  }
  float* pT = (float*)((char*)ws + o_pT);
  float* qT = (float*)((char*)ws + o_qT);
  _Float16* Rq = (_Float16*)((char*)ws + o_Rq);
  float* mx = (float*)((char*)ws + o_mx);
  WgradPrepDesc pd;
  WgradBatchDesc u;
  memset(&pd, 0, sizeof(pd));
  memset(&u, 0, sizeof(u));
  for (int l = 0; l < n_layers; ++l) { pd.p[l] = p[l]; pd.q[l] = q[l]; pd.r[l] = r[l]; u.out[l] = out[l]; }
  u.slab = (float*)((char*)ws + o_slab);
  u.sT = (long)np * 128;
  u.sR = (long)np * 128 * 2 * 2 / 16;
  u.n_layers = n_layers; u.splits = splits; u.npairs = npairs; u.NA = NA; u.rows_pad = np; u.rows_per_split = rps;
  if (!prepared) {
    CGAT_TRY(fill_launch(mx, 0.f, 64, stream));
    hipLaunchKernelGGL(absmax_rows_batch_kernel, dim3(256, 3 * n_layers), dim3(256), 0, stream, pd, ldp, ldq, ldr, nrows, NA,
                       mx);
    CGAT_LAUNCH_CHECK();
    hipLaunchKernelGGL(transpose_pad_batch_kernel, dim3(np / 32, 4, 2 * n_layers), dim3(256), 0, stream, pd, ldp, ldq, nrows,
                       NA, np, rps, pT, qT, u.sT, (const float*)mx);
    CGAT_LAUNCH_CHECK();
    hipLaunchKernelGGL(split_rows_f16_batch_kernel, dim3(cdiv((long)np * 128, 256), n_layers), dim3(256), 0, stream, pd, ldr,
                       nrows, np, Rq, u.sR * 8, (const float*)mx);
    CGAT_LAUNCH_CHECK();
  }
  const int units = n_layers * splits * npairs;
  int grid = units < max_wgs ? units : max_wgs;
  {
    CGAT_PROF("bilinear_wgrad", stream);
    hipLaunchKernelGGL(bilinear_wgrad128_f16p_kernel, dim3(grid), dim3(512), 0, stream, (const float*)pT, (const float*)qT,
                       (const uint4*)Rq, (const float*)mx, u);
  }
  CGAT_LAUNCH_CHECK();
  if (splits > 1) {
    const long n = (long)NA * 128 * 128;
    hipLaunchKernelGGL(slab_sum_batch_kernel, dim3(cdiv(n, 256), n_layers), dim3(256), 0, stream, (const float*)u.slab, splits,
                       n, u);
    CGAT_LAUNCH_CHECK();
  }
  return CGAT_OK;
}

size_t bilinear_wgrad_ws_bytes(int nrows, int NA, int NB, int NC) {
  if (!(NB == 128 && NC == 128) && nrows > 0 && NA > 0 && NB > 0 && NC > 0) {   // fp32 engine, split over the rows
    const int sp = gemm_pick_splits(NA, NB * NC, nrows);
    return sp > 1 ? ws_round((size_t)sp * NA * NB * NC, 4) : 0;
  }
  if (NB == 128 && NC == 128) {
    size_t a, b, c, d;
    size_t bf = wgrad_bf16_ws(nrows, NA, &a, &b, &c, &d);
    size_t f32 = ws_round((size_t)wgrad_splits(nrows, NA) * NA * NB * NC, 4);
    if (f32 > bf) bf = f32;
    if (NA >= 1 && NA <= 128 && nrows > 0) {   // the f16x3 / f16x3c forms: batched kernels with one layer
      size_t e;
      const size_t one = wgrad_batch_ws(1, nrows, NA, wgrad_batch_pick(1, nrows, NA, nullptr), &a, &b, &c, &d, &e);
      if (one > bf) bf = one;
      const size_t one_c = wgradc_ws_bytes(1, nrows, NA);
      if (one_c > bf) bf = one_c;
    }
    return bf;
  }
  return 0;
}

// force_splits > 0: number of row splits = workgroups per `a` pair (default: enough for one workgroup per CU; 2 gives
// 128 workgroups, i.e. half the chip, for running beside an HBM-bound kernel on another stream)
int bilinear_wgrad_launch(const float* p, long ldp, const float* q, long ldq, const float* r, long ldr, float* out,
                          int nrows, int NA, int NB, int NC, void* ws, size_t ws_bytes, hipStream_t stream,
                          int force_splits) {
  if (wgrad_fast(q, ldq, r, ldr, NB, NC) && (bilinear_mode() == 2 || bilinear_mode() == 4) && nrows > 0 &&
      nrows <= 8000000 && NA <= 128) {
    // f16x3 / f16x3c: the batched kernels with one layer (row splits fill the chip)
    return bilinear_wgrad_batch_launch(1, &p, ldp, &q, ldq, &r, ldr, &out, nrows, NA, NB, NC, ws, ws_bytes, stream,
                                       force_splits > 0 ? force_splits * cdiv(NA, 2) : 0);
  }
  if (wgrad_fast(q, ldq, r, ldr, NB, NC) && bilinear_mode() != 0 && nrows > 0) {
    size_t o_pT, o_qT, o_Rq, o_slab;
    const size_t need = wgrad_bf16_ws(nrows, NA, &o_pT, &o_qT, &o_Rq, &o_slab);
    if (!ws || ws_bytes < need) {
      cgat_set_error("bilinear_wgrad: workspace too small (%zu < %zu)", ws_bytes, need);
      return CGAT_ERR_WORKSPACE;
    }
    const int np = cdiv(nrows, 32) * 32;
    float* pT = (float*)((char*)ws + o_pT);
    float* qT = (float*)((char*)ws + o_qT);
    __bf16* Rq = (__bf16*)((char*)ws + o_Rq);
    float* slab = (float*)((char*)ws + o_slab);
    float* mx = (float*)((char*)ws + need - 16);
    const bool f16 = bilinear_mode() == 2;
    if (f16) CGAT_TRY(fill_launch(mx, 0.f, 4, stream));
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(np / 32, cdiv(NA, 32)), dim3(256), 0, stream, p, ldp, nrows, NA, np, pT,
                       f16 ? mx : (float*)nullptr);
    CGAT_LAUNCH_CHECK();
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(np / 32, 4), dim3(256), 0, stream, q, ldq, nrows, 128, np, qT,
                       f16 ? mx + 1 : (float*)nullptr);
    CGAT_LAUNCH_CHECK();
    if (f16) {
      hipLaunchKernelGGL(absmax_rows128_kernel, dim3(512), dim3(256), 0, stream, r, ldr, nrows, mx + 2);
      CGAT_LAUNCH_CHECK();
      hipLaunchKernelGGL(split_rows_bf16_kernel<true>, dim3(cdiv((long)np * 128, 256)), dim3(256), 0, stream, r, ldr, nrows, np, Rq, (const float*)mx);
    } else {
      hipLaunchKernelGGL(split_rows_bf16_kernel<false>, dim3(cdiv((long)np * 128, 256)), dim3(256), 0, stream, r, ldr, nrows, np, Rq, (const float*)mx);
    }
    CGAT_LAUNCH_CHECK();
    int splits = wgrad_bf16_splits(NA);
    if (force_splits > 0 && force_splits < splits) splits = force_splits;
    int rps = cdiv(np / 32, splits) * 32;
    splits = cdiv(np, rps);
    {
      CGAT_PROF("bilinear_wgrad", stream);
#ifdef CGAT_DEV_ABLATIONS   // timing-only variants (wrong results): only in builds made for tools/wgrad_ablation.py
      static int abl = -1;
      if (abl < 0) { const char* e = getenv("CGAT_WGRAD_ABL"); abl = e ? atoi(e) : 0; }
#else
      const int abl = 0;
#endif
#define WG_GO(A_) hipLaunchKernelGGL((bilinear_wgrad128_bf16_kernel<6, A_>), dim3(cdiv(NA, 2), splits), dim3(512), 0, stream, pT, qT, (const uint4*)Rq, slab, np, rps, NA, (const float*)mx)
      if (f16)
        hipLaunchKernelGGL(bilinear_wgrad128_bf16_kernel<2>, dim3(cdiv(NA, 2), splits), dim3(512), 0, stream, pT, qT,
                           (const uint4*)Rq, slab, np, rps, NA, (const float*)mx);
      else if (bilinear_mode() != 3 && abl) {
        switch (abl) { case 1: WG_GO(1); break; case 2: WG_GO(2); break; case 3: WG_GO(3); break; case 4: WG_GO(4); break; default: WG_GO(7); break; }
      } else if (bilinear_mode() != 3)
        hipLaunchKernelGGL(bilinear_wgrad128_bf16_kernel<6>, dim3(cdiv(NA, 2), splits), dim3(512), 0, stream, pT, qT,
                           (const uint4*)Rq, slab, np, rps, NA, (const float*)mx);
      else
        hipLaunchKernelGGL(bilinear_wgrad128_bf16_kernel<3>, dim3(cdiv(NA, 2), splits), dim3(512), 0, stream, pT, qT,
                           (const uint4*)Rq, slab, np, rps, NA, (const float*)mx);
    }
    CGAT_LAUNCH_CHECK();
    long n = (long)NA * NB * NC;
    hipLaunchKernelGGL(slab_sum_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, (const float*)slab, splits, n, out);
    CGAT_LAUNCH_CHECK();
    return CGAT_OK;
  }
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
  } else if (!force_generic()) {
    // widths other than 128: out [NA, NB * NC] = p^T (q (x) r) on the fp32 engine, rows split over workgroups when the
    // output has few tiles (33 ms -> 0.7 at 83 340 rows of width 64)
    GemmParams g = gemm_params(NA, NB * NC, nrows, p, ldp, r, ldr, out, (long)NB * NC);
    g.a_kmajor = 1; g.b_kmajor = 1;
    g.b_outer = q; g.ld_b_outer = ldq; g.outer_n = NC;
    g.splits = gemm_pick_splits(NA, NB * NC, nrows);
    if (g.splits > 1 && (!ws || ws_bytes < ws_round((size_t)g.splits * NA * NB * NC, 4))) g.splits = 1;
    CGAT_TRY(gemm_launch(g, ws, ws_bytes, stream));
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
  if (bilinear_T_interleaved(dims[perm1], dims[perm2]) && bilinear_mode() != 0) {
    long st[3] = {(long)n1 * n2, (long)n2, 1};   // source strides of dims 0, 1, 2
    if (bilinear_mode() == 2) return prepare_T_f16_launch(src, dst, dims[perm0], st[perm0], st[perm1], st[perm2], 1, stream);
    if (bilinear_mode() == 4) return prepare_T_f16c_launch(src, dst, dims[perm0], st[perm0], st[perm1], st[perm2], 1, stream);
    return prepare_T_bf16_launch(src, dst, dims[perm0], st[perm0], st[perm1], st[perm2], 1, stream);
  }
  return permute3_launch(src, dst, n0, n1, n2, perm0, perm1, perm2,
                         bilinear_T_interleaved(dims[perm1], dims[perm2]) ? 1 : 0, stream);
}
