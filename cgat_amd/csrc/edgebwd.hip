// Backward products of the per-edge first-layer weight W_e over the 1536-wide pre-activation gradient gZ
// (the `edge_attr` slice of MultiHeadNetwork.fc_in for both message networks, reference CGAT.py:96):
//     g_edge_attr[perm[t], :] = gZ[t, :] @ W_e               (edge_ge_kernel,  K = 1536, 128 outputs per edge)
// gZ is produced once per step by edge_seg_bwd_kernel in 128-column blocks [W2/128][E][128] and read here
// straight from HBM: every element is used for only 128 multiply-adds, so the split into three bf16 planes
// (bilinear.hip: six v_mfma_f32_16x16x32_bf16 passes, fp32 accumulation, fp32-equivalent) happens in the loop
// -- ~64 VALU instructions per 96 MFMAs and wave, overlapped by the second wave of the SIMD -- while W_e
// arrives pre-split in fragment order through a double-buffered LDS tile shared by the 8 waves.
#include "common.h"
#include "kernels.h"
#include "mfma_bf16.h"

// Wq: prepare_T_bf16 planes of the operand (a = 128-column block of gZ, b = column in block, c = output) =
// W_e[128 a + b][c]; chunk (a, half, s) = 12 KB at Wq + ((a*2 + half)*4 + s) * 768 uint4.
// RC: the operand is not read but rebuilt from its ingredients (struct EdgeRC, kernels.h): per k-step and row one mask
// word, one coefficient and 8 floats of a per-node row (or of the constant wA) instead of 8 floats of gZ
// LeakyReLU'(z) from the sign bit: 1.0f or 0.01f chosen by a bitwise select of the two bit patterns (v_bfe_i32 +
// v_bfi_b32: two instructions instead of the and / compare / cndmask a `?:` compiles to)
__device__ __forceinline__ float rc_d(unsigned m, int bit) {
  const unsigned s = (unsigned)((int)(m << (31 - bit)) >> 31);
  return __uint_as_float((s & 0x3F800000u) | (~s & 0x3C23D70Au));
}
template <int PASSES, bool RC = false>
__global__ __launch_bounds__(512, 2) void edge_ge_kernel(const float* __restrict__ gZ, long ldg, long gzb,
                                                         const uint4* __restrict__ Wq, int ncb,
                                                         float* __restrict__ out, long ldo,
                                                         const int* __restrict__ scatter, int E, int accumulate,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ amax, const EdgeRC rc, HeadBatch hb) {
  // grid.y = head of a multi-head second layer (edge_ge_heads_launch): per-head operand offsets
  gZ += (long)blockIdx.y * hb.in;
  Wq += (long)blockIdx.y * hb.w;
  out += (long)blockIdx.y * hb.out;
  if (bias) bias += (long)blockIdx.y * hb.bias;
  // PASSES == 2: two fp16 planes, three passes (mfma_bf16.h).  The rows of gZ are consumed k-step by k-step, so their
  // scale is per tensor: amax[0] = max |gZ| from the kernel that produced it; the weight's max sits behind its planes.
  // PASSES == 1 (edge storage "bf16-mma", BASELINE configs[4]'s "bf16 activations with MFMA edge-MLP"): ONE bf16 pass on
  // the leading planes of both operands -- bf16 operands, fp32 accumulation; the three-plane weight image is read as
  // it is (its first plane is the round-to-nearest bf16 of the weight), only that plane is staged
  constexpr bool F16 = PASSES == 2;
  constexpr bool ONE = PASSES == 1;
  constexpr int NP = F16 ? 2 : 3;
  constexpr int HP = NP * 256;                   // 16-byte pieces of one (a, half, k-step) block
  __shared__ uint4 Bs[2][2 * HP];                // [buffer][half][plane][cb][lane]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kg = lane >> 4;
  const int row_w = blockIdx.x * 256 + wave * 32;
  const int row_a = row_w + n16, row_b = row_a + 16;
  const long rca = row_a < E ? row_a : E - 1, rcb = row_b < E ? row_b : E - 1;
  // element (t, 128 a + j) of the operand lives at gZ[t * ldg + a * gzb + j]: (ldg, gzb) = (128, E * 128) for the
  // column-blocked gZ, (W2, 128) for a plain row-major matrix
  const float* ga = gZ + rca * ldg + 8 * kg;     // + a * gzb + 32 s
  const float* gb = gZ + rcb * ldg + 8 * kg;
  const int nk = ncb * 4;
  const int ks0 = RC ? (int)((long)blockIdx.y * hb.ks0) : 0;
  // RC: per-row bases of the ingredients
  const unsigned* mka = nullptr; const unsigned* mkb = nullptr;
  const float *gsa = nullptr, *gsb = nullptr, *caA = nullptr, *cbA = nullptr, *caM = nullptr, *cbM = nullptr;
  if constexpr (RC) {
    mka = rc.mask + rca * rc.nw; mkb = rc.mask + rcb * rc.nw;
    gsa = rc.gS + (long)rc.dst[rca] * rc.HHd + 8 * kg; gsb = rc.gS + (long)rc.dst[rcb] * rc.HHd + 8 * kg;
    caA = rc.ga + rca * rc.H; cbA = rc.ga + rcb * rc.H;
    caM = rc.alpha + rca * rc.H; cbM = rc.alpha + rcb * rc.H;
  }

  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float sA = 1.f, inv_all = 1.f;
  if constexpr (F16) {
    float iA, sW, iW;
    pow2_scale(amax[0], sA, iA);
    pow2_scale(reinterpret_cast<const float*>(Wq + (long)ncb * 8 * HP)[0], sW, iW);
    inv_all = iA * iW;
  }

  // B staging: thread tid moves 16-byte pieces tid, tid + 512 (, tid + 1024) of the k-step's [half0 | half1] image
  // (sb*: the tile stored at the end of this iteration, loaded during the previous one; tb*: the tile after it, loaded
  // during this iteration -- a tile loaded and stored within one iteration left every wave waiting for its L2 latency
  // in front of the barrier)
  uint4 sb0, sb1, sb2, tb0, tb1, tb2;
  const int p1 = tid + 512, p2 = tid + 1024;
  const int p0 = tid < 256 ? tid : HP + tid - 256;      // ONE: where the thread's plane-0 piece lives in the tile
#define GE_BLOAD(ks_, B0_, B1_, B2_)                                                                     \
  {                                                                                                      \
    const long a_ = (ks_) >> 2, s_ = (ks_) & 3;                                                          \
    const uint4* h0 = Wq + ((a_ * 2 + 0) * 4 + s_) * HP;                                                 \
    const uint4* h1 = Wq + ((a_ * 2 + 1) * 4 + s_) * HP;                                                 \
    if (ONE) {                       /* plane 0 of each half: 256 + 256 pieces, one per thread */          \
      B0_ = tid < 256 ? h0[tid] : h1[tid - 256];                                                         \
    } else {                                                                                             \
      B0_ = h0[tid];                                                                                     \
      B1_ = p1 < HP ? h0[p1] : h1[p1 - HP];                                                              \
      if (NP == 3) B2_ = h1[p2 - HP];                                                                    \
    }                                                                                                    \
  }
#define GE_BSTORE(buf_)                                                                                  \
  {                                                                                                      \
    if (ONE) {                                                                                           \
      Bs[buf_][p0] = sb0;                                                                                \
    } else {                                                                                             \
      Bs[buf_][tid] = sb0; Bs[buf_][p1] = sb1;                                                           \
      if (NP == 3) Bs[buf_][p2] = sb2;                                                                   \
    }                                                                                                    \
  }
  // raw gZ (rows a/b, 8 columns each) of the next k-step (ra*, rb*) and of the one after (sa*, sb*): loads are
  // issued two k-steps (~3 us) before their values are split, enough bytes in flight per CU to cover HBM latency
  float4 ra0, ra1, rb0, rb1, sa0, sa1, sb0_, sb1_;
  // RC: coefficient and mask word of rows a / b for the k-step in r* (rc*) and in s* (sc*)
  float rca_c = 0.f, rcb_c = 0.f, sca_c = 0.f, scb_c = 0.f;
  unsigned rca_m = 0, rcb_m = 0, sca_m = 0, scb_m = 0;
#define GE_ALOAD(ks_, A0_, A1_, B0_, B1_, CA_, CB_, MA_, MB_)                                            \
  {                                                                                                      \
    if constexpr (RC) {                                                                                  \
      const int kk_ = (ks_) + ks0;                     /* k-step within the whole row (K groups, grid.y) */  \
      const int c0_ = 32 * kk_;                        /* first column of the k-step (uniform) */           \
      const bool isA_ = c0_ < rc.HHd;                                                                    \
      const int cc_ = isA_ ? c0_ : c0_ - rc.HHd, h_ = cc_ / rc.Hd;                                       \
      const float4* pa = reinterpret_cast<const float4*>(isA_ ? rc.wA + cc_ + 8 * kg : gsa + cc_);      \
      const float4* pb = reinterpret_cast<const float4*>(isA_ ? rc.wA + cc_ + 8 * kg : gsb + cc_);      \
      A0_ = pa[0]; A1_ = pa[1]; B0_ = pb[0]; B1_ = pb[1];                                                \
      CA_ = (isA_ ? caA : caM)[h_]; CB_ = (isA_ ? cbA : cbM)[h_];                                        \
      MA_ = mka[kk_]; MB_ = mkb[kk_];                                                                    \
    } else {                                                                                             \
      const long off = (long)((ks_) >> 2) * gzb + 32 * ((ks_) & 3);                                      \
      const float4* pa = reinterpret_cast<const float4*>(ga + off);                                      \
      const float4* pb = reinterpret_cast<const float4*>(gb + off);                                      \
      A0_ = pa[0]; A1_ = pa[1]; B0_ = pb[0]; B1_ = pb[1];                                                \
    }                                                                                                    \
  }
  // RC: the stored value was (coefficient * vector) * d, rebuilt in the same order
#define GE_SPLIT(R0_, R1_, C_, M_, Q1_, Q2_, Q3_)                                                        \
  {                                                                                                      \
    float g_[8] = {R0_.x, R0_.y, R0_.z, R0_.w, R1_.x, R1_.y, R1_.z, R1_.w};                              \
    if constexpr (RC) {           /* the power-of-two scale of the fp16 form rides on the coefficient: exact */ \
      const unsigned mb_ = (M_) >> (8 * kg);                                                             \
      const float cs_ = F16 ? (C_) * sA : (C_);                                                          \
      /* __fmul_rn: the stored value was ROUNDED; contracted into the split's v - hi it would not be */   \
      _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) g_[i_] = __fmul_rn(cs_ * g_[i_], rc_d(mb_, i_));  \
    }                                                                                                    \
    if constexpr (F16 && !RC) {                                                                          \
      const float v[8] = {g_[0] * sA, g_[1] * sA, g_[2] * sA, g_[3] * sA,                                \
                          g_[4] * sA, g_[5] * sA, g_[6] * sA, g_[7] * sA};                               \
      split2_x8_f16(v, Q1_, Q2_);                                                                        \
    } else if constexpr (F16) {                                                                          \
      split2_x8_f16(g_, Q1_, Q2_);                                                                       \
    } else {                                                                                             \
      split3_x8(g_, Q1_, Q2_, Q3_);                                                                      \
    }                                                                                                    \
  }
  bf16x8 qa1, qa2, qa3, qb1, qb2, qb3;           // current k-step's gZ fragments (rows a, b)
  bf16x8 na1, na2, na3, nb1, nb2, nb3;           // next k-step's
  GE_ALOAD(0, ra0, ra1, rb0, rb1, rca_c, rcb_c, rca_m, rcb_m);
  GE_BLOAD(0, sb0, sb1, sb2);
  GE_SPLIT(ra0, ra1, rca_c, rca_m, qa1, qa2, qa3);
  GE_SPLIT(rb0, rb1, rcb_c, rcb_m, qb1, qb2, qb3);
  GE_BSTORE(0);
  ra0 = ra1 = rb0 = rb1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (PASSES < 6) GE_BLOAD(1, sb0, sb1, sb2);   // nk >= 4
  GE_ALOAD(1, ra0, ra1, rb0, rb1, rca_c, rcb_c, rca_m, rcb_m);
  __syncthreads();

#define GE_MFMA1(F1_, F2_, F3_, Q1_, Q2_, Q3_, P_)                                                       \
  {                                                                                                      \
    if (PASSES >= 6) {                                                                                   \
      P_ = mma16<F16>(F3_, Q1_, P_);                                                                     \
      P_ = mma16<F16>(F1_, Q3_, P_);                                                                     \
      P_ = mma16<F16>(F2_, Q2_, P_);                                                                     \
    }                                                                                                    \
    if (!ONE) {                                                                                          \
      P_ = mma16<F16>(F2_, Q1_, P_);                                                                     \
      P_ = mma16<F16>(F1_, Q2_, P_);                                                                     \
    }                                                                                                    \
    P_ = mma16<F16>(F1_, Q1_, P_);                                                                       \
  }
  // One iteration: loads for k-step ks + 2 go into the (S, T) register sets, the (R, SB) sets -- loaded one iteration
  // ago -- are consumed.  The loop is unrolled by two with the sets swapped instead of copied: a register copy of an
  // in-flight load is a wait for it, and such copies at the loop end drained vmcnt every k-step.
#define GE_ITER(ks_, buf_, RA0, RA1, RB0, RB1, RCA, RCB, RMA, RMB, SA0, SA1, SB0, SB1, SCA, SCB, SMA, SMB,           \
                UB0, UB1, UB2, TB0, TB1, TB2)                                                            \
  {                                                                                                      \
    const bf16x8* bs = reinterpret_cast<const bf16x8*>(&Bs[buf_][lane]);                                 \
    /* no branches in the body (loads past the end re-fetch the last k-step, splits and stores of such tiles are */ \
    /* harmless): with conditional loads the compiler's wait-count bookkeeping turns conservative at every merge */ \
    const int kl_ = (ks_) + 2 < nk ? (ks_) + 2 : nk - 1;                                                 \
    if constexpr (PASSES >= 6) {   /* six-pass form: its three-plane tile one k-step ahead only (12 VGPRs less; */ \
      const int kb_ = (ks_) + 1 < nk ? (ks_) + 1 : nk - 1;   /* the loads sit in front of the operand loads, so the */ \
      GE_BLOAD(kb_, UB0, UB1, UB2);                          /* store's wait leaves those in flight) */     \
    } else {                                                                                             \
      GE_BLOAD(kl_, TB0, TB1, TB2);                                                                      \
    }                                                                                                    \
    GE_ALOAD(kl_, SA0, SA1, SB0, SB1, SCA, SCB, SMA, SMB);                                               \
    __builtin_amdgcn_sched_barrier(0);     /* the loads are ISSUED here, not where the scheduler likes them */ \
    /* the raw values in the R set belong to k-step ks + 1: split them while this step's MFMAs run */      \
    bf16x8 f1 = bs[0], f2, f3;                                                                           \
    if (!ONE) f2 = bs[256];                                                                              \
    if (PASSES >= 6) f3 = bs[512];                                                                       \
    _Pragma("unroll") for (int g = 0; g < 8; ++g) {      /* 16-column output block g = (half, cb) */        \
      bf16x8 n1, n2, n3;                                                                                 \
      if (g < 7) {                                                                                       \
        const int o = ((g + 1) >> 2) * HP + ((g + 1) & 3) * 64;                                          \
        n1 = bs[o];                                                                                      \
        if (!ONE) n2 = bs[o + 256];                                                                      \
        if (PASSES >= 6) n3 = bs[o + 512];                                                               \
      }                                                                                                  \
      GE_MFMA1(f1, f2, f3, qa1, qa2, qa3, acc[2 * g + 0]);                                               \
      if (g == 1) GE_SPLIT(RA0, RA1, RCA, RMA, na1, na2, na3);                                           \
      GE_MFMA1(f1, f2, f3, qb1, qb2, qb3, acc[2 * g + 1]);                                               \
      if (g == 4) GE_SPLIT(RB0, RB1, RCB, RMB, nb1, nb2, nb3);                                           \
      if (g < 7) { f1 = n1; f2 = n2; f3 = n3; }                                                          \
    }                                                                                                    \
    if (ONE) { Bs[(buf_) ^ 1][p0] = UB0; }                                                               \
    else { Bs[(buf_) ^ 1][tid] = UB0; Bs[(buf_) ^ 1][p1] = UB1; if (NP == 3) Bs[(buf_) ^ 1][p2] = UB2; } \
    /* LDS writes of this wave done, then the barrier -- NOT __syncthreads(): its fence also drains vmcnt, i.e. the */ \
    /* operand loads just issued two k-steps ahead */                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
    __builtin_amdgcn_s_barrier();                                                                        \
    asm volatile("" ::: "memory");                                                                       \
    qa1 = na1; qa2 = na2; qa3 = na3; qb1 = nb1; qb2 = nb2; qb3 = nb3;                                    \
  }
  // One 128-column block of the operand (four k-steps) per trip.  The matrix instruction's accumulator rounds with a
  // sign-independent bias (-0.9e-9 ... -1.5e-9 of sum |g||w| on every output when all k-steps accumulated
  // with one sign, tools/f16_bias_probe.py): the weight planes of odd blocks are prepared NEGATED (edge_ge_launch) and
  // the accumulators change sign between blocks (exact), so a block's bias enters the result with the sign (-1)^a and
  // consecutive blocks cancel; after the last block the accumulators hold (-1)^ncb times the sum.
  for (int a = 0; a < ncb; ++a) {
    const int ks = 4 * a;
    GE_ITER(ks, 0, ra0, ra1, rb0, rb1, rca_c, rcb_c, rca_m, rcb_m, sa0, sa1, sb0_, sb1_, sca_c, scb_c, sca_m, scb_m,
            sb0, sb1, sb2, tb0, tb1, tb2)
    GE_ITER(ks + 1, 1, sa0, sa1, sb0_, sb1_, sca_c, scb_c, sca_m, scb_m, ra0, ra1, rb0, rb1, rca_c, rcb_c, rca_m, rcb_m,
            tb0, tb1, tb2, sb0, sb1, sb2)
    GE_ITER(ks + 2, 0, ra0, ra1, rb0, rb1, rca_c, rcb_c, rca_m, rcb_m, sa0, sa1, sb0_, sb1_, sca_c, scb_c, sca_m, scb_m,
            sb0, sb1, sb2, tb0, tb1, tb2)
    GE_ITER(ks + 3, 1, sa0, sa1, sb0_, sb1_, sca_c, scb_c, sca_m, scb_m, ra0, ra1, rb0, rb1, rca_c, rcb_c, rca_m, rcb_m,
            tb0, tb1, tb2, sb0, sb1, sb2)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = -acc[i];
  }
#undef GE_ITER
#undef GE_BLOAD
#undef GE_BSTORE
#undef GE_ALOAD
#undef GE_SPLIT
#undef GE_MFMA1
  // acc[2 g + nb][j] = out[row(nb)][16 g + 4 kg + j]
  const long oa = scatter ? (long)scatter[rca] : rca, ob = scatter ? (long)scatter[rcb] : rcb;
  {
    const float fin = (F16 ? inv_all : 1.f) * ((ncb & 1) ? -1.f : 1.f);   // undo the scales (fp16 form) and the last sign
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = acc[i] * fin;
  }
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = *reinterpret_cast<const float4*>(bias + 16 * g + 4 * kg);
    if (row_a < E) {
      const f32x4 v = acc[2 * g + 0];
      float4* o = reinterpret_cast<float4*>(out + oa * ldo + 16 * g + 4 * kg);
      float4 w = make_float4(v[0] + bv.x, v[1] + bv.y, v[2] + bv.z, v[3] + bv.w);
      if (accumulate) { const float4 u = *o; w.x += u.x; w.y += u.y; w.z += u.z; w.w += u.w; }
      *o = w;
    }
    if (row_b < E) {
      const f32x4 v = acc[2 * g + 1];
      float4* o = reinterpret_cast<float4*>(out + ob * ldo + 16 * g + 4 * kg);
      float4 w = make_float4(v[0] + bv.x, v[1] + bv.y, v[2] + bv.z, v[3] + bv.w);
      if (accumulate) { const float4 u = *o; w.x += u.x; w.y += u.y; w.z += u.z; w.w += u.w; }
      *o = w;
    }
  }
}

// ---------------------------------------------------------------------------------------
//     g_W_e[col, :] = sum_t gZ[t, col] * e[perm[t], :]           (edge_gw_kernel,  K = E, 1536 x 128 outputs)
// Both operands of this product are needed with t (the edge slot) as the MFMA's k index, i.e. transposed with
// respect to their [t][feature] storage:
//   * e[perm[t]] is pre-split ONCE per step into fragment-ordered planes (prepare_T_bf16 with a row gather:
//     operand (a = t/128, b = t%128, c = k) -- 0.77 GB at E = 1M, zero-padded past E) and streamed through a
//     double-buffered LDS tile shared by all waves of the workgroup;
//   * gZ tiles (32 slots x 128 columns = 16 KB contiguous in the blocked layout) are loaded coalesced, split in
//     registers, written as bf16 planes into an XOR-swizzled [32][128] LDS image and read back transposed with
//     ds_read_b64_tr_b16 (image (b) of cdna_hip_programming.md T10: conflict-free writes and transposed reads).
// A workgroup owns two 128-column blocks (8 waves x 32 columns, all 128 outputs k per wave: 64 accumulator
// VGPRs) and a contiguous range of k-steps; the workgroups of one range (one per column-block pair) are adjacent
// in the grid so that the e planes they share are fetched from HBM once and hit the Infinity Cache afterwards.
// Two-level summation through memory: every FLUSH k-steps (2048 slots) the accumulators are added into the
// workgroup's private slab tile and cleared; consecutive groups accumulate products of opposite sign (the split
// multiplies the raw gZ values by +-1), which cancels the bf16 MFMA's floor bias (bilinear.hip).  The slabs of
// all ranges are then summed in fixed order by splitk_reduce.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int PASSES, bool RC = false>   // RC: the gZ tile is rebuilt from its ingredients (struct EdgeRC), not read
__global__ __launch_bounds__(512, 2) void edge_gw_kernel(const float* __restrict__ gZ, long ldg, long gzb,
                                                         const uint4* __restrict__ Eq, float* __restrict__ slab,
                                                         int E, int ncb, int nsteps, int S,
                                                         const float* __restrict__ gmax, const float* __restrict__ emax,
                                                         const EdgeRC rc, int xcd_order) {
  // PASSES == 2: two fp16 planes, three passes; both operands are indexed by the reduction index (the edge slot), so
  // both scales are per tensor: gmax[0] = max |gZ| (from its producer), emax[0] = max |e| (the planes carry 2^k e)
  // PASSES == 1 (edge storage "bf16-mma"): one bf16 pass on the leading planes of both operands (see edge_ge_kernel)
  constexpr bool F16 = PASSES == 2;
  constexpr bool ONE = PASSES == 1;
  constexpr int NP = F16 ? 2 : 3;
  constexpr int HP = NP * 256;
  constexpr int FLUSH = 64;
  __shared__ uint4 Es[2][2 * HP];                        // e planes of one k-step: [half][plane][cb][lane]
  __shared__ __attribute__((aligned(16))) unsigned char Gs[2][2][NP][8192];   // [buffer][column block][plane][32 x 256 B]
  float sG = 1.f, inv_all = 1.f;
  if constexpr (F16) {
    float iG, sE, iE;
    pow2_scale(gmax[0], sG, iG);
    pow2_scale(emax[0], sE, iE);
    inv_all = iG * iE;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kg = lane >> 4;
  const int grp = wave >> 2, wq = wave & 3;              // column block of the pair, 32-column slice in it
  const int npair = ncb / 2;
  // XCD-aware order: consecutive workgroup ids go to different XCDs (own L2 each), and the npair workgroups of one range
  // read the same e planes.  The first 8 * floor(S / 8) ranges are dealt out so that a range's workgroups sit on ONE XCD,
  // next to each other in its dispatch order (the planes then come from HBM once and from that L2 afterwards; in plain
  // order every column-block pair fetched them into its own XCD: 6.5 GB of FETCH_SIZE for 0.77 GB of planes at
  // W2 = 1536); the remaining S % 8 ranges keep the plain order.  Same number of workgroups either way.
  int pair, split;
  {
    const int s8 = S >> 3, body = 8 * s8 * npair;
    if (xcd_order && (int)blockIdx.x < body) {
      const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
      pair = j % npair;
      split = xcd * s8 + j / npair;
    } else if (xcd_order) {
      const int r = blockIdx.x - body;
      pair = r % npair;
      split = 8 * s8 + r / npair;
    } else {
      pair = blockIdx.x % npair;
      split = blockIdx.x / npair;
    }
  }
  // ranges in units of two k-steps (the loop is unrolled by two without a tail); a k-step past nsteps holds slots >= E,
  // which contribute zeros, and its e planes are the zero padding of the last 128-slot block
  const int npairs = (nsteps + 1) / 2;
  const int ks0 = 2 * (int)((long)npairs * split / S), ks1 = 2 * (int)((long)npairs * (split + 1) / S);
  const int cb128 = pair * 2 + grp;
  const float* gblk = gZ + (long)cb128 * gzb;
  const int gt = tid & 255;                              // thread within the column block's group
  // RC: this thread's four columns 128 cb128 + 4 (gt & 31) never change: head, half, mask word and bit offset are fixed
  const int rc_col = 128 * cb128 + 4 * (gt & 31);
  const bool rc_isA = RC && rc_col < rc.HHd;
  const int rc_cc = rc_isA ? rc_col : rc_col - rc.HHd;
  const int rc_h = RC ? rc_cc / rc.Hd : 0;
  const int rc_word = rc_col >> 5, rc_bit = rc_col & 31;
  const float* rc_coef = RC ? (rc_isA ? rc.ga : rc.alpha) + rc_h : nullptr;
  // (attention half: the "row" is the constant wA, fetched through the same load as a gS row -- selecting between a
  // register copy and a global pointer makes the compiler spill the copy and load it back through flat memory)
  const int last_row = E - 1;

  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // (se*: the tile stored at the end of this iteration, loaded during the previous one; te*: the one after it)
  uint4 se0, se1, se2, te0, te1, te2;
  const int p1 = tid + 512, p2 = tid + 1024;
  const int p0 = tid < 256 ? tid : HP + tid - 256;      // ONE: where the thread's plane-0 piece lives in the tile
#define GW_ELOAD(ks_, E0_, E1_, E2_)                                                                     \
  {                                                                                                      \
    const long a_ = (ks_) >> 2, s_ = (ks_) & 3;                                                          \
    const uint4* h0 = Eq + ((a_ * 2 + 0) * 4 + s_) * HP;                                                 \
    const uint4* h1 = Eq + ((a_ * 2 + 1) * 4 + s_) * HP;                                                 \
    if (ONE) {                       /* plane 0 of each half: one piece per thread */                      \
      E0_ = tid < 256 ? h0[tid] : h1[tid - 256];                                                         \
    } else {                                                                                             \
      E0_ = h0[tid];                                                                                     \
      E1_ = p1 < HP ? h0[p1] : h1[p1 - HP];                                                              \
      if (NP == 3) E2_ = h1[p2 - HP];                                                                    \
    }                                                                                                    \
  }
  // raw gZ tile pieces: float4 number gt + 256 i of the k-step's contiguous [32][128] tile, i < 4
  float4 r0, r1, r2, r3, s0, s1, s2, s3;
  // RC: (coefficient, mask word) of the four rows in r* / s*, and the destination nodes of the rows one k-step further
  // (the row of gS is a dependent load: its index is fetched a k-step before the row itself)
  float rk0 = 0.f, rk1 = 0.f, rk2 = 0.f, rk3 = 0.f, sk0 = 0.f, sk1 = 0.f, sk2 = 0.f, sk3 = 0.f;
  unsigned rm0 = 0, rm1 = 0, rm2 = 0, rm3 = 0, sm0 = 0, sm1 = 0, sm2 = 0, sm3 = 0;
  int dn0 = 0, dn1 = 0, dn2 = 0, dn3 = 0;
#define GW_D1(ks_, i_, D_)                                                                               \
  {                                                                                                      \
    const long t = (long)(ks_) * 32 + ((gt + 256 * (i_)) >> 5);                                          \
    D_ = rc.dst[t < E ? t : last_row];                                                                   \
  }
  // (also in the attention half, where they are not used: no branches inside the loop body, see GW_ITER)
#define GW_DLOAD(ks_) { if constexpr (RC) { GW_D1(ks_, 0, dn0) GW_D1(ks_, 1, dn1) GW_D1(ks_, 2, dn2) GW_D1(ks_, 3, dn3) } }
  // row pointers of k-step ks_ from the dn* loaded for it (attention half: the constant wA -- through the same load,
  // because selecting between a register copy and a pointer makes the compiler keep the copy in scratch)
  const float *gp0 = nullptr, *gp1 = nullptr, *gp2 = nullptr, *gp3 = nullptr;
#define GW_PTRS()                                                                                        \
  {                                                                                                      \
    if constexpr (RC) {                                                                                  \
      gp0 = rc_isA ? rc.wA + rc_cc : rc.gS + (long)dn0 * rc.HHd + rc_cc;                                 \
      gp1 = rc_isA ? rc.wA + rc_cc : rc.gS + (long)dn1 * rc.HHd + rc_cc;                                 \
      gp2 = rc_isA ? rc.wA + rc_cc : rc.gS + (long)dn2 * rc.HHd + rc_cc;                                 \
      gp3 = rc_isA ? rc.wA + rc_cc : rc.gS + (long)dn3 * rc.HHd + rc_cc;                                 \
    }                                                                                                    \
  }
#define GW_G1(ks_, i_, R_, K_, M_, P_)                                                                   \
  {                                                                                                      \
    const int idx = gt + 256 * (i_);                                                                     \
    const long t = (long)(ks_) * 32 + (idx >> 5);                                                        \
    if constexpr (RC) {                                                                                  \
      const long tc = t < E ? t : last_row;                                                              \
      R_ = *reinterpret_cast<const float4*>(P_);                                                         \
      const float kk_ = rc_coef[tc * rc.H];              /* unconditional load, then the select */        \
      K_ = t < E ? kk_ : 0.f;                                                                            \
      M_ = rc.mask[tc * rc.nw + rc_word];                                                                \
    } else {                                                                                             \
      R_ = t < E ? *reinterpret_cast<const float4*>(gblk + t * ldg + 4 * (idx & 31))                     \
                 : make_float4(0.f, 0.f, 0.f, 0.f);                                                      \
    }                                                                                                    \
  }
  // Order inside one group of loads (vmcnt retires in order): the row indices of the k-step AFTER this one first --
  // they are needed (as addresses) at the top of the next iteration, and being the oldest of their group they can be
  // waited for with everything behind them still in flight --, then the e tile, then the rows themselves.
#define GW_LOADS(ks_, E0_, E1_, E2_, A_, B_, C_, D_, KA_, KB_, KC_, KD_, MA_, MB_, MC_, MD_)              \
  {                                                                                                      \
    GW_PTRS()                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    GW_DLOAD((ks_) + 1)                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    GW_ELOAD(ks_, E0_, E1_, E2_);                                                                        \
    GW_G1(ks_, 0, A_, KA_, MA_, gp0) GW_G1(ks_, 1, B_, KB_, MB_, gp1)                                    \
    GW_G1(ks_, 2, C_, KC_, MC_, gp2) GW_G1(ks_, 3, D_, KD_, MD_, gp3)                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
  }
  // split one float4 (row idx >> 5, columns 4 (idx & 31) ...) and store 8 bytes per plane into image (b)
#define GW_S1(i_, R_, K_, M_, buf_, sg_)                                                                 \
  {                                                                                                      \
    const int idx = gt + 256 * (i_);                                                                     \
    const int row = idx >> 5, c4 = idx & 31;                                                             \
    const int off = 256 * row + 16 * ((c4 >> 1) ^ (((row & 3) << 2) | ((row >> 2) & 3))) + 8 * (c4 & 1); \
    uint2 x1, x2, x3;                                                                                    \
    float4 g_ = R_;                                                                                      \
    if constexpr (RC) {   /* the stored value was (coefficient * vector) * d; the sign and the power-of-two scale */ \
      const unsigned mb_ = (M_) >> rc_bit;     /* of the split ride on the coefficient (exact) */           \
      const float ks_ = (K_) * (F16 ? (sg_) * sG : (sg_));                                                \
      /* __fmul_rn: the stored value was ROUNDED; contracted into the split's v - hi it would not be */   \
      g_.x = __fmul_rn(ks_ * g_.x, rc_d(mb_, 0)); g_.y = __fmul_rn(ks_ * g_.y, rc_d(mb_, 1));            \
      g_.z = __fmul_rn(ks_ * g_.z, rc_d(mb_, 2)); g_.w = __fmul_rn(ks_ * g_.w, rc_d(mb_, 3));            \
    }                                                                                                    \
    if constexpr (RC && F16) {                                                                           \
      split2_pair_f16(g_.x, g_.y, x1.x, x2.x);                                                           \
      split2_pair_f16(g_.z, g_.w, x1.y, x2.y);                                                           \
    } else if constexpr (RC) {                                                                           \
      split3_pair(g_.x, g_.y, x1.x, x2.x, x3.x);                                                         \
      split3_pair(g_.z, g_.w, x1.y, x2.y, x3.y);                                                         \
    } else if constexpr (F16) {                                                                          \
      const float m_ = (sg_) * sG;                                                                       \
      split2_pair_f16(g_.x * m_, g_.y * m_, x1.x, x2.x);                                                 \
      split2_pair_f16(g_.z * m_, g_.w * m_, x1.y, x2.y);                                                 \
    } else {                                                                                             \
      split3_pair(g_.x * sg_, g_.y * sg_, x1.x, x2.x, x3.x);                                             \
      split3_pair(g_.z * sg_, g_.w * sg_, x1.y, x2.y, x3.y);                                             \
    }                                                                                                    \
    *reinterpret_cast<uint2*>(&Gs[buf_][grp][0][off]) = x1;                                              \
    if (!ONE) *reinterpret_cast<uint2*>(&Gs[buf_][grp][1][off]) = x2;                                    \
    if (PASSES >= 6) *reinterpret_cast<uint2*>(&Gs[buf_][grp][NP - 1][off]) = x3;                        \
  }
  // transposed fragment of plane pl_, column block nb_ of this wave: k = 8 kg + j  <->  slot 8 kg + j
  const int l16q = n16 >> 2, l16p = n16 & 3;
#define GW_TRADDR(nb_, h_)                                                                               \
  ({                                                                                                     \
    const int row = 8 * kg + 4 * (h_) + l16q;                                                            \
    const int ch = 4 * wq + 2 * (nb_) + (l16p >> 1);                                                     \
    256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) + 8 * (l16p & 1);                      \
  })
  const int tr00 = GW_TRADDR(0, 0), tr01 = GW_TRADDR(0, 1), tr10 = GW_TRADDR(1, 0), tr11 = GW_TRADDR(1, 1);
#define GW_TRREAD(buf_, pl_, o0_, o1_)                                                                   \
  ({                                                                                                     \
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(&Gs[buf_][grp][pl_][o0_]));   \
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(&Gs[buf_][grp][pl_][o1_]));   \
    typedef short s16x8 __attribute__((ext_vector_type(8)));                                             \
    const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                         \
    __builtin_bit_cast(bf16x8, both);                                                                    \
  })
#define GW_MFMA1(F1_, F2_, F3_, Q1_, Q2_, Q3_, P_)                                                       \
  {                                                                                                      \
    if (PASSES >= 6) {                                                                                   \
      P_ = mma16<F16>(F3_, Q1_, P_);                                                                     \
      P_ = mma16<F16>(F1_, Q3_, P_);                                                                     \
      P_ = mma16<F16>(F2_, Q2_, P_);                                                                     \
    }                                                                                                    \
    if (!ONE) {                                                                                          \
      P_ = mma16<F16>(F2_, Q1_, P_);                                                                     \
      P_ = mma16<F16>(F1_, Q2_, P_);                                                                     \
    }                                                                                                    \
    P_ = mma16<F16>(F1_, Q1_, P_);                                                                       \
  }
  // the workgroup's slab tile: rows = its 256 columns of gZ, 128 outputs each
  float* tile = slab + ((long)split * ncb * 128 + (long)cb128 * 128 + 32 * wq) * 128;
#define GW_FLUSH(first_, sg_)                                                                            \
  {                                                                                                      \
    _Pragma("unroll") for (int g = 0; g < 8; ++g)                                                        \
    _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) {                                                   \
      float4* o = reinterpret_cast<float4*>(tile + (long)(16 * nb + n16) * 128 + 16 * g + 4 * kg);       \
      const f32x4 v = acc[2 * g + nb];                                                                   \
      const float fs_ = F16 ? (sg_) * inv_all : (sg_);                                                   \
      float4 w = make_float4(v[0] * fs_, v[1] * fs_, v[2] * fs_, v[3] * fs_);                            \
      if (!(first_)) { const float4 u = *o; w.x += u.x; w.y += u.y; w.z += u.z; w.w += u.w; }            \
      *o = w;                                                                                            \
      acc[2 * g + nb] = f32x4{0.f, 0.f, 0.f, 0.f};                                                       \
    }                                                                                                    \
  }
  if (ks1 <= ks0) {   // empty range: the reducer still sums this slab
    GW_FLUSH(true, 1.f);
    return;
  }
  GW_DLOAD(ks0);
  GW_LOADS(ks0, se0, se1, se2, r0, r1, r2, r3, rk0, rk1, rk2, rk3, rm0, rm1, rm2, rm3);
  if (ONE) { Es[0][p0] = se0; }
  else {
    Es[0][tid] = se0; Es[0][p1] = se1;
    if (NP == 3) Es[0][p2] = se2;
  }
  GW_S1(0, r0, rk0, rm0, 0, 1.f) GW_S1(1, r1, rk1, rm1, 0, 1.f) GW_S1(2, r2, rk2, rm2, 0, 1.f) GW_S1(3, r3, rk3, rm3, 0, 1.f)
  r0 = r1 = r2 = r3 = make_float4(0.f, 0.f, 0.f, 0.f);
  rk0 = rk1 = rk2 = rk3 = 0.f;
  GW_LOADS(ks0 + 1, se0, se1, se2, r0, r1, r2, r3, rk0, rk1, rk2, rk3, rm0, rm1, rm2, rm3);   // ranges are even
  __syncthreads();
  bool first = true;
  // One iteration: the loads of k-step ks + 2 go into the (S, TE) register sets, the (R, UE) sets -- loaded one iteration
  // ago -- are consumed.  Unrolled by two with the sets swapped instead of copied: a register copy of an in-flight load
  // is a wait for it, and such copies at the loop end drained vmcnt every k-step.
#define GW_ITER(ks_, buf_, R0, R1, R2, R3, RK0, RK1, RK2, RK3, RM0, RM1, RM2, RM3,                           \
                S0, S1, S2, S3, SK0, SK1, SK2, SK3, SM0, SM1, SM2, SM3, UE0, UE1, UE2, TE0, TE1, TE2)    \
  {                                                                                                      \
    const int rel = (ks_) - ks0;                                                                         \
    /* sign of the flush group the NEXT k-step belongs to (its tile is split during this iteration) */     \
    const float sgn_next = (((rel + 1) / FLUSH) & 1) ? -1.f : 1.f;                                       \
    /* no branches around the loads (past the range: the last k-step again, never used): with conditional loads */ \
    /* the compiler's wait-count bookkeeping turns conservative at every merge */                         \
    const int kl_ = (ks_) + 2 < ks1 ? (ks_) + 2 : ks1 - 1;                                               \
    GW_LOADS(kl_, TE0, TE1, TE2, S0, S1, S2, S3, SK0, SK1, SK2, SK3, SM0, SM1, SM2, SM3);                \
    const bf16x8* es = reinterpret_cast<const bf16x8*>(&Es[buf_][lane]);                                 \
    /* this wave's two transposed gZ fragments (32 columns x 32 slots), three planes each */               \
    bf16x8 qa1 = GW_TRREAD(buf_, 0, tr00, tr01), qa2, qa3;                                               \
    bf16x8 qb1 = GW_TRREAD(buf_, 0, tr10, tr11), qb2, qb3;                                               \
    if (!ONE) { qa2 = GW_TRREAD(buf_, 1, tr00, tr01); qb2 = GW_TRREAD(buf_, 1, tr10, tr11); }            \
    if (PASSES >= 6) { qa3 = GW_TRREAD(buf_, 2, tr00, tr01); qb3 = GW_TRREAD(buf_, 2, tr10, tr11); }     \
    bf16x8 f1 = es[0], f2, f3;                                                                           \
    if (!ONE) f2 = es[256];                                                                              \
    if (PASSES >= 6) f3 = es[512];                                                                       \
    _Pragma("unroll") for (int g = 0; g < 8; ++g) {      /* 16 outputs k = 16 g ...: e fragment g = (half, cb) */ \
      bf16x8 n1, n2, n3;                                                                                 \
      if (g < 7) {                                                                                       \
        const int o = ((g + 1) >> 2) * HP + ((g + 1) & 3) * 64;                                          \
        n1 = es[o];                                                                                      \
        if (!ONE) n2 = es[o + 256];                                                                      \
        if (PASSES >= 6) n3 = es[o + 512];                                                               \
      }                                                                                                  \
      GW_MFMA1(f1, f2, f3, qa1, qa2, qa3, acc[2 * g + 0]);                                               \
      if (g == 0) GW_S1(0, R0, RK0, RM0, (buf_) ^ 1, sgn_next)                                           \
      if (g == 2) GW_S1(1, R1, RK1, RM1, (buf_) ^ 1, sgn_next)                                           \
      if (g == 4) GW_S1(2, R2, RK2, RM2, (buf_) ^ 1, sgn_next)                                           \
      if (g == 6) GW_S1(3, R3, RK3, RM3, (buf_) ^ 1, sgn_next)                                           \
      GW_MFMA1(f1, f2, f3, qb1, qb2, qb3, acc[2 * g + 1]);                                               \
      if (g < 7) { f1 = n1; f2 = n2; f3 = n3; }                                                          \
    }                                                                                                    \
    if (ONE) { Es[(buf_) ^ 1][p0] = UE0; }                                                               \
    else { Es[(buf_) ^ 1][tid] = UE0; Es[(buf_) ^ 1][p1] = UE1; if (NP == 3) Es[(buf_) ^ 1][p2] = UE2; } \
    if ((rel + 1) % FLUSH == 0 || (ks_) + 1 == ks1) {                                                    \
      const float sg = ((rel / FLUSH) & 1) ? -1.f : 1.f;                                                 \
      GW_FLUSH(first, sg);                                                                               \
      first = false;                                                                                     \
    }                                                                                                    \
    /* LDS writes of this wave done, then the barrier -- NOT __syncthreads(): its fence also drains vmcnt, i.e. the */ \
    /* operand loads just issued two k-steps ahead */                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
    __builtin_amdgcn_s_barrier();                                                                        \
    asm volatile("" ::: "memory");                                                                       \
  }
  for (int ks = ks0; ks < ks1; ks += 2) {
    GW_ITER(ks, 0, r0, r1, r2, r3, rk0, rk1, rk2, rk3, rm0, rm1, rm2, rm3,
            s0, s1, s2, s3, sk0, sk1, sk2, sk3, sm0, sm1, sm2, sm3, se0, se1, se2, te0, te1, te2)
    GW_ITER(ks + 1, 1, s0, s1, s2, s3, sk0, sk1, sk2, sk3, sm0, sm1, sm2, sm3,
            r0, r1, r2, r3, rk0, rk1, rk2, rk3, rm0, rm1, rm2, rm3, te0, te1, te2, se0, se1, se2)
  }
#undef GW_ITER
#undef GW_LOADS
#undef GW_PTRS
#undef GW_ELOAD
#undef GW_G1
#undef GW_D1
#undef GW_DLOAD
#undef GW_S1
#undef GW_TRADDR
#undef GW_TRREAD
#undef GW_MFMA1
#undef GW_FLUSH
}

bool edge_ge_fast(int Ce, int W2, long ldg, long gzb, long ldo, const void* gZ, const void* out) {
  return bilinear_mode() != 0 && Ce == 128 && W2 % 128 == 0 && gzb != 0 && (gzb % 4) == 0 && (ldg % 4) == 0 &&
         (ldo % 4) == 0 && ((((uintptr_t)gZ) | ((uintptr_t)out)) & 15) == 0;
}

// We: element (col, k) at We[col * s_col + k * s_out] (col < W2 inputs, k < 128 outputs).  Wq: edge_z_wq_floats(W2)
// floats of workspace.  bias [128] or null (added to the product, before any accumulation into `out`).
// amax (f16x3 mode only): device pointer to max |gZ|; without it the bf16x6 form runs.
int edge_ge_launch(const float* gZ, long ldg, long gzb, const float* We, long s_col, long s_out, float* Wq, int W2,
                   float* out, long ldo, const int* scatter, int E, int accumulate, const float* bias,
                   hipStream_t stream, const float* amax, const EdgeRC* rc) {
  if (E <= 0) return CGAT_OK;
  const int ncb = W2 / 128;
  // f16x3 form: with max |gZ| known, for a weight whose 128 outputs are contiguous (s_out == 1: the backward products)
  // or whose W2 inputs are (s_col == 1: a forward nn.Linear weight [128, W2], round 3)
  const bool out_contig = s_out == 1 && (s_col % 4) == 0, in_contig = s_col == 1 && (s_out % 4) == 0 && W2 % 128 == 0;
  const bool f16 = bilinear_mode() == 2 && amax && (out_contig || in_contig) && (((uintptr_t)We) & 15) == 0;
  // operand (a = column block, b = column in block, c = output k) = We[(128 a + b) * s_col + c * s_out]
  if (f16) {   // per-tensor weight scale: max |We| behind the two planes
    float* wmax = Wq + (size_t)ncb * 16384;
    CGAT_TRY(fill_launch(wmax, 0.f, 1, stream));
    if (out_contig) CGAT_TRY(absmax_rows128_launch(We, s_col, W2, wmax, stream));
    else
      for (int j = 0; j < ncb; ++j) CGAT_TRY(absmax_rows128_launch(We + 128 * j, s_out, 128, wmax, stream));
    CGAT_TRY(prepare_T_f16_scaled_launch(We, Wq, ncb, 128 * s_col, s_col, s_out, wmax, stream, /*alternate=*/1));
  } else {
    CGAT_TRY(prepare_T_bf16_launch(We, Wq, ncb, 128 * s_col, s_col, s_out, /*alternate=*/1, stream));
  }
  CGAT_PROF(scatter ? "edge_ge" : "rows_ge", stream);   // the per-edge launch / node-side and dense-layer uses
  const int grid = cdiv(E, 256);
  const EdgeRC none = {};
#define GE_GO(P_, R_)                                                                                                 \
  hipLaunchKernelGGL((edge_ge_kernel<P_, R_>), dim3(grid), dim3(512), 0, stream, gZ, ldg, gzb, (const uint4*)Wq, ncb, \
                     out, ldo, scatter, E, accumulate, bias, amax, R_ ? *rc : none, HeadBatch{})
  // (the per-EDGE launch only -- rc: the rebuilt gZ rows -- and never in the fp16 mode, which has its own bf16 storage form)
  const bool one = rc && !f16 && edge_mma_bf16() && bilinear_mode() != 3;
  if (one) GE_GO(1, true);
  else if (rc) { if (f16) GE_GO(2, true); else if (bilinear_mode() != 3) GE_GO(6, true); else GE_GO(3, true); }
  else { if (f16) GE_GO(2, false); else if (bilinear_mode() != 3) GE_GO(6, false); else GE_GO(3, false); }
#undef GE_GO
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// K-split form of the product above for FEW row tiles beside a long K (round 6; the harness' shipped batch: g_edge_attr at
// 30 720 edges is 120 workgroups walking 20 column blocks, the node-side input gradients at 1 280 atoms 5 workgroups --
// or, as small-row programs, 0.84 GFLOP on the fp64 matrix instruction: 100-160 us a launch).  The column blocks are dealt
// to S groups (grid.y), group s multiplies its blocks into slab s ([E, 128] each, at slabs + s * E * 128, scattered rows as
// `out` would be); the caller adds the slabs in group order (sum_slabs_launch).  A group holds an EVEN number of blocks,
// so that a block's position in its group has the parity of its position in the row (the sign alternation above).
// Returns the number of groups a launch of this shape takes; 1 = not worth it / not available (then use edge_ge_launch).
int edge_ge_ksplit_groups(int E, int W2) {
  if (bilinear_mode() != 4 && bilinear_mode() != 6) return 1;
  const int ncb = W2 / 128, tiles = cdiv(E, 256);
  if (W2 % 128 != 0 || E < 1024 || tiles >= 192) return 1;     // (the reference's fixtures stay on the unsplit forms)
  int best = 1;
  for (int S = 2; S <= ncb; ++S)
    if (ncb % S == 0 && ((ncb / S) & 1) == 0) { best = S; if (tiles * S >= 192) break; }
  return best;
}
int edge_ge_ksplit_launch(const float* gZ, long ldg, long gzb, const float* We, long s_col, long s_out, float* Wq, int W2,
                          float* slabs, const int* scatter, int E, int S, hipStream_t stream, const EdgeRC* rc) {
  if (E <= 0) return CGAT_OK;
  const int ncb = W2 / 128;
  CGAT_CHECK_ARG(S >= 1 && ncb % S == 0 && (S == 1 || ((ncb / S) & 1) == 0) && (bilinear_mode() == 4 || bilinear_mode() == 6),
                 "edge_ge_ksplit: %d groups of %d column blocks", S, ncb);
  CGAT_TRY(prepare_T_bf16_launch(We, Wq, ncb, 128 * s_col, s_col, s_out, /*alternate=*/1, stream));
  CGAT_PROF(scatter ? "edge_ge" : "rows_ge", stream);
  const EdgeRC none = {};
  const int ncb_g = ncb / S;
  const HeadBatch hb = {(long)ncb_g * gzb, (long)ncb_g * 6144, 0, (long)E * 128, 0, 0, (long)ncb_g * 4};
  if (rc)     // the rows rebuilt from their ingredients (struct EdgeRC): a group's k-steps start at ks0
    hipLaunchKernelGGL((edge_ge_kernel<6, true>), dim3(cdiv(E, 256), S), dim3(512), 0, stream, gZ, ldg, gzb, (const uint4*)Wq,
                       ncb_g, slabs, 128l, scatter, E, 0, (const float*)nullptr, (const float*)nullptr, *rc, hb);
  else
    hipLaunchKernelGGL((edge_ge_kernel<6, false>), dim3(cdiv(E, 256), S), dim3(512), 0, stream, gZ, ldg, gzb, (const uint4*)Wq,
                       ncb_g, slabs, 128l, scatter, E, 0, (const float*)nullptr, (const float*)nullptr, none, hb);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// out[t, :] (+)= sum over the ncb 128-column blocks of x[t, :] times an ALREADY prepared six-pass image (block a at
// Wq + a * 24576 floats, odd blocks negated: prepare_T_bf16_launch / _heads_launch with alternate = 1) -- the per-node
// second layer of the message network, whose H per-head weights are not one affine operand (layers.hip)
int edge_ge_prepared_launch(const float* x, long ldx, const void* Wq, int ncb, float* out, long ldo, int rows,
                            int accumulate, hipStream_t stream) {
  if (rows <= 0) return CGAT_OK;
  CGAT_PROF("rows_ge", stream);
  const EdgeRC none = {};
  hipLaunchKernelGGL((edge_ge_kernel<6, false>), dim3(cdiv(rows, 256)), dim3(512), 0, stream, x, ldx, 128l, (const uint4*)Wq,
                     ncb, out, ldo, (const int*)nullptr, rows, accumulate, (const float*)nullptr, (const float*)nullptr, none,
                     HeadBatch{});
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// `heads` products y_h = x_h W_h^T + b_h (x_h = x + h * s_x: W2 columns of a wider matrix; W_h = W + h * s_w: [128, W2]
// row-major with leading dimension W2, contiguous; y_h = y + h * s_y) in three launches instead of 4-5 per head: the
// heads' weight maxima and fp16 planes through the batched preparation of the contraction kernels (bilinear.hip), then
// ONE launch of the kernel above with grid.y = head.  f16x3 mode with max |x| known (amax), heads <= TPREP_MAX.
// ws: heads * edge_ge_heads_image_floats(W2) + bilinear_prepare_T_batch_ws_floats(heads) floats.
// (the six-pass image is the larger one: three bf16 planes per 128 x 128 block)
size_t edge_ge_heads_image_floats(int W2) { return (size_t)(W2 / 128) * 24576 + 4; }
bool edge_ge_heads_fast(int heads, int W2, long ldx, long ldy, long ldw, const void* x, const void* w, const void* y,
                        const float* amax) {
  const bool f16 = bilinear_mode() == 2 && amax;
  const bool six = bilinear_mode() == 4 || bilinear_mode() == 6;      // round 6: the 24-bit modes batch their heads too
  return (f16 || six) && heads >= 1 && heads <= TPREP_MAX && W2 % 128 == 0 && ldw == W2 &&
         (((uintptr_t)w) & 15) == 0 && edge_ge_fast(128, W2, ldx, 128, ldy, x, y);
}
int edge_ge_heads_launch(int heads, const float* x, long ldx, long s_x, const float* W, long s_w, const float* bias,
                         long s_bias, float* y, long ldy, long s_y, int E, int W2, float* ws, hipStream_t stream,
                         const float* amax) {
  if (E <= 0 || heads <= 0) return CGAT_OK;
  const int ncb = W2 / 128;
  const size_t img = edge_ge_heads_image_floats(W2);
  const HeadBatch hb = {s_x, (long)(img / 4), s_bias, s_y, 0};
  const EdgeRC none = {};
  if (bilinear_mode() != 2) {
    // operand (a = column block, b = column in block, c = output k) = W_h[c * W2 + 128 a + b], as edge_ge_launch's
    CGAT_TRY(prepare_T_bf16_heads_launch(W, ws, ncb, 128, 1, W2, /*alternate=*/1, heads, s_w, (long)img, stream));
    CGAT_PROF("rows_ge", stream);
    hipLaunchKernelGGL((edge_ge_kernel<6, false>), dim3(cdiv(E, 256), heads), dim3(512), 0, stream, x, ldx, 128l,
                       (const uint4*)ws, ncb, y, ldy, (const int*)nullptr, E, 0, bias, (const float*)nullptr, none, hb);
    CGAT_LAUNCH_CHECK();
    return CGAT_OK;
  }
  float* part = ws + (size_t)heads * img;
  const float* src[TPREP_MAX];
  float* dst[TPREP_MAX];
  for (int h = 0; h < heads; ++h) { src[h] = W + (long)h * s_w; dst[h] = ws + (size_t)h * img; }
  // operand (a = column block, b = column in block, c = output k) = W_h[c * W2 + 128 a + b]: source dims (k, a, b)
  const int rc_ = bilinear_prepare_T_batch(heads, src, dst, 128, ncb, 128, 1, 2, 0, part, stream, /*alternate=*/1);
  if (rc_ != CGAT_OK) return rc_;
  CGAT_PROF("rows_ge", stream);
  hipLaunchKernelGGL((edge_ge_kernel<2, false>), dim3(cdiv(E, 256), heads), dim3(512), 0, stream, x, ldx, 128l,
                     (const uint4*)ws, ncb, y, ldy, (const int*)nullptr, E, 0, bias, amax, none, hb);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

bool edge_gw_fast(int Ce, int W2, long ldg, long gzb, const void* gZ) {
  return bilinear_mode() != 0 && Ce == 128 && W2 % 256 == 0 && gzb != 0 && (gzb % 4) == 0 && (ldg % 4) == 0 &&
         (((uintptr_t)gZ) & 15) == 0;
}
static int gw_xcd_order() {   // CGAT_GW_XCD=0: plain workgroup order (A/B switch)
  static const int v = [] { const char* e = getenv("CGAT_GW_XCD"); return (e && e[0] == '0') ? 0 : 1; }();
  return v;
}
static int edge_gw_splits(int W2) {   // ranges x column-block pairs ~ one workgroup per CU
  const int npair = W2 / 256;
  if (npair <= 0) return 1;
  return 256 / npair > 0 ? 256 / npair : 1;
}
// workspace floats: e planes (zero-padded to a multiple of 128 slots) + slabs
size_t edge_gw_ws_floats(int E, int W2) {
  const size_t planes = ((size_t)cdiv(E, 128) * 128 * 128 * 3 + 1) / 2;
  return planes + (size_t)edge_gw_splits(W2) * W2 * 128 + 64;
}

// out[col * ldo + k] = sum_t G[t, col] * e[perm[t] * lde + k],   G[t, 128 a + j] at gZ[t * ldg + a * gzb + j]
// gmax, emax (f16x3 mode only): device pointers to max |gZ| and max |e|; without them the bf16x6 form runs.
int edge_gw_launch(const float* gZ, long ldg, long gzb, const float* e, long lde, const int* perm, int E, int W2,
                   float* ws, float* out, long ldo, hipStream_t stream, const float* gmax, const float* emax,
                   const EdgeRC* rc) {
  if (E <= 0) {
    GemmParams z = gemm_params(W2, 128, 0, nullptr, 1, nullptr, 1, out, ldo);
    return gemm_launch(z, nullptr, 0, stream);   // K = 0: zero fill
  }
  const int ncb = W2 / 128, na = cdiv(E, 128), S = edge_gw_splits(W2);
  float* planes = ws;
  float* slab = ws + (((size_t)na * 128 * 128 * 3 + 1) / 2 + 15) / 16 * 16;
  // operand (a = slot block, b = slot in block, c = k) = e[perm[128 a + b] * lde + c], zero past E
  const bool f16 = bilinear_mode() == 2 && gmax && emax;
  CGAT_TRY(prepare_T_bf16_rows_launch(e, lde, perm, E, planes, na, stream, f16 ? emax : nullptr));
  {
    CGAT_PROF(perm ? "edge_gw" : "rows_gw", stream);
    const int nsteps = cdiv(E, 32);
    const EdgeRC none = {};
#define GW_GO(P_, R_)                                                                                                  \
  hipLaunchKernelGGL((edge_gw_kernel<P_, R_>), dim3(S * (ncb / 2)), dim3(512), 0, stream, gZ, ldg, gzb,                \
                     (const uint4*)planes, slab, E, ncb, nsteps, S, gmax, emax, R_ ? *rc : none, gw_xcd_order())
    const bool one = rc && !f16 && edge_mma_bf16() && bilinear_mode() != 3;
    if (one) GW_GO(1, true);
    else if (rc) { if (f16) GW_GO(2, true); else if (bilinear_mode() != 3) GW_GO(6, true); else GW_GO(3, true); }
    else { if (f16) GW_GO(2, false); else if (bilinear_mode() != 3) GW_GO(6, false); else GW_GO(3, false); }
#undef GW_GO
    CGAT_LAUNCH_CHECK();
  }
  return splitk_reduce_launch(slab, S, W2, 128, out, ldo, stream);
}

// ---------------------------------------------------------------------------------------
//     Gj[n, :] = sum_{edges leaving n} gZ[t, :]          (edge_gj_kernel: the x_j-side segment sum of the rebuilt rows)
// One workgroup per GJ_NODES source nodes, a thread per four columns (256 threads walk the W2 / 4 quads); per edge a thread fetches its
// mask word, its coefficient and -- message half -- four floats of the destination's gS row (the destinations of a
// node's edges are its neighbours in the same crystal: those rows stay in L2), instead of 16 bytes of a 6-KB gZ row
// gathered from HBM.  Edges four at a time so that the dependent chain slot -> destination -> row overlaps.
// ---------------------------------------------------------------------------------------
#define GJ_NODES 4
__global__ __launch_bounds__(256) void edge_gj_kernel(const EdgeRC rc, const int* __restrict__ src_rowptr,
                                                      const int* __restrict__ src_pos, int N, int W2,
                                                      float* __restrict__ Gj, long ldo, float* __restrict__ gjmax) {
  float gmax = 0.f;
  // XCD-aware order: consecutive workgroup ids go to different XCDs (own L2 each), but the gS rows a node gathers are
  // those of its neighbours, i.e. of nearby nodes: give every XCD one contiguous range of nodes, so that a crystal's rows
  // are fetched into one L2 instead of into five to eight (PMC: 3.1 GB of HBM traffic per launch for a 0.26-GB table)
  const int per_xcd = gridDim.x / 8;                       // the grid is a multiple of 8
  const int blk = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int n0 = blk * GJ_NODES, n1 = min(N, n0 + GJ_NODES);
  for (int q = threadIdx.x; 4 * q < W2; q += 256) {      // four columns per thread (W2 = 1536: 1.5 rounds)
    const int col = 4 * q;
    const bool isA = col < rc.HHd;
    const int cc = isA ? col : col - rc.HHd, h = cc / rc.Hd;
    const int word = col >> 5, bit = col & 31;
    const float* coef = (isA ? rc.ga : rc.alpha) + h;
    for (int n = n0; n < n1; ++n) {
      const int r0 = src_rowptr[n], r1 = src_rowptr[n + 1];
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int rb = r0; rb < r1; rb += 4) {
        long t[4];
        float k[4];
        unsigned m[4];
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = src_pos[rb + u < r1 ? rb + u : r1 - 1];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          k[u] = rb + u < r1 ? coef[t[u] * rc.H] : 0.f;
          m[u] = rc.mask[t[u] * rc.nw + word];
          // (attention half: the constant wA through the same load -- see edge_gw_kernel)
          v[u] = *reinterpret_cast<const float4*>(isA ? rc.wA + cc : rc.gS + (long)rc.dst[t[u]] * rc.HHd + cc);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned mb = m[u] >> bit;
          acc.x += (k[u] * v[u].x) * rc_d(mb, 0); acc.y += (k[u] * v[u].y) * rc_d(mb, 1);
          acc.z += (k[u] * v[u].z) * rc_d(mb, 2); acc.w += (k[u] * v[u].w) * rc_d(mb, 3);
        }
      }
      *reinterpret_cast<float4*>(Gj + (long)n * ldo + col) = acc;
      gmax = fmaxf(fmaxf(gmax, fmaxf(fabsf(acc.x), fabsf(acc.y))), fmaxf(fabsf(acc.z), fabsf(acc.w)));
    }
  }
  if (gjmax) block_absmax_commit(gmax, gjmax);
}
int edge_gj_launch(const EdgeRC& rc, const int* src_rowptr, const int* src_pos, int N, int W2, float* Gj, long ldo,
                   hipStream_t stream, float* gjmax) {
  if (N <= 0) return CGAT_OK;
  CGAT_CHECK_ARG(W2 % 4 == 0 && (ldo % 4) == 0 && (((uintptr_t)Gj) & 15) == 0,
                 "edge_gj: W2 = %d must be a multiple of 4 with a 16-byte aligned output", W2);
  CGAT_PROF("edge_gj", stream);
  hipLaunchKernelGGL(edge_gj_kernel, dim3(cdiv(cdiv(N, GJ_NODES), 8) * 8), dim3(256), 0, stream, rc, src_rowptr, src_pos,
                     N, W2, Gj, ldo, gjmax);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}
