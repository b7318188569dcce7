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
template <int PASSES>
__global__ __launch_bounds__(512, 2) void edge_ge_kernel(const float* __restrict__ gZ, long gzb,
                                                         const uint4* __restrict__ Wq, int ncb,
                                                         float* __restrict__ out, long ldo,
                                                         const int* __restrict__ scatter, int E) {
  __shared__ uint4 Bs[2][1536];                  // [buffer][half][plane][cb][lane]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kg = lane >> 4;
  const int row_w = blockIdx.x * 256 + wave * 32;
  const int row_a = row_w + n16, row_b = row_a + 16;
  const long rca = row_a < E ? row_a : E - 1, rcb = row_b < E ? row_b : E - 1;
  const float* ga = gZ + rca * 128 + 8 * kg;     // + a * gzb + 32 s
  const float* gb = gZ + rcb * 128 + 8 * kg;
  const int nk = ncb * 4;

  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // B staging: thread tid moves 16-byte pieces tid, tid + 512, tid + 1024 of the k-step's [half0 | half1] image
  uint4 sb0, sb1, sb2;
  const int p1 = tid + 512, p2 = tid + 1024;
#define GE_BLOAD(ks_)                                                                                    \
  {                                                                                                      \
    const long a_ = (ks_) >> 2, s_ = (ks_) & 3;                                                          \
    const uint4* h0 = Wq + ((a_ * 2 + 0) * 4 + s_) * 768;                                                \
    const uint4* h1 = Wq + ((a_ * 2 + 1) * 4 + s_) * 768;                                                \
    sb0 = h0[tid];                                                                                       \
    sb1 = p1 < 768 ? h0[p1] : h1[p1 - 768];                                                              \
    sb2 = h1[p2 - 768];                                                                                  \
  }
#define GE_BSTORE(buf_)                                                                                  \
  {                                                                                                      \
    Bs[buf_][tid] = sb0; Bs[buf_][p1] = sb1; Bs[buf_][p2] = sb2;                                         \
  }
  // raw gZ (rows a/b, 8 columns each) of the next k-step (ra*, rb*) and of the one after (sa*, sb*): loads are
  // issued two k-steps (~3 us) before their values are split, enough bytes in flight per CU to cover HBM latency
  float4 ra0, ra1, rb0, rb1, sa0, sa1, sb0_, sb1_;
#define GE_ALOAD(ks_, A0_, A1_, B0_, B1_)                                                                \
  {                                                                                                      \
    const long off = (long)((ks_) >> 2) * gzb + 32 * ((ks_) & 3);                                        \
    const float4* pa = reinterpret_cast<const float4*>(ga + off);                                        \
    const float4* pb = reinterpret_cast<const float4*>(gb + off);                                        \
    A0_ = pa[0]; A1_ = pa[1]; B0_ = pb[0]; B1_ = pb[1];                                                  \
  }
#define GE_SPLIT(R0_, R1_, Q1_, Q2_, Q3_)                                                                \
  {                                                                                                      \
    const float v[8] = {R0_.x, R0_.y, R0_.z, R0_.w, R1_.x, R1_.y, R1_.z, R1_.w};                         \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                      \
      __bf16 x1, x2, x3;                                                                                 \
      split3_bf16(v[j], x1, x2, x3);                                                                     \
      Q1_[j] = x1; Q2_[j] = x2; Q3_[j] = x3;                                                             \
    }                                                                                                    \
  }
  bf16x8 qa1, qa2, qa3, qb1, qb2, qb3;           // current k-step's gZ fragments (rows a, b)
  bf16x8 na1, na2, na3, nb1, nb2, nb3;           // next k-step's
  GE_ALOAD(0, ra0, ra1, rb0, rb1);
  GE_BLOAD(0);
  GE_SPLIT(ra0, ra1, qa1, qa2, qa3);
  GE_SPLIT(rb0, rb1, qb1, qb2, qb3);
  GE_BSTORE(0);
  ra0 = ra1 = rb0 = rb1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nk > 1) GE_ALOAD(1, ra0, ra1, rb0, rb1);
  __syncthreads();

#define GE_MFMA1(F1_, F2_, F3_, Q1_, Q2_, Q3_, P_)                                                       \
  {                                                                                                      \
    if (PASSES >= 6) {                                                                                   \
      P_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F3_, Q1_, P_, 0, 0, 0);                               \
      P_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F1_, Q3_, P_, 0, 0, 0);                               \
      P_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F2_, Q2_, P_, 0, 0, 0);                               \
    }                                                                                                    \
    P_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F2_, Q1_, P_, 0, 0, 0);                                 \
    P_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F1_, Q2_, P_, 0, 0, 0);                                 \
    P_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F1_, Q1_, P_, 0, 0, 0);                                 \
  }
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    const bf16x8* bs = reinterpret_cast<const bf16x8*>(&Bs[buf][lane]);
    if (ks + 1 < nk) GE_BLOAD(ks + 1);
    if (ks + 2 < nk) GE_ALOAD(ks + 2, sa0, sa1, sb0_, sb1_);
    // the raw values in ra/rb belong to k-step ks + 1: split them while this step's MFMAs run
    bf16x8 f1 = bs[0], f2 = bs[256], f3;
    if (PASSES >= 6) f3 = bs[512];
#pragma unroll
    for (int g = 0; g < 8; ++g) {                // 16-column output block g = (half, cb)
      bf16x8 n1, n2, n3;
      if (g < 7) {
        const int o = ((g + 1) >> 2) * 768 + ((g + 1) & 3) * 64;
        n1 = bs[o]; n2 = bs[o + 256];
        if (PASSES >= 6) n3 = bs[o + 512];
      }
      GE_MFMA1(f1, f2, f3, qa1, qa2, qa3, acc[2 * g + 0]);
      if (g == 1 && ks + 1 < nk) GE_SPLIT(ra0, ra1, na1, na2, na3);
      GE_MFMA1(f1, f2, f3, qb1, qb2, qb3, acc[2 * g + 1]);
      if (g == 4 && ks + 1 < nk) GE_SPLIT(rb0, rb1, nb1, nb2, nb3);
      if (g < 7) { f1 = n1; f2 = n2; f3 = n3; }
    }
    if (ks + 1 < nk) GE_BSTORE(buf ^ 1);
    __syncthreads();
    qa1 = na1; qa2 = na2; qa3 = na3; qb1 = nb1; qb2 = nb2; qb3 = nb3;
    ra0 = sa0; ra1 = sa1; rb0 = sb0_; rb1 = sb1_;
  }
#undef GE_BLOAD
#undef GE_BSTORE
#undef GE_ALOAD
#undef GE_SPLIT
#undef GE_MFMA1
  // acc[2 g + nb][j] = out[row(nb)][16 g + 4 kg + j]
  const long oa = scatter ? (long)scatter[rca] : rca, ob = scatter ? (long)scatter[rcb] : rcb;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    if (row_a < E) {
      const f32x4 v = acc[2 * g + 0];
      *reinterpret_cast<float4*>(out + oa * ldo + 16 * g + 4 * kg) = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (row_b < E) {
      const f32x4 v = acc[2 * g + 1];
      *reinterpret_cast<float4*>(out + ob * ldo + 16 * g + 4 * kg) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

bool edge_ge_fast(int Ce, int W2, long gzb, long ldo, const void* gZ, const void* out) {
  return bilinear_mode() != 0 && Ce == 128 && W2 % 128 == 0 && gzb != 0 && (gzb % 4) == 0 && (ldo % 4) == 0 &&
         ((((uintptr_t)gZ) | ((uintptr_t)out)) & 15) == 0;
}

// We: element (col, k) at We[col * ldw + k] (col < W2, k < 128).  Wq: edge_z_wq_floats(W2) floats of workspace.
int edge_ge_launch(const float* gZ, long gzb, const float* We, long ldw, float* Wq, int W2, float* out, long ldo,
                   const int* scatter, int E, hipStream_t stream) {
  if (E <= 0) return CGAT_OK;
  const int ncb = W2 / 128;
  // operand (a = column block, b = column in block, c = output k) = We[(128 a + b) * ldw + c]
  CGAT_TRY(prepare_T_bf16_launch(We, Wq, ncb, 128 * ldw, ldw, 1, 0, stream));
  CGAT_PROF("edge_ge", stream);
  const int grid = cdiv(E, 256);
  if (bilinear_mode() == 6)
    hipLaunchKernelGGL(edge_ge_kernel<6>, dim3(grid), dim3(512), 0, stream, gZ, gzb, (const uint4*)Wq, ncb, out, ldo,
                       scatter, E);
  else
    hipLaunchKernelGGL(edge_ge_kernel<3>, dim3(grid), dim3(512), 0, stream, gZ, gzb, (const uint4*)Wq, ncb, out, ldo,
                       scatter, E);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}
