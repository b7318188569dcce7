// Edge pre-activations of both message networks, the per-edge part of the operand-split first conv
// (reference CGAT.py:96 MultiHeadNetwork.fc_in applied to cat([x_i, edge_attr, x_j]), CGAT.py:316-318):
//     Z[t, :] = W_e e[perm[t]] + Pi[dst[t], :] + Pj[src[t], :]            t = destination-sorted edge slot
// fused with the attention logits of MH_A (CGAT.py:97-98: fc_out(LeakyReLU(fc_in(.))), one output per head):
//     a[t, h] = b_A[h] + sum_j leaky(Z[t, h*Hd + j]) * w_A[h*Hd + j]
//
// Shape: E x 1536 outputs with K = Ce = 128.  The product runs on the matrix cores in the split-bf16
// arithmetic of bilinear.hip (three bf16 planes per fp32 operand, six v_mfma_f32_16x16x32_bf16 passes,
// fp32 accumulation: fp32-equivalent), organised like the hypernetwork kernel: a wave's 32 edge rows are
// split once and stay in 96 VGPRs, the pre-split weight column blocks stream through a 4-slot LDS ring by
// LDS-DMA (12-KB chunks = one 32-deep k-step of 64 columns), fragments are read one group ahead.  The
// kernel is bound by its epilogue traffic -- per edge 6 KB of Z written and 6 KB of Pj gathered (Pi rows
// are shared by the edges of a destination segment and hit L2) -- so two 4-wave workgroups share a CU:
// while one gathers/stores a 64-column slice, the other one runs MFMAs.
//
// vmcnt bookkeeping: the ring's counted waits assume that the three LDS-DMA loads of the current
// iteration are the youngest vector-memory operations; the epilogue's ordinary loads and stores are older
// than the next iteration's LDS-DMA loads, so a counted wait can only be stricter than needed, never looser
// (vmcnt retires in order).  RAW/WAR reasoning of the ring: bilinear.hip.
//
// Packed math: when hipcc SLP-packs the two rows' logit accumulations into v_pk_mul/v_pk_fma_f32, the logits come
// out slightly wrong and differ from run to run on MI355X (Z, computed from the same registers, stays bit-exact;
// found by tests/test_hip_golden.py::test_determinism_bitwise, bisected with an empty asm between the two
// accumulations).  This file is therefore built with -fno-slp-vectorize and the two accumulators are pinned apart.
//
// Kernels in this file (round 5): edge_z_kernel (below: 128-row workgroups; the per-node projections, the dense layer at
// width 128, and the per-edge launch of shapes / modes the wide-tile kernels do not take), edge_z6w_kernel (the per-edge
// launch of the default 24-bit modes: the same six-pass arithmetic on 256-row workgroups, bit-identical, 3.20 -> 2.84 ms),
// edge_zc_kernel (the per-edge launch in the h + l + t arithmetic, opt-in), edge_zx_kernel (f16x3 mode, K = 256).
#include "common.h"
#include "kernels.h"
#include "mfma_bf16.h"

// ZB (the per-edge launch only): Z is stored as bf16 (the "bf16" edge-storage mode, BASELINE configs[4]; ldz counts
// elements either way).  The logits are formed from the unrounded values.
template <int PASSES, bool ADDS, bool ZB = false>
__global__ __launch_bounds__(256, 2) void edge_z_kernel(const float* __restrict__ e, long lde,
                                                        const int* __restrict__ perm, const uint4* __restrict__ Wq,
                                                        int ncb, const float* __restrict__ Pi,
                                                        const int* __restrict__ dsti, const float* __restrict__ Pj,
                                                        const int* __restrict__ srci, long ld_add,
                                                        float* __restrict__ Z, long ldz, int E,
                                                        const float* __restrict__ wA, const float* __restrict__ bA,
                                                        int H, int cb_per_head, float* __restrict__ a_out, int act,
                                                        int accumulate, float* __restrict__ omax,
                                                        const float* __restrict__ dact, long ld_dact, HeadBatch hb) {
  // grid.y = head of a multi-head second layer (linear128_heads_launch): every operand moves by its per-head offset
  e += (long)blockIdx.y * hb.in;
  Wq += (long)blockIdx.y * hb.w;
  Z += (long)blockIdx.y * hb.out;
  if (Pi) Pi += (long)blockIdx.y * hb.bias;
  if (dact) dact += (long)blockIdx.y * hb.dact;
  if (ADDS) Pj += (long)blockIdx.y * hb.add2;
  constexpr bool F16 = PASSES == 2;             // two fp16 planes, three passes (mfma_bf16.h): rows scaled per row,
  constexpr int NP = F16 ? 2 : 3;               // the weight per 128-column block (wmax behind the planes)
  constexpr int CH16 = NP * 4 * 64;             // 16-byte pieces per chunk = 12 KB (8 KB)
  // DEEP (the per-edge launch in the fp16 form): 8 ring slots, vmcnt allowances that let the epilogue's stores stay
  // in flight, all gathers of a slice before its first store, unconditional stores (see edge_zx_kernel's header)
  constexpr bool DEEP = F16 && ADDS;
  constexpr bool ALT = true;                    // odd k-steps into a second, subtracted accumulator set (see below)
  constexpr int RING = DEEP ? 8 : 4;
  __shared__ uint4 smem[RING * CH16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kg = lane >> 4;
  const int row_w = blockIdx.x * 128 + wave * 32;
  const int row_a = row_w + n16, row_b = row_a + 16;
  const int rca = row_a < E ? row_a : E - 1, rcb = row_b < E ? row_b : E - 1;
  const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const unsigned wave_t = __builtin_amdgcn_readfirstlane(sbase + wave * 1024);
  const bf16x8* ring = reinterpret_cast<const bf16x8*>(smem) + lane;
  const unsigned t_off = (unsigned)tid * 16;
  const long last_chunk = (long)ncb * 8 - 1;

  // the lane's two edge rows, split once: q[plane][2 s + nb] holds e[row(nb), 32 s + 8 kg + 0..7]
  bf16x8 q1[8], q2[8], q3[F16 ? 1 : 8];
  float rs_a = 1.f, rs_b = 1.f;                 // F16: 1 / scale of the lane's two rows
  const float* wmax = reinterpret_cast<const float*>(Wq + (long)ncb * 8 * CH16);   // F16: max |W| per column block
  {
    const long ea = perm ? perm[rca] : rca, eb = perm ? perm[rcb] : rcb;
    if constexpr (F16) {
      float qv[2][32];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float4* qp = reinterpret_cast<const float4*>(e + (nb ? eb : ea) * lde + 32 * s + 8 * kg);
          const float4 t0 = qp[0], t1 = qp[1];
          qv[nb][8 * s + 0] = t0.x; qv[nb][8 * s + 1] = t0.y; qv[nb][8 * s + 2] = t0.z; qv[nb][8 * s + 3] = t0.w;
          qv[nb][8 * s + 4] = t1.x; qv[nb][8 * s + 5] = t1.y; qv[nb][8 * s + 6] = t1.z; qv[nb][8 * s + 7] = t1.w;
        }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) m = fmaxf(m, fabsf(qv[nb][j]));
        m = fmaxf(m, __shfl_xor(m, 16));        // the row's 128 values live in the four lanes n16 + 16 kg
        m = fmaxf(m, __shfl_xor(m, 32));
        float sq, iq;
        pow2_scale(m, sq, iq);
        (nb ? rs_b : rs_a) = iq;
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
          const float4* qp = reinterpret_cast<const float4*>(e + (nb ? eb : ea) * lde + 32 * s + 8 * kg);
          const float4 t0 = qp[0], t1 = qp[1];
          const float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
          split3_x8(v, q1[2 * s + nb], q2[2 * s + nb], q3[2 * s + nb]);
        }
    }
  }
  // The matrix instruction's accumulator rounds with a sign-independent bias (-0.9e-9 of sum |x||w| on every
  // output of this kernel when the four k-steps of a slice accumulated straight into one accumulator: 50-100 x the
  // statistical level, tools/f16_bias_probe.py).  The odd k-steps therefore multiply the NEGATED row fragments into a
  // second accumulator set that is subtracted before the epilogue: their bias has the opposite sign in the result.
  if constexpr (ALT) {
#pragma unroll
    for (int s = 1; s < 4; s += 2)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        q1[2 * s + nb] = neg_x8(q1[2 * s + nb]);
        q2[2 * s + nb] = neg_x8(q2[2 * s + nb]);
        if constexpr (!F16) q3[2 * s + nb] = neg_x8(q3[2 * s + nb]);
      }
  }
  // ADDS: gathered addends Pi[dst], Pj[src] (the edge kernel).  !ADDS: a plain product plus bias, Pi = the bias
  // vector or null (the per-node projections x W_i^T + b and x W_j^T, same kernel with e = x)
  const float* pia = ADDS ? Pi + (long)dsti[rca] * ld_add : Pi;
  const float* pib = ADDS ? Pi + (long)dsti[rcb] * ld_add : Pi;
  const float* pja = ADDS ? Pj + (long)srci[rca] * ld_add : nullptr;
  const float* pjb = ADDS ? Pj + (long)srci[rcb] * ld_add : nullptr;
  float* za = Z + (long)rca * ldz;
  float* zb = Z + (long)rcb * ldz;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#define EZ_TLOAD(gi_)                                                                          \
  {                                                                                            \
    const long gi = (gi_) < last_chunk ? (gi_) : last_chunk;                                   \
    const uint4* tb = Wq + gi * CH16;                                                          \
    const unsigned dst = wave_t + (unsigned)((gi_) & (RING - 1)) * (CH16 * 16);                \
    glds_b128(tb, t_off, dst);                                                                 \
    glds_b128(tb + 256, t_off, dst + 4096);                                                    \
    if (NP == 3) glds_b128(tb + 512, t_off, dst + 8192);                                       \
  }
  EZ_TLOAD(0l);
  EZ_TLOAD(1l);
  EZ_TLOAD(2l);
  if constexpr (DEEP) { EZ_TLOAD(3l); EZ_TLOAD(4l); EZ_TLOAD(5l); EZ_TLOAD(6l); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  bf16x8 fa1, fa2, fa3, fb1, fb2, fb3;
  // group = the three planes of one 16-column block of the chunk in ring slot slot_
#define EZ_READ(F1_, F2_, F3_, slot_, cb_)                                                     \
  {                                                                                            \
    const bf16x8* fp = ring + (slot_) * (CH16) + (cb_) * 64;                                   \
    F1_ = fp[0];                                                                               \
    F2_ = fp[4 * 64];                                                                          \
    if (PASSES >= 6) F3_ = fp[8 * 64];                                                         \
  }
#define EZ_MFMA1(F1_, F2_, F3_, qi_, P_)                                                       \
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
#define EZ_MFMA(F1_, F2_, F3_, s_, cb_)                                                        \
  {                                                                                            \
    if (ALT && ((s_) & 1)) {                                                                   \
      EZ_MFMA1(F1_, F2_, F3_, 2 * (s_) + 0, partn[2 * (cb_) + 0])                              \
      EZ_MFMA1(F1_, F2_, F3_, 2 * (s_) + 1, partn[2 * (cb_) + 1])                              \
    } else {                                                                                   \
      EZ_MFMA1(F1_, F2_, F3_, 2 * (s_) + 0, part[2 * (cb_) + 0])                               \
      EZ_MFMA1(F1_, F2_, F3_, 2 * (s_) + 1, part[2 * (cb_) + 1])                               \
    }                                                                                          \
  }
  EZ_READ(fa1, fa2, fa3, 0, 0);
  f32x4 part[8], partn[ALT ? 8 : 1];
  float dot_a = 0.f, dot_b = 0.f, omx = 0.f;
  const int ncbA = a_out ? H * cb_per_head : 0;      // column blocks that belong to the attention network
  const int cb0 = (int)((long)blockIdx.y * hb.ks0);  // column blocks in front of this column group (whole heads; 0: no groups)
  if (wA) wA += (long)cb0 * 128;
  for (int cb = 0; cb < ncb; ++cb) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int i = 0; i < 8; ++i) part[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (ALT) {
#pragma unroll
        for (int i = 0; i < 8; ++i) partn[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {                  // chunk (cb, half, s) sits in ring slot (4 half + s) % RING
        constexpr int SM = RING - 1;
        const int slot = (half * 4 + s) & SM;
        EZ_TLOAD((long)cb * 8 + half * 4 + s + RING - 1);
#pragma unroll
        for (int cbp = 0; cbp < 2; ++cbp) {
          EZ_READ(fb1, fb2, fb3, slot, 2 * cbp + 1);
          __builtin_amdgcn_sched_barrier(0);
          EZ_MFMA(fa1, fa2, fa3, s, 2 * cbp);
          if (cbp == 0) EZ_READ(fa1, fa2, fa3, slot, 2)
          else EZ_READ(fa1, fa2, fa3, (slot + 1) & SM, 0);
          __builtin_amdgcn_sched_barrier(0);
          EZ_MFMA(fb1, fb2, fb3, s, 2 * cbp + 1);
        }
        if constexpr (DEEP) {
          // chunk i + 2 was issued five iterations ago; younger than it: five chunks (10 loads), the eight stores
          // of the epilogue that preceded this slice and, for s == 0, also those of the slice before
          if (s == 0) wait_vmcnt<26>();
          else wait_vmcnt<18>();
        } else if constexpr (F16) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if constexpr (ALT) {
#pragma unroll
        for (int i = 0; i < 8; ++i) part[i] -= partn[i];
      }
      if constexpr (F16) {                           // undo the row and column-block scales
        float sw, iw;
        pow2_scale(wmax[cb], sw, iw);
        const float ma = rs_a * iw, mb = rs_b * iw;
#pragma unroll
        for (int i = 0; i < 4; ++i) { part[2 * i + 0] = part[2 * i + 0] * ma; part[2 * i + 1] = part[2 * i + 1] * mb; }
      }
      // ---- epilogue of the 64-column slice: z = part + Pi[dst] + Pj[src]; store; logits ----
      const int col0 = cb * 128 + half * 64 + 4 * kg;
      const bool isA = cb + cb0 < ncbA;
      float4 hia[DEEP ? 4 : 1], hja[DEEP ? 4 : 1], hib[DEEP ? 4 : 1], hjb[DEEP ? 4 : 1];
      if constexpr (DEEP) {                          // every gather of the slice before its first store
#pragma unroll
        for (int c16 = 0; c16 < 4; ++c16) {
          hia[c16] = *reinterpret_cast<const float4*>(pia + col0 + 16 * c16);
          hja[c16] = *reinterpret_cast<const float4*>(pja + col0 + 16 * c16);
          hib[c16] = *reinterpret_cast<const float4*>(pib + col0 + 16 * c16);
          hjb[c16] = *reinterpret_cast<const float4*>(pjb + col0 + 16 * c16);
        }
        asm volatile("" ::: "memory");
      }
#pragma unroll
      for (int c16 = 0; c16 < 4; ++c16) {
        const int col = col0 + 16 * c16;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 ia = DEEP ? hia[DEEP ? c16 : 0] : ((ADDS || pia) ? *reinterpret_cast<const float4*>(pia + col) : zero4);
        const float4 ja = DEEP ? hja[DEEP ? c16 : 0] : (ADDS ? *reinterpret_cast<const float4*>(pja + col) : zero4);
        const float4 ib = DEEP ? hib[DEEP ? c16 : 0] : (ADDS ? *reinterpret_cast<const float4*>(pib + col) : ia);
        const float4 jb = DEEP ? hjb[DEEP ? c16 : 0] : (ADDS ? *reinterpret_cast<const float4*>(pjb + col) : zero4);
        const f32x4 pa = part[2 * c16 + 0], pb = part[2 * c16 + 1];
        float4 va = make_float4(pa[0] + ia.x + ja.x, pa[1] + ia.y + ja.y, pa[2] + ia.z + ja.z, pa[3] + ia.w + ja.w);
        float4 vb = make_float4(pb[0] + ib.x + jb.x, pb[1] + ib.y + jb.y, pb[2] + ib.z + jb.z, pb[3] + ib.w + jb.w);
        if (act == CGAT_ACT_LEAKY) {   // store the hidden activations instead of the pre-activations (edge_hidden op)
          va = make_float4(va.x > 0.f ? va.x : 0.01f * va.x, va.y > 0.f ? va.y : 0.01f * va.y,
                           va.z > 0.f ? va.z : 0.01f * va.z, va.w > 0.f ? va.w : 0.01f * va.w);
          vb = make_float4(vb.x > 0.f ? vb.x : 0.01f * vb.x, vb.y > 0.f ? vb.y : 0.01f * vb.y,
                           vb.z > 0.f ? vb.z : 0.01f * vb.z, vb.w > 0.f ? vb.w : 0.01f * vb.w);
        }
        if (!ADDS) {   // the dense-layer form: out = act(x W^T + b) [+ out]
          if (act == CGAT_ACT_TANH) {
            va = make_float4(tanhf(va.x), tanhf(va.y), tanhf(va.z), tanhf(va.w));
            vb = make_float4(tanhf(vb.x), tanhf(vb.y), tanhf(vb.z), tanhf(vb.w));
          }
          if (dact) {   // (kernel argument: uniform) times LeakyReLU'(0.01) at the sign of dact[row, col]: the product IS a
                        // pre-activation gradient (the per-head second layers' input gradient, vector attention)
            const float4 ha = *reinterpret_cast<const float4*>(dact + (long)rca * ld_dact + col);
            const float4 hb = *reinterpret_cast<const float4*>(dact + (long)rcb * ld_dact + col);
            va = make_float4(va.x * (ha.x > 0.f ? 1.f : 0.01f), va.y * (ha.y > 0.f ? 1.f : 0.01f),
                             va.z * (ha.z > 0.f ? 1.f : 0.01f), va.w * (ha.w > 0.f ? 1.f : 0.01f));
            vb = make_float4(vb.x * (hb.x > 0.f ? 1.f : 0.01f), vb.y * (hb.y > 0.f ? 1.f : 0.01f),
                             vb.z * (hb.z > 0.f ? 1.f : 0.01f), vb.w * (hb.w > 0.f ? 1.f : 0.01f));
          }
          if (accumulate) {
            if (row_a < E) { const float4 u = *reinterpret_cast<const float4*>(za + col); va.x += u.x; va.y += u.y; va.z += u.z; va.w += u.w; }
            if (row_b < E) { const float4 u = *reinterpret_cast<const float4*>(zb + col); vb.x += u.x; vb.y += u.y; vb.z += u.z; vb.w += u.w; }
          }
        }
        // DEEP: unconditional (the allowances count eight stores; clamped rows rewrite identical values)
        if constexpr (ZB) {
          if (row_a < E) store4_bf16(reinterpret_cast<__bf16*>(Z) + (long)rca * ldz + col, va);
          if (row_b < E) store4_bf16(reinterpret_cast<__bf16*>(Z) + (long)rcb * ldz + col, vb);
        } else {
          if (DEEP || row_a < E) *reinterpret_cast<float4*>(za + col) = va;
          if (DEEP || row_b < E) *reinterpret_cast<float4*>(zb + col) = vb;
        }
        if (omax) {   // (kernel argument: uniform) max |stored value|: the per-tensor fp16 scale of the kernels that read it
          omx = fmaxf(fmaxf(omx, fmaxf(fabsf(va.x), fabsf(va.y))), fmaxf(fabsf(va.z), fabsf(va.w)));
          omx = fmaxf(fmaxf(omx, fmaxf(fabsf(vb.x), fabsf(vb.y))), fmaxf(fabsf(vb.z), fabsf(vb.w)));
        }
        if (isA) {
          const float4 w = *reinterpret_cast<const float4*>(wA + col);
          dot_a += (va.x > 0.f ? va.x : 0.01f * va.x) * w.x + (va.y > 0.f ? va.y : 0.01f * va.y) * w.y +
                   (va.z > 0.f ? va.z : 0.01f * va.z) * w.z + (va.w > 0.f ? va.w : 0.01f * va.w) * w.w;
          asm volatile("" : "+v"(dot_a));   // keep the two accumulations apart: see the note on packed math above
          dot_b += (vb.x > 0.f ? vb.x : 0.01f * vb.x) * w.x + (vb.y > 0.f ? vb.y : 0.01f * vb.y) * w.y +
                   (vb.z > 0.f ? vb.z : 0.01f * vb.z) * w.z + (vb.w > 0.f ? vb.w : 0.01f * vb.w) * w.w;
          asm volatile("" : "+v"(dot_b));
        }
      }
      if (isA && half == 1 && (cb + 1) % cb_per_head == 0) {   // a head is complete: reduce over the 4 lane groups
        const int h = (cb + cb0) / cb_per_head;
        float da = dot_a, db = dot_b;
        da += __shfl_xor(da, 16, 64); da += __shfl_xor(da, 32, 64);
        db += __shfl_xor(db, 16, 64); db += __shfl_xor(db, 32, 64);
        if (kg == 0) {
          const float bh = bA ? bA[h] : 0.f;
          if (row_a < E) a_out[(long)row_a * H + h] = da + bh;
          if (row_b < E) a_out[(long)row_b * H + h] = db + bh;
        }
        dot_a = 0.f; dot_b = 0.f;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef EZ_TLOAD
#undef EZ_READ
#undef EZ_MFMA1
#undef EZ_MFMA
  if (omax) block_absmax_commit(omx, omax);
}

// ---------------------------------------------------------------------------------------
// f16x3c form of the per-edge launch (round 5):  Z[t, :] = W_e e[perm[t]] + Pi[dst[t]] + Pj[src[t]]  + attention logits, with
// 24-bit operands at 3.75 matrix pass-equivalents per product where edge_z_kernel<6, true> pays six.
// The machinery is bilinear_rows128_ring16c_kernel's (bilinear.hip): 512 threads = 8 waves x 32 rows, the lane's two rows
// scaled per row and split ONCE into two fp16 planes + three 6-bit images (100 VGPRs), the weight as prepare_T_f16c's
// image (a = 128-column block of the output; 25-KB chunks (a, column half, pair of 16-column blocks)) through a 4-slot
// LDS-DMA ring, per chunk 8 groups of fp16 MFMAs + 6 correction instructions.  What differs from the contraction kernel:
//  * every chunk is a finished 32-column slice of the output, so the 64 registers of cross-`a` accumulators are free: they
//    hold the chunk's partial sums in TWO sets (odd k-steps multiply the negated row fragments into the second one, which
//    is subtracted: the matrix instruction's accumulator rounding bias cancels) and the GATHERED
//    addends of the NEXT chunk (8 x 16 bytes per lane), requested a whole chunk of matrix work before their use;
//  * counted waits that never wait for a store: per chunk a wave issues, in this order, 3 (wave 0: 4) LDS-DMA pieces for
//    chunk i + 3, the 8 gathers of chunk i + 1, and -- after the matrix work -- the 4 stores of chunk i.  The gathers of
//    chunk i are needed after the matrix work of chunk i: younger than them are the 4 stores of chunk i - 1 and this
//    iteration's 3 (4) + 8 loads, so the wait is vmcnt(4 + 3 + 8) = 15 (16 in wave 0) -- it also covers the ring (chunk i + 2's pieces are
//    older still) and leaves every store two chunks of matrix work to drain.  Rows beyond E are clamped (they rewrite row
//    E - 1 with identical values) so that every wave issues exactly these operations.
// edge_z_kernel<6, true> alternates two 128-row workgroups per CU and waits vmcnt(3) per k-step: every wait behind an
// epilogue drains that epilogue's stores (in-order vmcnt) and its gathers are requested where they are used: 3.2 ms at the
// BASELINE shape with 1.9 ms of matrix work and 0.66 of wave cycles waiting on memory.
// ---------------------------------------------------------------------------------------
#if !defined(CGAT_DEV_ABLATIONS)   // the product build: the timing-only variants below do not exist, whatever -DEZC_ABL says
#undef EZC_ABL
#define EZC_ABL 0
#elif !defined(EZC_ABL)
#define EZC_ABL 0   // timing-only ablations (wrong results): 1 no Z stores, 2 no gathers, 4 no matrix instructions
#endif
template <bool ZB>
__global__ __launch_bounds__(512, 2) void edge_zc_kernel(const float* __restrict__ e, long lde, const int* __restrict__ perm,
                                                         const uint4* __restrict__ Tq, int ncb,
                                                         const float* __restrict__ Pi, const int* __restrict__ dsti,
                                                         const float* __restrict__ Pj, const int* __restrict__ srci,
                                                         long ld_add, float* __restrict__ Z, long ldz, int E,
                                                         const float* __restrict__ wA, const float* __restrict__ bA,
                                                         int H, int cb_per_head, float* __restrict__ a_out) {
  constexpr int CH16 = F16C_CHUNK16;
  constexpr int SLOTS = 4;
  __shared__ uint4 smem[SLOTS * CH16 + 512 + 8];      // the ring + fc_out_A's weight (<= 2048 floats) + 32 block scales
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int n16 = lane & 15, kg = lane >> 4;
  const int row_w = blockIdx.x * 256 + wave * 32;
  const int row_a = row_w + n16, row_b = row_w + 16 + n16;
  const int rca = row_a < E ? row_a : E - 1, rcb = row_b < E ? row_b : E - 1;
  const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const unsigned wave_t = __builtin_amdgcn_readfirstlane(sbase + wave * 1024);
  const bf16x8* ring = reinterpret_cast<const bf16x8*>(smem) + lane;
  const unsigned char* cring = reinterpret_cast<const unsigned char*>(smem) + 16384;
  const unsigned t_off = (unsigned)tid * 16;
  const long last_chunk = (long)ncb * 4 - 1;
  const float* tmax = reinterpret_cast<const float*>(Tq) + (size_t)ncb * F16C_A_FLOATS;

  // the lane's two edge rows e[row, 32 s + 8 kg + j], scaled per row: fp16 planes q1 / q2 [2 s + nb] (odd k-steps negated)
  // + the three 6-bit images
  bf16x8 q1[8], q2[8];
  frag6 ql6[2], qh6[2], qt6[2];
  float rs_a, rs_b;
  {
    const long ea = perm ? perm[rca] : rca, eb = perm ? perm[rcb] : rcb;
    float qv[2][32];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4* qp = reinterpret_cast<const float4*>(e + (nb ? eb : ea) * lde + 32 * s + 8 * kg);
        const float4 t0 = qp[0], t1 = qp[1];
        qv[nb][8 * s + 0] = t0.x; qv[nb][8 * s + 1] = t0.y; qv[nb][8 * s + 2] = t0.z; qv[nb][8 * s + 3] = t0.w;
        qv[nb][8 * s + 4] = t1.x; qv[nb][8 * s + 5] = t1.y; qv[nb][8 * s + 6] = t1.z; qv[nb][8 * s + 7] = t1.w;
      }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float m = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) m = fmaxf(m, fabsf(qv[nb][j]));
      m = fmaxf(m, __shfl_xor(m, 16));          // the row's 128 values live in the four lanes n16 + 16 kg
      m = fmaxf(m, __shfl_xor(m, 32));
      float sq, iq;
      pow2_scale(m, sq, iq);
      (nb ? rs_b : rs_a) = iq;                  // (the weight block's inverse scale joins it per block `a`)
#pragma unroll
      for (int j = 0; j < 32; ++j) qv[nb][j] *= sq;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = qv[nb][8 * s + j];
        split2_x8_f16(v, q1[2 * s + nb], q2[2 * s + nb]);
        if (s & 1) { q1[2 * s + nb] = neg_x8(q1[2 * s + nb]); q2[2 * s + nb] = neg_x8(q2[2 * s + nb]); }
      }
      f16c_pack32(qv[nb], ql6[nb], qh6[nb], qt6[nb]);   // element 8 s + j <-> k = 32 s + 8 kg + j, as in the image
    }
  }
  // gathered rows as 32-bit byte offsets from Pi / Pj (< 4 GB: checked by the launcher)
  const unsigned oia = (unsigned)(((long)dsti[rca] * ld_add + 4 * kg) * 4), oib = (unsigned)(((long)dsti[rcb] * ld_add + 4 * kg) * 4);
  const unsigned oja = (unsigned)(((long)srci[rca] * ld_add + 4 * kg) * 4), ojb = (unsigned)(((long)srci[rcb] * ld_add + 4 * kg) * 4);
  const int ncbA = (a_out && wA) ? H * cb_per_head : 0;      // column blocks that belong to the attention network
  float* wAs = reinterpret_cast<float*>(smem + SLOTS * CH16);
  for (int i = tid; i < ncbA * 128; i += 512) wAs[i] = wA[i];
  float* its = wAs + 2048;                           // 1 / scale of weight block a (a global load in the loop would make the
  if (tid < ncb && tid < 32) {                       // compiler wait vmcnt(0) there, i.e. for every store in flight)
    float st_, it_;
    pow2_scale(tmax[tid], st_, it_);
    its[tid] = it_;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#define ZC_TLOAD(gi_)                                                                          \
  {                                                                                            \
    const long gi = (gi_) < last_chunk ? (gi_) : last_chunk;                                   \
    const uint4* tb = Tq + gi * CH16;                                                          \
    const unsigned dst = wave_t + (unsigned)((gi_) & 3) * (CH16 * 16);                         \
    glds_b128(tb, t_off, dst);                                                                 \
    glds_b128(tb + 512, t_off, dst + 8192);                                                    \
    glds_b128(tb + 1024, t_off, dst + 16384);                                                  \
    if (wave_u == 0) glds_b128(tb + 1536, t_off, dst + 24576);                                 \
  }
  // the 8 gathered 16-byte pieces of chunk gi_ (clamped): [cb2] x {Pi row a, Pj row a, Pi row b, Pj row b}; issued from inline
  // asm and tied to the counted wait (the compiler cannot see the LDS-DMA and would wait vmcnt(0) before their first use)
#define ZC_GATHER(G_, gi_)                                                                     \
  {                                                                                            \
    const long gi = (gi_) < last_chunk ? (gi_) : last_chunk;                                   \
    const unsigned cbyte = (unsigned)(((gi >> 2) * 128 + ((gi >> 1) & 1) * 64 + (gi & 1) * 32) * 4); \
    if (EZC_ABL & 2) { _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) G_[i_] = f32x4{1.f, 2.f, 3.f, 4.f}; } else \
    _Pragma("unroll") for (int cb2 = 0; cb2 < 2; ++cb2) {                                      \
      asm volatile("global_load_dwordx4 %0, %4, %8\n\tglobal_load_dwordx4 %1, %5, %9\n\t"      \
                   "global_load_dwordx4 %2, %6, %8\n\tglobal_load_dwordx4 %3, %7, %9"          \
                   : "=&v"(G_[4 * cb2 + 0]), "=&v"(G_[4 * cb2 + 1]), "=&v"(G_[4 * cb2 + 2]), "=&v"(G_[4 * cb2 + 3]) \
                   : "v"(oia + cbyte + 64 * cb2), "v"(oja + cbyte + 64 * cb2), "v"(oib + cbyte + 64 * cb2),  \
                     "v"(ojb + cbyte + 64 * cb2), "s"(Pi), "s"(Pj)                             \
                   : "memory");                                                                \
    }                                                                                          \
  }
  f32x4 GA[8], GB[8];
  ZC_TLOAD(0l);
  ZC_TLOAD(1l);
  ZC_TLOAD(2l);
  ZC_GATHER(GA, 0l);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  bf16x8 fa1, fa2, fb1, fb2;
  frag6 ce;
#define ZC_READ(F1_, F2_, slot_, g_)                                                           \
  {                                                                                            \
    const bf16x8* fp = ring + (slot_) * CH16 + ((((g_) >> 1) * 2) * 2 + ((g_) & 1)) * 64;      \
    F1_ = fp[0];                                                                               \
    F2_ = fp[2 * 64];                                                                          \
  }
#define ZC_CREAD(slot_, j_)                                                                    \
  {                                                                                            \
    const unsigned char* cp = cring + (slot_) * (CH16 * 16) + (j_) * 1536;                     \
    const uint4 u_ = *reinterpret_cast<const uint4*>(cp + lane * 16);                          \
    const uint2 w_ = *reinterpret_cast<const uint2*>(cp + 1024 + lane * 8);                    \
    ce.w[0] = u_.x; ce.w[1] = u_.y; ce.w[2] = u_.z; ce.w[3] = u_.w; ce.w[4] = w_.x; ce.w[5] = w_.y; \
  }
  // group g = 2 s + cb2: even k-steps into part, odd ones (negated row fragments) into partn
#define ZC_MFMA(F1_, F2_, g_)                                                                  \
  {                                                                                            \
    _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) {                                         \
      f32x4& P_ = (((g_) >> 1) & 1) ? partn[2 * ((g_) & 1) + nb] : part[2 * ((g_) & 1) + nb];  \
      if (EZC_ABL & 4) { P_[0] += F1_[0] * (float)q1[2 * ((g_) >> 1) + nb][0] + (float)F2_[1] * (float)q2[2 * ((g_) >> 1) + nb][1]; continue; } \
      P_ = mma16<true>(F2_, q1[2 * ((g_) >> 1) + nb], P_);                                     \
      P_ = mma16<true>(F1_, q2[2 * ((g_) >> 1) + nb], P_);                                     \
      P_ = mma16<true>(F1_, q1[2 * ((g_) >> 1) + nb], P_);                                     \
    }                                                                                          \
  }
#define ZC_CORR(j_)                                                                            \
  {                                                                                            \
    _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) {                                         \
      f32x4& P_ = part[2 * ((j_) / 3) + nb];                                                   \
      if ((j_) % 3 == 0) P_ = f16c_mma_th(ce, qh6[nb], P_);                                    \
      else if ((j_) % 3 == 1) P_ = f16c_mma_ht(ce, qt6[nb], P_);                               \
      else P_ = f16c_mma_ll(ce, ql6[nb], P_);                                                  \
    }                                                                                          \
  }
  // epilogue of chunk (a_, ch_): z = (part - partn) * row scale + gathered addends; store; logits
#define ZC_EPILOGUE(G_, a_, ch_)                                                               \
  {                                                                                            \
    asm volatile("" : "+v"(G_[0]), "+v"(G_[1]), "+v"(G_[2]), "+v"(G_[3]), "+v"(G_[4]), "+v"(G_[5]), "+v"(G_[6]), "+v"(G_[7])); \
    int ra_ = rca, rb_ = rcb, lk_ = lane;   /* laundered: the 64-bit row addresses are formed HERE, not kept (= spilled: a   \
                                              scratch reload in this loop waits vmcnt(0), i.e. for every store) */        \
    asm volatile("" : "+v"(ra_), "+v"(rb_), "+v"(lk_));                                        \
    const int col0 = (a_) * 128 + ((ch_) >> 1) * 64 + ((ch_) & 1) * 32 + 4 * (lk_ >> 4);       \
    const float it_ = its[a_];        /* the block's inverse scale */                           \
    const bool isA = (a_) < ncbA;                                                              \
    _Pragma("unroll") for (int cb2 = 0; cb2 < 2; ++cb2) {                                      \
      const int col = col0 + 16 * cb2;                                                         \
      const f32x4 pa = (part[2 * cb2 + 0] - partn[2 * cb2 + 0]) * (rs_a * it_), pb = (part[2 * cb2 + 1] - partn[2 * cb2 + 1]) * (rs_b * it_); \
      const f32x4 ia = G_[4 * cb2 + 0], ja = G_[4 * cb2 + 1], ib = G_[4 * cb2 + 2], jb = G_[4 * cb2 + 3]; \
      const float4 va = make_float4(pa[0] + ia[0] + ja[0], pa[1] + ia[1] + ja[1], pa[2] + ia[2] + ja[2], pa[3] + ia[3] + ja[3]); \
      const float4 vb = make_float4(pb[0] + ib[0] + jb[0], pb[1] + ib[1] + jb[1], pb[2] + ib[2] + jb[2], pb[3] + ib[3] + jb[3]); \
      if (EZC_ABL & 1) {                                                                       \
        if (va.x == 123.456f) Z[0] = vb.y;                                                     \
      } else if constexpr (ZB) {                                                               \
        store4_bf16(reinterpret_cast<__bf16*>(Z) + (long)ra_ * ldz + col, va);                 \
        store4_bf16(reinterpret_cast<__bf16*>(Z) + (long)rb_ * ldz + col, vb);                 \
      } else {                                                                                 \
        *reinterpret_cast<float4*>(Z + (long)ra_ * ldz + col) = va;                            \
        *reinterpret_cast<float4*>(Z + (long)rb_ * ldz + col) = vb;                            \
      }                                                                                        \
      if (isA) {                                                                               \
        const float4 w = *reinterpret_cast<const float4*>(wAs + col);                          \
        dot_a += (va.x > 0.f ? va.x : 0.01f * va.x) * w.x + (va.y > 0.f ? va.y : 0.01f * va.y) * w.y + \
                 (va.z > 0.f ? va.z : 0.01f * va.z) * w.z + (va.w > 0.f ? va.w : 0.01f * va.w) * w.w; \
        asm volatile("" : "+v"(dot_a));   /* keep the two accumulations apart: see the note on packed math above */ \
        dot_b += (vb.x > 0.f ? vb.x : 0.01f * vb.x) * w.x + (vb.y > 0.f ? vb.y : 0.01f * vb.y) * w.y + \
                 (vb.z > 0.f ? vb.z : 0.01f * vb.z) * w.z + (vb.w > 0.f ? vb.w : 0.01f * vb.w) * w.w; \
        asm volatile("" : "+v"(dot_b));                                                        \
      }                                                                                        \
    }                                                                                          \
    if (isA && (ch_) == 3 && ((a_) + 1) % cb_per_head == 0) {   /* a head is complete: reduce over the 4 lane groups */ \
      const int h = (a_) / cb_per_head;                                                        \
      float da = dot_a, db = dot_b;                                                            \
      da += __shfl_xor(da, 16, 64); da += __shfl_xor(da, 32, 64);                              \
      db += __shfl_xor(db, 16, 64); db += __shfl_xor(db, 32, 64);                              \
      if ((lk_ >> 4) == 0) {   /* (clamped rows rewrite row E - 1's logits with identical values) */ \
        const float bh = bA ? bA[h] : 0.f;                                                     \
        a_out[(long)ra_ * H + h] = da + bh;                                                    \
        a_out[(long)rb_ * H + h] = db + bh;                                                    \
      }                                                                                        \
      dot_a = 0.f; dot_b = 0.f;                                                                \
    }                                                                                          \
  }
  // one chunk: ring prefetch, the next chunk's gathers, 8 groups + 6 corrections (every fragment read one group ahead), the
  // counted wait, the epilogue, the barrier
#define ZC_CHUNK(a_, ch_, GCUR_, GNEXT_)                                                       \
  {                                                                                            \
    constexpr int sl_ = (ch_), sn_ = ((ch_) + 1) & 3;   /* ring slot = chunk index mod 4 = ch_ */ \
    ZC_TLOAD((long)(a_) * 4 + (ch_) + 3);                                                      \
    ZC_GATHER(GNEXT_, (long)(a_) * 4 + (ch_) + 1);                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { part[i] = f32x4{0.f, 0.f, 0.f, 0.f}; partn[i] = f32x4{0.f, 0.f, 0.f, 0.f}; } \
    _Pragma("unroll") for (int gp = 0; gp < 4; ++gp) {                                         \
      ZC_READ(fb1, fb2, sl_, 2 * gp + 1);                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                       \
      if (2 * gp < 6) { ZC_CORR(2 * gp); ZC_CREAD(sl_, 2 * gp + 1); }                          \
      ZC_MFMA(fa1, fa2, 2 * gp);                                                               \
      if (gp < 3) ZC_READ(fa1, fa2, sl_, 2 * gp + 2)                                           \
      else ZC_READ(fa1, fa2, sn_, 0);                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                       \
      if (2 * gp + 1 < 6) {                                                                    \
        ZC_CORR(2 * gp + 1);                                                                   \
        if (2 * gp + 2 < 6) ZC_CREAD(sl_, 2 * gp + 2)                                          \
        else ZC_CREAD(sn_, 0);                                                                 \
      }                                                                                        \
      ZC_MFMA(fb1, fb2, 2 * gp + 1);                                                           \
    }                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    /* gathers of THIS chunk (issued one iteration ago): younger are the 4 stores of the previous chunk and this       \
       iteration's 3 (4) pieces + 8 gathers; the ring needs nothing more (chunk i + 2's pieces are older still) */        \
    if (EZC_ABL & 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            \
    else if (wave_u == 0) wait_vmcnt<16>(); else wait_vmcnt<15>();                             \
    ZC_EPILOGUE(GCUR_, a_, ch_)                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    __builtin_amdgcn_s_barrier();                                                              \
    asm volatile("" ::: "memory");                                                             \
  }
  ZC_READ(fa1, fa2, 0, 0);
  ZC_CREAD(0, 0);
  f32x4 part[4], partn[4];
  float dot_a = 0.f, dot_b = 0.f;
  for (int a = 0; a < ncb; ++a) {
    ZC_CHUNK(a, 0, GA, GB)
    ZC_CHUNK(a, 1, GB, GA)
    ZC_CHUNK(a, 2, GA, GB)
    ZC_CHUNK(a, 3, GB, GA)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef ZC_TLOAD
#undef ZC_GATHER
#undef ZC_READ
#undef ZC_CREAD
#undef ZC_MFMA
#undef ZC_CORR
#undef ZC_EPILOGUE
#undef ZC_CHUNK
}

// ---------------------------------------------------------------------------------------
// The six-pass per-edge launch on edge_zc_kernel's skeleton (round 5): the SAME arithmetic as edge_z_kernel<6, true>
// -- three bf16 planes per operand, the six products in the same order, even / odd k-steps into the two accumulator sets,
// the same epilogue order: bit-identical Z and logits -- with what made the f16x3c form faster although the matrix work was
// never its bound: 256-row workgroups (half the weight traffic through the ring per row), the gathered addends of the
// NEXT 32-column chunk requested a whole chunk of matrix work before their use into registers the 32-column
// granularity frees, and counted waits that never wait for a store (derivation at edge_zc_kernel).
// Operand: prepare_T_bf16's image as it is (12-KB k-step chunks (a, half, s) = [plane][cb][lane] x 16 B); a ring chunk
// here is (a, half, pair of 16-column blocks) = 24 one-KB pieces [s][plane][cb2] picked out of four of those, three per
// wave.  LDS: 4 slots x 24 KB + fc_out_A's weight.
// ---------------------------------------------------------------------------------------
template <bool ZB>
__global__ __launch_bounds__(512, 2) void edge_z6w_kernel(const float* __restrict__ e, long lde, const int* __restrict__ perm,
                                                          const uint4* __restrict__ Wq, int ncb,
                                                          const float* __restrict__ Pi, const int* __restrict__ dsti,
                                                          const float* __restrict__ Pj, const int* __restrict__ srci,
                                                          long ld_add, float* __restrict__ Z, long ldz, int E,
                                                          const float* __restrict__ wA, const float* __restrict__ bA,
                                                          int H, int cb_per_head, float* __restrict__ a_out, int act,
                                                          float* __restrict__ omax) {
  constexpr int CH = 24 * 64;                         // 16-byte pieces per ring chunk (24 KB)
  constexpr int SLOTS = 4;
  __shared__ uint4 smem[SLOTS * CH + 512];            // the ring + fc_out_A's weight (<= 2048 floats)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int n16 = lane & 15, kg = lane >> 4;
  const int row_w = blockIdx.x * 256 + wave * 32;
  const int row_a = row_w + n16, row_b = row_w + 16 + n16;
  const int rca = row_a < E ? row_a : E - 1, rcb = row_b < E ? row_b : E - 1;
  const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const bf16x8* ring = reinterpret_cast<const bf16x8*>(smem) + lane;
  const unsigned l_off = (unsigned)lane * 16;
  const long last_chunk = (long)ncb * 4 - 1;
  // this wave's three pieces pc = 3 wave + i of a chunk: (s, plane, cb2) = (pc / 6, (pc % 6) / 2, pc % 2); source offset
  // inside the (a, half) group of four k-step chunks in 16-byte units, destination offset inside the slot in bytes
  int psrc[3];
  unsigned pdst[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int pc = 3 * wave_u + i, s4 = pc / 6, pl = (pc % 6) >> 1, c2 = pc & 1;
    psrc[i] = s4 * 768 + pl * 256 + c2 * 64;
    pdst[i] = __builtin_amdgcn_readfirstlane(sbase + (unsigned)pc * 1024);
  }

  // the lane's two edge rows, split once: q[plane][2 s + nb] holds e[row(nb), 32 s + 8 kg + 0..7]; odd k-steps negated
  bf16x8 q1[8], q2[8], q3[8];
  {
    const long ea = perm ? perm[rca] : rca, eb = perm ? perm[rcb] : rcb;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const float4* qp = reinterpret_cast<const float4*>(e + (nb ? eb : ea) * lde + 32 * s4 + 8 * kg);
        const float4 t0 = qp[0], t1 = qp[1];
        const float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        split3_x8(v, q1[2 * s4 + nb], q2[2 * s4 + nb], q3[2 * s4 + nb]);
        if (s4 & 1) {
          q1[2 * s4 + nb] = neg_x8(q1[2 * s4 + nb]);
          q2[2 * s4 + nb] = neg_x8(q2[2 * s4 + nb]);
          q3[2 * s4 + nb] = neg_x8(q3[2 * s4 + nb]);
        }
      }
  }
  // gathered rows as 32-bit byte offsets from Pi / Pj (< 4 GB: checked by the launcher)
  const unsigned oia = (unsigned)(((long)dsti[rca] * ld_add + 4 * kg) * 4), oib = (unsigned)(((long)dsti[rcb] * ld_add + 4 * kg) * 4);
  const unsigned oja = (unsigned)(((long)srci[rca] * ld_add + 4 * kg) * 4), ojb = (unsigned)(((long)srci[rcb] * ld_add + 4 * kg) * 4);
  const int ncbA = (a_out && wA) ? H * cb_per_head : 0;      // column blocks that belong to the attention network
  float* wAs = reinterpret_cast<float*>(smem + SLOTS * CH);
  for (int i = tid; i < ncbA * 128; i += 512) wAs[i] = wA[i];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // chunk gi = 4 a + ch, ch = 2 half + pair: its 24 pieces live at Wq + ((2 a + half) * 4) * 768 + pair * 128 + psrc
#define Z6_TLOAD(gi_)                                                                          \
  {                                                                                            \
    const long gi = (gi_) < last_chunk ? (gi_) : last_chunk;                                   \
    const uint4* tb = Wq + (gi >> 1) * (4 * 768) + (gi & 1) * 128;                             \
    const unsigned so = (unsigned)((gi_) & 3) * (CH * 16);                                     \
    glds_b128(tb + psrc[0], l_off, pdst[0] + so);                                              \
    glds_b128(tb + psrc[1], l_off, pdst[1] + so);                                              \
    glds_b128(tb + psrc[2], l_off, pdst[2] + so);                                              \
  }
#define Z6_GATHER(G_, gi_)                                                                     \
  {                                                                                            \
    const long gi = (gi_) < last_chunk ? (gi_) : last_chunk;                                   \
    const unsigned cbyte = (unsigned)(((gi >> 2) * 128 + ((gi >> 1) & 1) * 64 + (gi & 1) * 32) * 4); \
    _Pragma("unroll") for (int cb2 = 0; cb2 < 2; ++cb2) {                                      \
      asm volatile("global_load_dwordx4 %0, %4, %8\n\tglobal_load_dwordx4 %1, %5, %9\n\t"      \
                   "global_load_dwordx4 %2, %6, %8\n\tglobal_load_dwordx4 %3, %7, %9"          \
                   : "=&v"(G_[4 * cb2 + 0]), "=&v"(G_[4 * cb2 + 1]), "=&v"(G_[4 * cb2 + 2]), "=&v"(G_[4 * cb2 + 3]) \
                   : "v"(oia + cbyte + 64 * cb2), "v"(oja + cbyte + 64 * cb2), "v"(oib + cbyte + 64 * cb2),  \
                     "v"(ojb + cbyte + 64 * cb2), "s"(Pi), "s"(Pj)                             \
                   : "memory");                                                                \
    }                                                                                          \
  }
  f32x4 GA[8], GB[8];
  Z6_TLOAD(0l);
  Z6_TLOAD(1l);
  Z6_TLOAD(2l);
  Z6_GATHER(GA, 0l);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  bf16x8 fa1, fa2, fa3, fb1, fb2, fb3;
  // group g = 2 s + cb2: the three planes of 16-column block cb2 at k-step s
#define Z6_READ(F1_, F2_, F3_, slot_, g_)                                                      \
  {                                                                                            \
    const bf16x8* fp = ring + (slot_) * CH + (((g_) >> 1) * 6 + ((g_) & 1)) * 64;              \
    F1_ = fp[0];                                                                               \
    F2_ = fp[2 * 64];                                                                          \
    F3_ = fp[4 * 64];                                                                          \
  }
  // even k-steps into part, odd ones (negated row fragments) into partn; the six products in edge_z_kernel's order
#define Z6_MFMA(F1_, F2_, F3_, g_)                                                             \
  {                                                                                            \
    _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) {                                         \
      f32x4& P_ = (((g_) >> 1) & 1) ? partn[2 * ((g_) & 1) + nb] : part[2 * ((g_) & 1) + nb];  \
      const int qi_ = 2 * ((g_) >> 1) + nb;                                                    \
      P_ = mma16<false>(F3_, q1[qi_], P_);                                                     \
      P_ = mma16<false>(F1_, q3[qi_], P_);                                                     \
      P_ = mma16<false>(F2_, q2[qi_], P_);                                                     \
      P_ = mma16<false>(F2_, q1[qi_], P_);                                                     \
      P_ = mma16<false>(F1_, q2[qi_], P_);                                                     \
      P_ = mma16<false>(F1_, q1[qi_], P_);                                                     \
    }                                                                                          \
  }
  // epilogue of chunk (a_, ch_): z = (part - partn) + gathered addends; store; logits
#define Z6_EPILOGUE(G_, a_, ch_)                                                               \
  {                                                                                            \
    asm volatile("" : "+v"(G_[0]), "+v"(G_[1]), "+v"(G_[2]), "+v"(G_[3]), "+v"(G_[4]), "+v"(G_[5]), "+v"(G_[6]), "+v"(G_[7])); \
    int ra_ = rca, rb_ = rcb, lk_ = lane;   /* laundered: the 64-bit row addresses are formed HERE, not kept (a scratch \
                                              reload in this loop waits vmcnt(0), i.e. for every store) */                \
    asm volatile("" : "+v"(ra_), "+v"(rb_), "+v"(lk_));                                        \
    const int col0 = (a_) * 128 + ((ch_) >> 1) * 64 + ((ch_) & 1) * 32 + 4 * (lk_ >> 4);       \
    const bool isA = (a_) < ncbA;                                                              \
    _Pragma("unroll") for (int cb2 = 0; cb2 < 2; ++cb2) {                                      \
      const int col = col0 + 16 * cb2;                                                         \
      const f32x4 pa = part[2 * cb2 + 0] - partn[2 * cb2 + 0], pb = part[2 * cb2 + 1] - partn[2 * cb2 + 1]; \
      const f32x4 ia = G_[4 * cb2 + 0], ja = G_[4 * cb2 + 1], ib = G_[4 * cb2 + 2], jb = G_[4 * cb2 + 3]; \
      float4 va = make_float4(pa[0] + ia[0] + ja[0], pa[1] + ia[1] + ja[1], pa[2] + ia[2] + ja[2], pa[3] + ia[3] + ja[3]); \
      float4 vb = make_float4(pb[0] + ib[0] + jb[0], pb[1] + ib[1] + jb[1], pb[2] + ib[2] + jb[2], pb[3] + ib[3] + jb[3]); \
      if (act == CGAT_ACT_LEAKY) {   /* (kernel argument: uniform) the hidden activations instead of the pre-activations */ \
        va = make_float4(va.x > 0.f ? va.x : 0.01f * va.x, va.y > 0.f ? va.y : 0.01f * va.y,   \
                         va.z > 0.f ? va.z : 0.01f * va.z, va.w > 0.f ? va.w : 0.01f * va.w);  \
        vb = make_float4(vb.x > 0.f ? vb.x : 0.01f * vb.x, vb.y > 0.f ? vb.y : 0.01f * vb.y,   \
                         vb.z > 0.f ? vb.z : 0.01f * vb.z, vb.w > 0.f ? vb.w : 0.01f * vb.w);  \
      }                                                                                        \
      if constexpr (ZB) {                                                                      \
        store4_bf16(reinterpret_cast<__bf16*>(Z) + (long)ra_ * ldz + col, va);                 \
        store4_bf16(reinterpret_cast<__bf16*>(Z) + (long)rb_ * ldz + col, vb);                 \
      } else {                                                                                 \
        *reinterpret_cast<float4*>(Z + (long)ra_ * ldz + col) = va;                            \
        *reinterpret_cast<float4*>(Z + (long)rb_ * ldz + col) = vb;                            \
      }                                                                                        \
      if (omax) {   /* (kernel argument: uniform) max |stored value|; dot_b is free in a launch without logits */ \
        dot_b = fmaxf(fmaxf(dot_b, fmaxf(fabsf(va.x), fabsf(va.y))), fmaxf(fabsf(va.z), fabsf(va.w))); \
        dot_b = fmaxf(fmaxf(dot_b, fmaxf(fabsf(vb.x), fabsf(vb.y))), fmaxf(fabsf(vb.z), fabsf(vb.w))); \
      }                                                                                        \
      if (isA) {                                                                               \
        const float4 w = *reinterpret_cast<const float4*>(wAs + col);                          \
        dot_a += (va.x > 0.f ? va.x : 0.01f * va.x) * w.x + (va.y > 0.f ? va.y : 0.01f * va.y) * w.y + \
                 (va.z > 0.f ? va.z : 0.01f * va.z) * w.z + (va.w > 0.f ? va.w : 0.01f * va.w) * w.w; \
        asm volatile("" : "+v"(dot_a));   /* keep the two accumulations apart: see the note on packed math above */ \
        dot_b += (vb.x > 0.f ? vb.x : 0.01f * vb.x) * w.x + (vb.y > 0.f ? vb.y : 0.01f * vb.y) * w.y + \
                 (vb.z > 0.f ? vb.z : 0.01f * vb.z) * w.z + (vb.w > 0.f ? vb.w : 0.01f * vb.w) * w.w; \
        asm volatile("" : "+v"(dot_b));                                                        \
      }                                                                                        \
    }                                                                                          \
    if (isA && (ch_) == 3 && ((a_) + 1) % cb_per_head == 0) {   /* a head is complete: reduce over the 4 lane groups */ \
      const int h = (a_) / cb_per_head;                                                        \
      float da = dot_a, db = dot_b;                                                            \
      da += __shfl_xor(da, 16, 64); da += __shfl_xor(da, 32, 64);                              \
      db += __shfl_xor(db, 16, 64); db += __shfl_xor(db, 32, 64);                              \
      if ((lk_ >> 4) == 0) {   /* (clamped rows rewrite row E - 1's logits with identical values) */ \
        const float bh = bA ? bA[h] : 0.f;                                                     \
        a_out[(long)ra_ * H + h] = da + bh;                                                    \
        a_out[(long)rb_ * H + h] = db + bh;                                                    \
      }                                                                                        \
      dot_a = 0.f; dot_b = 0.f;                                                                \
    }                                                                                          \
  }
  // one chunk: ring prefetch, the next chunk's gathers, 8 groups of 12 matrix instructions (every fragment read one
  // group ahead), the counted wait, the epilogue, the barrier
#define Z6_CHUNK(a_, ch_, GCUR_, GNEXT_)                                                       \
  {                                                                                            \
    constexpr int sl_ = (ch_), sn_ = ((ch_) + 1) & 3;   /* ring slot = chunk index mod 4 = ch_ */ \
    Z6_TLOAD((long)(a_) * 4 + (ch_) + 3);                                                      \
    Z6_GATHER(GNEXT_, (long)(a_) * 4 + (ch_) + 1);                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { part[i] = f32x4{0.f, 0.f, 0.f, 0.f}; partn[i] = f32x4{0.f, 0.f, 0.f, 0.f}; } \
    _Pragma("unroll") for (int gp = 0; gp < 4; ++gp) {                                         \
      Z6_READ(fb1, fb2, fb3, sl_, 2 * gp + 1);                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                       \
      Z6_MFMA(fa1, fa2, fa3, 2 * gp);                                                          \
      if (gp < 3) Z6_READ(fa1, fa2, fa3, sl_, 2 * gp + 2)                                      \
      else Z6_READ(fa1, fa2, fa3, sn_, 0);                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                       \
      Z6_MFMA(fb1, fb2, fb3, 2 * gp + 1);                                                      \
    }                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    /* gathers of THIS chunk (issued one iteration ago): younger are the 4 stores of the previous chunk and this       \
       iteration's 3 pieces + 8 gathers; the ring needs nothing more (chunk i + 2's pieces are older still) */          \
    wait_vmcnt<15>();                                                                          \
    Z6_EPILOGUE(GCUR_, a_, ch_)                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    __builtin_amdgcn_s_barrier();                                                              \
    asm volatile("" ::: "memory");                                                             \
  }
  Z6_READ(fa1, fa2, fa3, 0, 0);
  f32x4 part[4], partn[4];
  float dot_a = 0.f, dot_b = 0.f;
  for (int a = 0; a < ncb; ++a) {
    Z6_CHUNK(a, 0, GA, GB)
    Z6_CHUNK(a, 1, GB, GA)
    Z6_CHUNK(a, 2, GA, GB)
    Z6_CHUNK(a, 3, GB, GA)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (omax) block_absmax_commit(dot_b, omax);
#undef Z6_TLOAD
#undef Z6_GATHER
#undef Z6_READ
#undef Z6_MFMA
#undef Z6_EPILOGUE
#undef Z6_CHUNK
}

// ---------------------------------------------------------------------------------------
// The per-edge launch with the x_j projection folded in (f16x3 arithmetic):
//     Z[t, :] = [W_e | W_j] [e[perm[t]] ; x[src[t]]] + Pi[dst[t], :]
// Why, and what bounds these kernels: timing-only ablations (tools/edgez_ablation.sh; E = 1 000 080)
//                                   edge_z (K = 128, Pi and Pj gathered)     this kernel (K = 256, Pi gathered)
//     as is                                      3.15 ms                     3.2 -> 2.76 ms (see below)
//     without the Z stores                       (-17 %)                     2.1 -> 1.92
//     without the epilogue's loads               (-55 %)                     2.7 -> 2.27
//     MFMA + LDS-ring skeleton only              ~0.95                       1.7
// Neither form hid its epilogue behind the other workgroup of the CU; pointing the old kernel's Pj gather at cached
// rows (3.0 ms) or sharing the weight stream between 256 rows (3.6 ms) changed nothing.  The reason is the in-order
// vmcnt: stores count in it, so the first ring wait after an epilogue that needs a load issued AFTER the stores
// drains them, and inside the epilogue every load issued after a store does the same.  Two changes took this kernel
// from 3.2 to 2.76 ms: an 8-slot ring (the first load younger than the stores is needed five k-steps later instead of
// two; waits of vmcnt(18)/(10), derivation at the wait) and all Pi loads of a slice issued before its first store.
// Folding the x_j projection into the product (K = 256) is what makes room for that: it removes the Pj gather (6 KB
// per edge, a third of the epilogue's requests, 16 VGPRs of addresses) at the price of matrix work the kernel has
// room for, and the per-node projection Pj = x W_j^T is no longer computed.  Still exposed: ~0.85 ms of store
// latency (five k-steps = 1.3 us is not enough under load) and the Pi loads' L2 latency (0.2 ms); a ring on a
// producer wave, whose vmcnt never sees a store, is the next step.
// Operand: prepare_W2_f16 planes, chunk (a, half, s) = 8 KB at Wq + ((a*2 + half)*8 + s) * 512 uint4, s < 4 the W_e
// k-steps, s >= 4 the W_j ones, one scale per column block a over both (wmax behind the planes); the row scale is the
// maximum over the concatenated 256-value row.
// ZB: Z is stored as bf16 (the "bf16" edge-storage mode: half the bytes of the store that bounds this kernel; the
// logits are still taken from the fp32 values in registers)
template <int ABL, bool ZB = false>   // ABL: timing-only ablations (wrong results): 1 no Z stores, 2 no Pi loads, 4 no logits
__global__ __launch_bounds__(256, 2) void edge_zx_kernel(const float* __restrict__ e, long lde,
                                                         const int* __restrict__ perm, const float* __restrict__ xn,
                                                         long ldx, const uint4* __restrict__ Wq, int ncb,
                                                         const float* __restrict__ Pi, const int* __restrict__ dsti,
                                                         const int* __restrict__ srci, long ld_add,
                                                         float* __restrict__ Z, long ldz, int E,
                                                         const float* __restrict__ wA, const float* __restrict__ bA,
                                                         int H, int cb_per_head, float* __restrict__ a_out) {
  constexpr int CH16 = 2 * 4 * 64;              // 16-byte pieces per chunk = 8 KB
  constexpr int RING = 8;                       // ring slots (see the vmcnt note at the loop's wait)
  __shared__ uint4 smem[RING * CH16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kg = lane >> 4;
  const int row_w = blockIdx.x * 128 + wave * 32;
  const int row_a = row_w + n16, row_b = row_a + 16;
  const int rca = row_a < E ? row_a : E - 1, rcb = row_b < E ? row_b : E - 1;
  const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const unsigned wave_t = __builtin_amdgcn_readfirstlane(sbase + wave * 1024);
  const bf16x8* ring = reinterpret_cast<const bf16x8*>(smem) + lane;
  const unsigned t_off = (unsigned)tid * 16;
  const long last_chunk = (long)ncb * 16 - 1;
  const float* wmax = reinterpret_cast<const float*>(Wq + (long)ncb * 16 * CH16);

  // q[plane][2 s + nb]: s < 4 from e[perm[row]], s >= 4 from x[src[row]]
  bf16x8 q1[16], q2[16];
  float rs_a, rs_b;
  {
    const float* ra[2] = {e + (perm ? (long)perm[rca] : (long)rca) * lde, e + (perm ? (long)perm[rcb] : (long)rcb) * lde};
    const float* rx[2] = {xn + (long)srci[rca] * ldx, xn + (long)srci[rcb] * ldx};
    float qv[2][64];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const float4* qp = reinterpret_cast<const float4*>((s < 4 ? ra[nb] : rx[nb]) + 32 * (s & 3) + 8 * kg);
        const float4 t0 = qp[0], t1 = qp[1];
        qv[nb][8 * s + 0] = t0.x; qv[nb][8 * s + 1] = t0.y; qv[nb][8 * s + 2] = t0.z; qv[nb][8 * s + 3] = t0.w;
        qv[nb][8 * s + 4] = t1.x; qv[nb][8 * s + 5] = t1.y; qv[nb][8 * s + 6] = t1.z; qv[nb][8 * s + 7] = t1.w;
      }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float m = 0.f;
#pragma unroll
      for (int j = 0; j < 64; ++j) m = fmaxf(m, fabsf(qv[nb][j]));
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      float sq, iq;
      pow2_scale(m, sq, iq);
      if (nb) rs_b = iq; else rs_a = iq;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = qv[nb][8 * s + j] * sq;
        split2_x8_f16(v, q1[2 * s + nb], q2[2 * s + nb]);
      }
    }
  }
  const float* pia = Pi + (long)dsti[rca] * ld_add;
  const float* pib = Pi + (long)dsti[rcb] * ld_add;
  float* za = Z + (long)rca * ldz;
  float* zb = Z + (long)rcb * ldz;
  __bf16* za16 = reinterpret_cast<__bf16*>(Z) + (long)rca * ldz;
  __bf16* zb16 = reinterpret_cast<__bf16*>(Z) + (long)rcb * ldz;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#define EX_TLOAD(gi_)                                                                          \
  {                                                                                            \
    const long gi = (gi_) < last_chunk ? (gi_) : last_chunk;                                   \
    const uint4* tb = Wq + gi * CH16;                                                          \
    const unsigned dst = wave_t + (unsigned)((gi_) & (RING - 1)) * (CH16 * 16);                \
    glds_b128(tb, t_off, dst);                                                                 \
    glds_b128(tb + 256, t_off, dst + 4096);                                                    \
  }
  EX_TLOAD(0l);
  EX_TLOAD(1l);
  EX_TLOAD(2l);
  EX_TLOAD(3l);
  EX_TLOAD(4l);
  EX_TLOAD(5l);
  EX_TLOAD(6l);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  bf16x8 fa1, fa2, fb1, fb2;
#define EX_READ(F1_, F2_, slot_, cb_)                                                          \
  {                                                                                            \
    const bf16x8* fp = ring + (slot_) * (CH16) + (cb_) * 64;                                   \
    F1_ = fp[0];                                                                               \
    F2_ = fp[4 * 64];                                                                          \
  }
#define EX_MFMA1(F1_, F2_, qi_, P_)                                                            \
  {                                                                                            \
    P_ = mma16<true>(F2_, q1[qi_], P_);                                                        \
    P_ = mma16<true>(F1_, q2[qi_], P_);                                                        \
    P_ = mma16<true>(F1_, q1[qi_], P_);                                                        \
  }
#define EX_MFMA(F1_, F2_, s_, cb_)                                                             \
  {                                                                                            \
    EX_MFMA1(F1_, F2_, 2 * (s_) + 0, part[2 * (cb_) + 0])                                      \
    EX_MFMA1(F1_, F2_, 2 * (s_) + 1, part[2 * (cb_) + 1])                                      \
  }
  EX_READ(fa1, fa2, 0, 0);
  f32x4 part[8];
  float dot_a = 0.f, dot_b = 0.f;
  const int ncbA = a_out ? H * cb_per_head : 0;
  for (int cb = 0; cb < ncb; ++cb) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int i = 0; i < 8; ++i) part[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s) {                  // chunk (cb, half, s) sits in ring slot s
        EX_TLOAD((long)cb * 16 + half * 8 + s + 7);
#pragma unroll
        for (int cbp = 0; cbp < 2; ++cbp) {
          EX_READ(fb1, fb2, s, 2 * cbp + 1);
          __builtin_amdgcn_sched_barrier(0);
          EX_MFMA(fa1, fa2, s, 2 * cbp);
          if (cbp == 0) EX_READ(fa1, fa2, s, 2)
          else EX_READ(fa1, fa2, (s + 1) & 7, 0);
          __builtin_amdgcn_sched_barrier(0);
          EX_MFMA(fb1, fb2, s, 2 * cbp + 1);
        }
        // At the end of k-step s chunk s + 2 must have landed (its fragments are read ahead during s + 1); it was
        // issued five k-steps ago.  vmcnt retires IN ORDER and counts stores: everything younger than that chunk may
        // stay in flight -- the five later chunks (10 loads) and, for s < 5, the eight Z stores of the previous
        // slice's epilogue, which were issued after it.  With a 4-slot ring the first wait after an epilogue already
        // had to drain the stores (a load issued after them was needed two k-steps later): the epilogue's write
        // latency was paid in full on every slice.
        if (s < 5) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      {
        float sw, iw;
        pow2_scale(wmax[cb], sw, iw);
        const float ma = rs_a * iw, mb = rs_b * iw;
#pragma unroll
        for (int i = 0; i < 4; ++i) { part[2 * i + 0] = part[2 * i + 0] * ma; part[2 * i + 1] = part[2 * i + 1] * mb; }
      }
      // ---- epilogue of the 64-column slice: z = part + Pi[dst]; store; logits ----
      const int col0 = cb * 128 + half * 64 + 4 * kg;
      const bool isA = cb < ncbA;
      // all Pi loads of the slice before its first store: a load issued after a store cannot be waited for without
      // draining that store (in-order vmcnt)
      float4 pia4[4], pib4[4];
#pragma unroll
      for (int c16 = 0; c16 < 4; ++c16) {
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        pia4[c16] = (ABL & 2) ? zero4 : *reinterpret_cast<const float4*>(pia + col0 + 16 * c16);
        pib4[c16] = (ABL & 2) ? zero4 : *reinterpret_cast<const float4*>(pib + col0 + 16 * c16);
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int c16 = 0; c16 < 4; ++c16) {
        const int col = col0 + 16 * c16;
        const float4 ia = pia4[c16], ib = pib4[c16];
        const f32x4 pa = part[2 * c16 + 0], pb = part[2 * c16 + 1];
        const float4 va = make_float4(pa[0] + ia.x, pa[1] + ia.y, pa[2] + ia.z, pa[3] + ia.w);
        const float4 vb = make_float4(pb[0] + ib.x, pb[1] + ib.y, pb[2] + ib.z, pb[3] + ib.w);
        // (non-temporal stores of Z were tried at the end of round 2: 2.77 -> 3.27 ms -- the 64-byte pieces of a row that
        // four consecutive store instructions write are merged into full lines by L2 only when they allocate there)
        // no `row < E` guard: the ring's vmcnt allowances count exactly eight stores per epilogue, and a wave whose
        // second row block lies past the end would skip four of them (lanes past the end hold the clamped row E - 1
        // and rewrite it with identical values)
        if constexpr (ZB) {
          store4_bf16(za16 + col, va);
          store4_bf16(zb16 + col, vb);
        } else {
          if (!(ABL & 1) || va.x == 1234.5f) *reinterpret_cast<float4*>(za + col) = va;
          if (!(ABL & 1) || vb.x == 1234.5f) *reinterpret_cast<float4*>(zb + col) = vb;
        }
        if (isA && !(ABL & 4)) {
          const float4 w = *reinterpret_cast<const float4*>(wA + col);
          dot_a += (va.x > 0.f ? va.x : 0.01f * va.x) * w.x + (va.y > 0.f ? va.y : 0.01f * va.y) * w.y +
                   (va.z > 0.f ? va.z : 0.01f * va.z) * w.z + (va.w > 0.f ? va.w : 0.01f * va.w) * w.w;
          asm volatile("" : "+v"(dot_a));   // keep the two accumulations apart: see the note on packed math above
          dot_b += (vb.x > 0.f ? vb.x : 0.01f * vb.x) * w.x + (vb.y > 0.f ? vb.y : 0.01f * vb.y) * w.y +
                   (vb.z > 0.f ? vb.z : 0.01f * vb.z) * w.z + (vb.w > 0.f ? vb.w : 0.01f * vb.w) * w.w;
          asm volatile("" : "+v"(dot_b));
        }
      }
      if (isA && half == 1 && (cb + 1) % cb_per_head == 0) {
        const int h = cb / cb_per_head;
        float da = dot_a, db = dot_b;
        da += __shfl_xor(da, 16, 64); da += __shfl_xor(da, 32, 64);
        db += __shfl_xor(db, 16, 64); db += __shfl_xor(db, 32, 64);
        if (kg == 0) {
          const float bh = bA ? bA[h] : 0.f;
          if (row_a < E) a_out[(long)row_a * H + h] = da + bh;
          if (row_b < E) a_out[(long)row_b * H + h] = db + bh;
        }
        dot_a = 0.f; dot_b = 0.f;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef EX_TLOAD
#undef EX_READ
#undef EX_MFMA1
#undef EX_MFMA
}

// planes of the two 128 x 128 blocks W_e[a], W_j[a] (element (k, c) = W[(128 a + c) * ldw + k]) under one scale
__global__ __launch_bounds__(256) void prepare_W2_f16_kernel(const float* __restrict__ We, const float* __restrict__ Wj,
                                                             long ldw, _Float16* __restrict__ dst,
                                                             float* __restrict__ wmax) {
  __shared__ float wm[4];
  const int a = blockIdx.x, tid = threadIdx.x;
  float v[2][64];
  float m = 0.f;
#pragma unroll
  for (int src = 0; src < 2; ++src)
#pragma unroll
    for (int r = 0; r < 64; ++r) {
      const int i = r * 256 + tid, c = i >> 7, b = i & 127;      // k is the contiguous index of the source
      v[src][r] = (src ? Wj : We)[((long)a * 128 + c) * ldw + b];
      m = fmaxf(m, fabsf(v[src][r]));
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
  for (int src = 0; src < 2; ++src)
#pragma unroll
    for (int r = 0; r < 64; ++r) {
      const int i = r * 256 + tid, c = i >> 7, b = i & 127;
      const float x = v[src][r] * st;
      const _Float16 h = (_Float16)x, l = (_Float16)(x - (float)h);
      const int half = c >> 6, cb = (c & 63) >> 4, i16 = c & 15;
      const int kh = 2 * src + (b >> 6), s2 = (b >> 5) & 1, kg = (b & 31) >> 3, j = b & 7;
      const long blk = ((((long)a * 2 + half) * 4 + kh) * 2 + s2) * 2;
      const long in = (((long)cb * 4 + kg) * 16 + i16) * 8 + j;
      dst[(blk + 0) * 2048 + in] = h;
      dst[(blk + 1) * 2048 + in] = l;
    }
}

size_t edge_zx_wq_floats(int W2) { return (size_t)W2 * 256 + W2 / 128 + 64; }
bool edge_zx_fast(int C, int Ce, int W2, int H, int Hd, long ld_add, long ldz, const void* e, const void* x,
                  const void* Pi, const void* Z, const void* wA) {
  static int off = -1;
  if (off < 0) { const char* ev = getenv("CGAT_NO_EDGE_ZX"); off = (ev && ev[0] == '1') ? 1 : 0; }
  return !off && bilinear_mode() == 2 && C == 128 && Ce == 128 && W2 % 128 == 0 && Hd % 128 == 0 && H * Hd * 2 == W2 &&
         (ld_add % 4) == 0 && (ldz % 4) == 0 &&
         ((((uintptr_t)e) | ((uintptr_t)x) | ((uintptr_t)Pi) | ((uintptr_t)Z) | ((uintptr_t)wA)) & 15) == 0;
}
// We / Wj: the edge_attr and x_j slices of the stacked first-layer weight (row stride ldw); Wq: edge_zx_wq_floats(W2)
int edge_zx_launch(const float* e, long lde, const int* perm, const float* x, long ldx, const float* We, const float* Wj,
                   long ldw, float* Wq, int W2, const float* Pi, const int* dsti, const int* srci, long ld_add, float* Z,
                   long ldz, int E, const float* wA, const float* bA, int H, int Hd, float* a_out, hipStream_t stream, int z_bf16) {
  if (E <= 0) return CGAT_OK;
  const int ncb = W2 / 128;
  hipLaunchKernelGGL(prepare_W2_f16_kernel, dim3(ncb), dim3(256), 0, stream, We, Wj, ldw, (_Float16*)Wq,
                     Wq + (size_t)ncb * 32768);
  CGAT_LAUNCH_CHECK();
  CGAT_PROF("edge_z", stream);
#define EZX_GO(A_, ZB_) hipLaunchKernelGGL((edge_zx_kernel<A_, ZB_>), dim3(cdiv(E, 128)), dim3(256), 0, stream, e, lde, perm, x, ldx, (const uint4*)Wq, ncb, Pi, dsti, srci, ld_add, Z, ldz, E, wA, bA, H, Hd / 128, a_out)
#ifdef CGAT_DEV_ABLATIONS   // timing-only variants (wrong results) exist only in builds made for tools/edgez_ablation.sh
  const char* abl = getenv("CGAT_EZX_ABL");
  switch (abl ? atoi(abl) : 0) {
    case 1: EZX_GO(1, false); break; case 2: EZX_GO(2, false); break; case 3: EZX_GO(3, false); break; case 7: EZX_GO(7, false); break;
    default: EZX_GO(0, false); break;
  }
#else
  if (z_bf16) EZX_GO(0, true);
  else EZX_GO(0, false);
#endif
#undef EZX_GO
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

bool edge_z_fast(int Ce, int W2, int H, int Hd, long lde, long ld_add, long ldz, const void* e, const void* Pi,
                 const void* Pj, const void* Z, const void* wA) {
  return bilinear_mode() != 0 && Ce == 128 && W2 % 128 == 0 && Hd % 128 == 0 && H * Hd * 2 == W2 && (lde % 4) == 0 &&
         (ld_add % 4) == 0 && (ldz % 4) == 0 &&
         ((((uintptr_t)e) | ((uintptr_t)Pi) | ((uintptr_t)Pj) | ((uintptr_t)Z) | ((uintptr_t)wA)) & 15) == 0;
}

// floats of workspace for the pre-split weight
size_t edge_z_wq_floats(int W2) {
  const size_t a = ((size_t)W2 * 128 * 3 + 1) / 2, b = prepare_W_f16c_rows_floats(W2);   // three bf16 planes / the f16x3c image
  return a > b ? a : b;
}
// the per-edge launch in the f16x3c form (edge_zc_kernel): OPT-IN (CGAT_EDGE_ZC=1).  Measured (round 5, BASELINE shape):
// 3.2 -> 2.85 ms per launch, and with it the number of tensors of the sin-filled fixture `net_mean` that need the
// oracle-noise term of the parity criterion goes from 3 to 11 (all within 2.8 x the oracle's own fp32 deviation from
// fp64; the h + l + t form is 1.26 x the error of the exact six-pass split, tests/arith_cases.py) -- 1 % of the step is
// not worth the margin, so the default per-edge forward stays the six-pass kernel.  What the experiment established
// (tools/edgezc_ablate.sh): this launch is bound by its memory path, not by the matrix cores -- without the Z stores
// 1.86 ms, without the gathers 2.22, without both 1.42, without any matrix instruction still 2.83.
static bool edge_z6w_on() {   // CGAT_EDGE_Z6W=0: the 128-row form of the six-pass per-edge launch (A/B switch)
  static const bool on = [] { const char* e = getenv("CGAT_EDGE_Z6W"); return !(e && e[0] == '0'); }();
  return on;
}
// CGAT_Z_COL_GROUPS=0: no column groups over grid.y at few row tiles (A/B switch of the bit-identity test, like CGAT_EDGE_Z6W)
static int z_groups(int row_tiles, int ncb, int unit = 1) {
  static const bool on = [] { const char* e = getenv("CGAT_Z_COL_GROUPS"); return !(e && e[0] == '0'); }();
  return on ? z_col_groups(row_tiles, ncb, unit) : 1;
}
static bool edge_zc_on() {
  static const bool on = [] { const char* e = getenv("CGAT_EDGE_ZC"); return e && e[0] == '1'; }();
  return on;
}

// We: the edge_attr slice of the stacked first-layer weight, element (out, k) at We[out * ldw + k].
// With Pj == nullptr the kernel computes the plain product Z = e We^T + bias (Pi = bias vector or nullptr, no logits):
// the per-node projections of the operand split.
int edge_z_launch(const float* e, long lde, const int* perm, const float* We, long ldw, float* Wq, int W2,
                  const float* Pi, const int* dsti, const float* Pj, const int* srci, long ld_add, float* Z, long ldz,
                  int E, const float* wA, const float* bA, int H, int Hd, float* a_out, hipStream_t stream, int act,
                  float* omax, int z_bf16, int n_add_rows) {
  if (E <= 0) return CGAT_OK;
  const int ncb = W2 / 128;
  CGAT_CHECK_ARG(!z_bf16 || (Pj != nullptr && bilinear_mode() != 2 && bilinear_mode() != 3 && bilinear_mode() != 0 &&
                             act == CGAT_ACT_NONE && !omax),
                 "edge_z: bf16 storage is the per-edge launch of the six-pass form only");
  // (fc_out_A's weight is staged in 8 KB of LDS; the gathered rows are addressed by 32-bit byte offsets)
  if (bilinear_mode() == 4 && Pj != nullptr && perm && act == CGAT_ACT_NONE && !omax && edge_zc_on() && (ldw % 4) == 0 &&
      (((uintptr_t)We) & 15) == 0 && (!a_out || (long)H * Hd <= 2048) && ncb <= 32 && n_add_rows > 0 && (long)n_add_rows * 4 * ld_add < (1l << 32)) {
    CGAT_TRY(prepare_W_f16c_rows_launch(We, ldw, W2, Wq, stream));
    CGAT_PROF("edge_z", stream);
    if (z_bf16)
      hipLaunchKernelGGL(edge_zc_kernel<true>, dim3(cdiv(E, 256)), dim3(512), 0, stream, e, lde, perm, (const uint4*)Wq, ncb, Pi,
                         dsti, Pj, srci, ld_add, Z, ldz, E, wA, bA, H, Hd / 128, a_out);
    else
      hipLaunchKernelGGL(edge_zc_kernel<false>, dim3(cdiv(E, 256)), dim3(512), 0, stream, e, lde, perm, (const uint4*)Wq, ncb, Pi,
                         dsti, Pj, srci, ld_add, Z, ldz, E, wA, bA, H, Hd / 128, a_out);
    CGAT_LAUNCH_CHECK();
    return CGAT_OK;
  }
  // operand (a = column block, b = k, c = column in block) = We[(128 a + c) * ldw + b]
  if (bilinear_mode() == 2) CGAT_TRY(prepare_W_f16_launch(We, Wq, ncb, 128 * ldw, 1, ldw, stream));
  else CGAT_TRY(prepare_T_bf16_launch(We, Wq, ncb, 128 * ldw, 1, ldw, 0, stream));
  CGAT_PROF(Pj ? "edge_z" : "edge_proj", stream);   // the per-edge launch / the per-node projections
  // the six-pass per-edge launch on 256-row workgroups (edge_z6w_kernel: same arithmetic, bit-identical results)
  // (below 128 of its 256-row tiles -- the harness' shipped batch: 60 workgroups walking 12 blocks, 119 us -- the 128-row
  // kernel with its column blocks dealt to grid.y groups fills the chip instead: bit-identical arithmetic, see below)
  const int unit_z = a_out ? Hd / 128 : 1;
  const bool few_rows = !z_bf16 && cdiv(E, 256) < 128 && z_groups(cdiv(E, 128), ncb, unit_z) > 1;
  if ((bilinear_mode() == 4 || bilinear_mode() == 6) && Pj != nullptr && perm && (act == CGAT_ACT_NONE || act == CGAT_ACT_LEAKY) &&
      !(omax && a_out) && edge_z6w_on() && !few_rows &&   // (the running maximum shares a register with the logits)
      (!a_out || (long)H * Hd <= 2048) && n_add_rows > 0 && (long)n_add_rows * 4 * ld_add < (1l << 32)) {
    if (z_bf16)
      hipLaunchKernelGGL(edge_z6w_kernel<true>, dim3(cdiv(E, 256)), dim3(512), 0, stream, e, lde, perm, (const uint4*)Wq, ncb, Pi,
                         dsti, Pj, srci, ld_add, Z, ldz, E, wA, bA, H, Hd / 128, a_out, act, omax);
    else
      hipLaunchKernelGGL(edge_z6w_kernel<false>, dim3(cdiv(E, 256)), dim3(512), 0, stream, e, lde, perm, (const uint4*)Wq, ncb, Pi,
                         dsti, Pj, srci, ld_add, Z, ldz, E, wA, bA, H, Hd / 128, a_out, act, omax);
    CGAT_LAUNCH_CHECK();
    return CGAT_OK;
  }
  const int grid = cdiv(E, 128);
  // Few rows beside many column blocks -- at the harness' shipped batch the per-node projections are 10 workgroups and the
  // per-edge first layer of a vector-attention layer 240, each walking 20 blocks (149 / 180 us per launch, 2.4 ms of an
  // 18-ms step) -- : the column blocks are dealt to grid.y groups (round 6).  A workgroup then splits its rows once per group
  // (64 KB, L2-resident) and walks ncb / G blocks; every block is computed as before: bit-identical.  The 24-bit modes
  // without logits (the fp16 image keeps its scale behind the LAST block).
  // With logits (a_out) a group holds whole heads: a head's logit is the sum over ITS column blocks only, and the group
  // carries the number of blocks in front of it (head index, fc_out_A's weight row).
  int G = 1;
  if ((bilinear_mode() == 4 || bilinear_mode() == 6) && !z_bf16) G = z_groups(grid, ncb, a_out ? Hd / 128 : 1);
  const int ncb_g = ncb / G;
  const HeadBatch hbz = {0, (long)ncb_g * 6144, (long)ncb_g * 128, (long)ncb_g * 128, 0, (long)ncb_g * 128, (long)ncb_g};
#define EZ_GO(P_, A_)                                                                                                \
  hipLaunchKernelGGL((edge_z_kernel<P_, A_>), dim3(grid, G), dim3(256), 0, stream, e, lde, perm, (const uint4*)Wq, ncb_g, \
                     Pi, dsti, Pj, srci, ld_add, Z, ldz, E, wA, bA, H, Hd / 128, a_out, act, 0, omax, (const float*)nullptr, 0l, \
                     G > 1 ? hbz : HeadBatch{})
  const bool adds = Pj != nullptr;
  if (bilinear_mode() == 2) { if (adds) EZ_GO(2, true); else EZ_GO(2, false); }
  else if (bilinear_mode() != 3 && z_bf16)
    hipLaunchKernelGGL((edge_z_kernel<6, true, true>), dim3(grid), dim3(256), 0, stream, e, lde, perm, (const uint4*)Wq, ncb,
                       Pi, dsti, Pj, srci, ld_add, Z, ldz, E, wA, bA, H, Hd / 128, a_out, act, 0, omax, (const float*)nullptr,
                       0l, HeadBatch{});
  else if (bilinear_mode() != 3) { if (adds) EZ_GO(6, true); else EZ_GO(6, false); }
  else { if (adds) EZ_GO(3, true); else EZ_GO(3, false); }
#undef EZ_GO
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// ---- dense layer at width 128: out[n, :] = act(in[n, :] W^T + bias) + beta * out[n, :],  W(o, k) = W[o*so + k*sk] ----
// The hypernetwork trunks (Linear + Tanh, reference Hypernetworksmp.py:24-60) and the linear terms of the predicted
// layers are [rows,128] x [128,128] products; on the generic 128x128x32 GEMM tile they are all prologue and epilogue.
// Here they run as the one-column-block case of the kernel above (rows split once into registers, 96 KB of weight
// planes through the LDS-DMA ring).
bool linear128_fast(int K, int N, long ldi, long ldo, const void* in, const void* out) {
  return bilinear_mode() != 0 && K == 128 && N >= 128 && N % 128 == 0 && (ldi % 4) == 0 && (ldo % 4) == 0 &&
         ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0;
}
size_t linear128_ws_bytes(int n_out) { return ws_round((size_t)n_out * 128 * 3 / 2 + 4, 4); }
int linear128_launch(const float* in, long ldi, const float* W, long so, long sk, const float* bias, int act, int accumulate,
                     float* out, long ldo, int rows, void* ws, hipStream_t stream, int n_out, const void* prepared,
                     const float* dact, long ld_dact, float* omax) {
  if (rows <= 0) return CGAT_OK;
  const int ncb = n_out / 128;
  // operand (a = output block, b = k, c = output in block) = W[(128 a + c) * so + b * sk]
  // prepared: the mode's image of this weight made ahead of time (f16x3: prepare_W_f16_batch_launch; the bf16 forms:
  // prepare_T_bf16_batch_launch), n_out == 128 only
  if (prepared && n_out == 128) ws = const_cast<void*>(prepared);
  else if (bilinear_mode() == 2) CGAT_TRY(prepare_W_f16_launch(W, ws, ncb, 128 * so, sk, so, stream));
  else CGAT_TRY(prepare_T_bf16_launch(W, ws, ncb, 128 * so, sk, so, 0, stream));
  CGAT_PROF("linear128", stream);
  const int grid = cdiv(rows, 128);
  // (column blocks over grid.y when the row tiles leave CUs idle: edge_z_launch above)
  const int G = (bilinear_mode() == 4 || bilinear_mode() == 6) ? z_groups(grid, ncb) : 1;
  const int ncb_g = ncb / G;
  const HeadBatch hbz = {0, (long)ncb_g * 6144, (long)ncb_g * 128, (long)ncb_g * 128, (long)ncb_g * 128, 0};
#define L128_GO(P_)                                                                                                   \
  hipLaunchKernelGGL((edge_z_kernel<P_, false>), dim3(grid, G), dim3(256), 0, stream, in, ldi, (const int*)nullptr,   \
                     (const uint4*)ws, ncb_g, bias, (const int*)nullptr, (const float*)nullptr, (const int*)nullptr, 0l, \
                     out, ldo, rows, (const float*)nullptr, (const float*)nullptr, 1, 1, (float*)nullptr, act, accumulate, \
                     omax, dact, ld_dact, G > 1 ? hbz : HeadBatch{})
  if (bilinear_mode() == 2) L128_GO(2); else if (bilinear_mode() != 3) L128_GO(6); else L128_GO(3);
#undef L128_GO
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// The same product for `heads` independent (input, weight, output) triples in ONE launch pair (f16x3 mode): head h
// reads in + h * s_in, W + h * s_w (the same (so, sk) element strides), bias + h * s_bias, dact + h * s_dact and writes
// out + h * s_out.  The per-head second layers of a vector-attention layer at the harness' shipped batch are 2 H chains
// of [prepare, product] launches of 20-40 us each that fill a fraction of the chip; as grid.y they are one.
// ws: heads * linear128_heads_image_floats(n_out) floats.
// (sized for the six-pass image: three bf16 planes per 128 x 128 block)
size_t linear128_heads_image_floats(int n_out) { return (size_t)(n_out / 128) * 24576 + ((n_out / 128 + 3) & ~3); }
int linear128_heads_launch(int heads, const float* in, long ldi, long s_in, const float* W, long so, long sk, long s_w,
                           const float* bias, long s_bias, int act, int accumulate, float* out, long ldo, long s_out, int rows,
                           void* ws, hipStream_t stream, int n_out, const float* dact, long ld_dact, long s_dact,
                           float* omax) {
  if (rows <= 0 || heads <= 0) return CGAT_OK;
  CGAT_CHECK_ARG((bilinear_mode() == 2 || bilinear_mode() == 4 || bilinear_mode() == 6) && n_out % 128 == 0,
                 "linear128_heads: split arithmetic modes and 128-wide output blocks only");
  const int ncb = n_out / 128;
  const long img = (long)linear128_heads_image_floats(n_out);
  if (bilinear_mode() != 2) {     // the 24-bit modes (round 6): one image launch for all heads, one six-pass product launch
    CGAT_TRY(prepare_T_bf16_heads_launch(W, ws, ncb, 128 * so, sk, so, 0, heads, s_w, img, stream));
    CGAT_PROF("linear128", stream);
    const HeadBatch hb6 = {s_in, img / 4, s_bias, s_out, s_dact};
    hipLaunchKernelGGL((edge_z_kernel<6, false>), dim3(cdiv(rows, 128), heads), dim3(256), 0, stream, in, ldi,
                       (const int*)nullptr, (const uint4*)ws, ncb, bias, (const int*)nullptr, (const float*)nullptr,
                       (const int*)nullptr, 0l, out, ldo, rows, (const float*)nullptr, (const float*)nullptr, 1, 1,
                       (float*)nullptr, act, accumulate, omax, dact, ld_dact, hb6);
    CGAT_LAUNCH_CHECK();
    return CGAT_OK;
  }
  CGAT_TRY(prepare_W_f16_heads_launch(W, ws, ncb, 128 * so, sk, so, heads, s_w, img, stream));
  CGAT_PROF("linear128", stream);
  const HeadBatch hb = {s_in, img / 4, s_bias, s_out, s_dact};
  hipLaunchKernelGGL((edge_z_kernel<2, false>), dim3(cdiv(rows, 128), heads), dim3(256), 0, stream, in, ldi,
                     (const int*)nullptr, (const uint4*)ws, ncb, bias, (const int*)nullptr, (const float*)nullptr,
                     (const int*)nullptr, 0l, out, ldo, rows, (const float*)nullptr, (const float*)nullptr, 1, 1,
                     (float*)nullptr, act, accumulate, omax, dact, ld_dact, hb);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}
