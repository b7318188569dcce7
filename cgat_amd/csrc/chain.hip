// A chain of width-128 dense layers in ONE launch: the hypernetwork's trunks (reference Hypernetworksmp.py:36-83 FCBlock:
// four Linear(128,128)+Tanh, then the linear terms of the predicted layer) forward and backward, and the edge update's
// two-layer network with its residual (CGAT.py:226-229 + CGAtNet's `edge_attr + Edge(...)`, 580-585).
//
// Round 1 ran every layer as its own launch of the dense-layer kernel (edgez.hip, edge_z_kernel<.., false>): 48 of
// them per hypernetwork step plus 16 activation-derivative passes, each reading and writing its [rows, 128] operands
// from HBM.  Here a workgroup keeps its 128 rows through all layers:
//
//   rows    a wave's 32 rows, scaled by a power of two per row and split into two fp16 planes (mfma_bf16.h), live in
//           64 VGPRs as the B operands of v_mfma_f32_16x16x32_f16 (three passes per product, fp32 accumulation);
//   weights every layer's prepared image (prepare_W_f16_batch: 8 chunks of 8 KB in A-fragment order, its scale behind)
//           streams through a 4-slot LDS ring by LDS-DMA, the chunks of layer l + 1 following those of layer l without
//           a seam;
//   layer   out = act(rows W^T + bias) (+ residual) (* act'(saved activation): backward) is stored if the caller wants
//           it (the activations backward needs, the pre-activation gradients the weight-gradient kernel needs) and
//           becomes the next layer's rows: the lane holding row n, columns 16 b + 4 kg .. + 3 (MFMA C layout) needs
//           row n, columns 32 s + 8 kg .. + 7 (B layout); the four lanes of a row exchange through a wave-private LDS
//           tile, the lane takes the new row maximum and splits again.  Nothing but the requested outputs touches HBM.
//
// Backward of a trunk is the same chain on the transposed weights: rows = g_z * tanh'(t_4) (input multiply, stored as
// the first pre-activation gradient), layer i: g_t = rows W_(3-i), times tanh'(t_(3-i)), stored, ..., the last product
// accumulated into the gradient of the hyper input.
#include "common.h"
#include "kernels.h"
#include "mfma_bf16.h"

__device__ __forceinline__ float4 chain_deriv(float4 v, float4 y, int type) {
  if (type == CGAT_ACT_TANH)
    return make_float4(v.x * (1.f - y.x * y.x), v.y * (1.f - y.y * y.y), v.z * (1.f - y.z * y.z), v.w * (1.f - y.w * y.w));
  if (type == CGAT_ACT_LEAKY)
    return make_float4(v.x * (y.x > 0.f ? 1.f : 0.01f), v.y * (y.y > 0.f ? 1.f : 0.01f), v.z * (y.z > 0.f ? 1.f : 0.01f),
                       v.w * (y.w > 0.f ? 1.f : 0.01f));
  if (type == CGAT_ACT_RELU)
    return make_float4(y.x > 0.f ? v.x : 0.f, y.y > 0.f ? v.y : 0.f, y.z > 0.f ? v.z : 0.f, y.w > 0.f ? v.w : 0.f);
  return v;
}
// tanh(x) = 1 - 2 / (2^(2 x log2 e) + 1) on the hardware exp2 / rcp (1 ulp each): absolute error <= 2e-7 over the
// whole range (saturates to +-1 through 2^x = inf / 0), 5 instructions instead of OCML tanhf's ~25 -- the chain is
// bound by its vector work (64 tanh per lane and layer), and backward differentiates the STORED value, so forward and
// backward stay consistent.  The relative error of tiny outputs (|x| < 1e-3) is larger than tanhf's; every parity
// criterion of the path is a max-norm one.
__device__ __forceinline__ float chain_tanh(float x) {
  const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);
}
__device__ __forceinline__ float4 chain_act(float4 v, int act) {
  if (act == CGAT_ACT_TANH) return make_float4(chain_tanh(v.x), chain_tanh(v.y), chain_tanh(v.z), chain_tanh(v.w));
  if (act == CGAT_ACT_LEAKY)
    return make_float4(v.x > 0.f ? v.x : 0.01f * v.x, v.y > 0.f ? v.y : 0.01f * v.y, v.z > 0.f ? v.z : 0.01f * v.z,
                       v.w > 0.f ? v.w : 0.01f * v.w);
  if (act == CGAT_ACT_RELU) return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
  return v;
}

__global__ __launch_bounds__(256, 2) void mlp_chain128_kernel(ChainDesc d) {
  constexpr int CH16 = 2 * 4 * 64;              // 16-byte pieces per chunk: two planes x four 16-column blocks x 64 lanes
  constexpr int XP = 68;                        // pitch (floats) of the layout-exchange tile: 64 columns + 4
  __shared__ uint4 smem[4 * CH16];
  __shared__ __attribute__((aligned(16))) float xch[4 * 32 * XP];   // per wave: 32 rows x 64 columns of one half
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kg = lane >> 4;
  const int rows = d.rows;
  const int row_w = blockIdx.x * 128 + wave * 32;
  const int row_a = row_w + n16, row_b = row_a + 16;
  const long rca = row_a < rows ? row_a : rows - 1, rcb = row_b < rows ? row_b : rows - 1;
  const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const unsigned wave_t = __builtin_amdgcn_readfirstlane(sbase + wave * 1024);
  const bf16x8* ring = reinterpret_cast<const bf16x8*>(smem) + lane;
  const unsigned t_off = (unsigned)tid * 16;
  const int n_chunks = d.n_layers * 8;

  // ---- the lane's two rows: q[plane][2 s + nb] holds x[row(nb), 32 s + 8 kg + 0..7] ----
  bf16x8 q1[8], q2[8];
  float rs_a, rs_b;
  {
    float qv[2][32];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const long rr = nb ? rcb : rca;
      const bool live = (nb ? row_b : row_a) < rows;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int col = 32 * s + 8 * kg;
        const float4* xp = reinterpret_cast<const float4*>(d.x + rr * d.ldx + col);
        float4 t0 = xp[0], t1 = xp[1];
        if (d.in_dact) {                          // backward: rows = g * act'(saved activation)
          const float4* yp = reinterpret_cast<const float4*>(d.in_dact + rr * d.ld_in_dact + col);
          t0 = chain_deriv(t0, yp[0], d.in_dact_type);
          t1 = chain_deriv(t1, yp[1], d.in_dact_type);
        }
        if (d.in_store && live) {
          float4* sp = reinterpret_cast<float4*>(d.in_store + rr * d.ld_in_store + col);
          sp[0] = t0; sp[1] = t1;
        }
        qv[nb][8 * s + 0] = t0.x; qv[nb][8 * s + 1] = t0.y; qv[nb][8 * s + 2] = t0.z; qv[nb][8 * s + 3] = t0.w;
        qv[nb][8 * s + 4] = t1.x; qv[nb][8 * s + 5] = t1.y; qv[nb][8 * s + 6] = t1.z; qv[nb][8 * s + 7] = t1.w;
      }
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
      (nb ? rs_b : rs_a) = iq;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = qv[nb][8 * s + j] * sq;
        split2_x8_f16(v, q1[2 * s + nb], q2[2 * s + nb]);
        if (s & 1) { q1[2 * s + nb] = neg_x8(q1[2 * s + nb]); q2[2 * s + nb] = neg_x8(q2[2 * s + nb]); }   // (see CN_MFMA)
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // chunk g = 8 l + 4 half + s of the whole chain -> ring slot g % 4; past the end the last chunk is re-loaded (keeps
  // the counted waits uniform)
#define CN_TLOAD(g_)                                                                           \
  {                                                                                            \
    const int gg = (g_) < n_chunks ? (g_) : n_chunks - 1;                                      \
    const uint4* tb = d.layer[gg >> 3].W + (long)(gg & 7) * CH16;                              \
    const unsigned dst = wave_t + (unsigned)((g_) & 3) * (CH16 * 16);                          \
    glds_b128(tb, t_off, dst);                                                                 \
    glds_b128(tb + 256, t_off, dst + 4096);                                                    \
  }
  CN_TLOAD(0);
  CN_TLOAD(1);
  CN_TLOAD(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  bf16x8 fa1, fa2, fb1, fb2;
#define CN_READ(F1_, F2_, slot_, cb_)                                                          \
  {                                                                                            \
    const bf16x8* fp = ring + (slot_) * (CH16) + (cb_) * 64;                                   \
    F1_ = fp[0];                                                                               \
    F2_ = fp[4 * 64];                                                                          \
  }
#define CN_MFMA1(F1_, F2_, qi_, P_)                                                            \
  {                                                                                            \
    P_ = mma16<true>(F2_, q1[qi_], P_);                                                        \
    P_ = mma16<true>(F1_, q2[qi_], P_);                                                        \
    P_ = mma16<true>(F1_, q1[qi_], P_);                                                        \
  }
  // The matrix instruction's accumulator rounds with a sign-independent bias (tools/f16_bias_probe.py): the
  // odd k-steps multiply the NEGATED row fragments into a second accumulator set that is subtracted before the epilogue,
  // so their bias enters the result with the opposite sign and cancels the even k-steps' pairwise.
#define CN_MFMA(F1_, F2_, s_, cb_)                                                             \
  {                                                                                            \
    if ((s_) & 1) {                                                                            \
      CN_MFMA1(F1_, F2_, 2 * (s_) + 0, partn[2 * (cb_) + 0])                                   \
      CN_MFMA1(F1_, F2_, 2 * (s_) + 1, partn[2 * (cb_) + 1])                                   \
    } else {                                                                                   \
      CN_MFMA1(F1_, F2_, 2 * (s_) + 0, part[2 * (cb_) + 0])                                    \
      CN_MFMA1(F1_, F2_, 2 * (s_) + 1, part[2 * (cb_) + 1])                                    \
    }                                                                                          \
  }
  CN_READ(fa1, fa2, 0, 0);
  f32x4 part[8], partn[8];
  f32x4 vals[2][8];                              // vals[nb][b] = this layer's out[row(nb), 16 b + 4 kg .. + 3]
  for (int l = 0; l < d.n_layers; ++l) {
    const ChainLayer L = d.layer[l];
    const float* wmax = reinterpret_cast<const float*>(L.W + 8 * CH16);
    float sw, iw;
    pow2_scale(wmax[0], sw, iw);
    const float ma = rs_a * iw, mb = rs_b * iw;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { part[i] = f32x4{0.f, 0.f, 0.f, 0.f}; partn[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {                  // chunk (l, half, s) sits in ring slot s
        CN_TLOAD(l * 8 + half * 4 + s + 3);
#pragma unroll
        for (int cbp = 0; cbp < 2; ++cbp) {
          CN_READ(fb1, fb2, s, 2 * cbp + 1);
          __builtin_amdgcn_sched_barrier(0);
          CN_MFMA(fa1, fa2, s, 2 * cbp);
          if (cbp == 0) CN_READ(fa1, fa2, s, 2)
          else CN_READ(fa1, fa2, (s + 1) & 3, 0);
          __builtin_amdgcn_sched_barrier(0);
          CN_MFMA(fb1, fb2, s, 2 * cbp + 1);
        }
        // chunk g + 2 (issued one k-step ago) must have landed; younger than it: this step's two loads.  The epilogue's
        // ordinary loads and stores are older than the next step's loads, so the counted wait can only be stricter
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      // ---- epilogue of the 64-column half ----
      const int col0 = half * 64 + 4 * kg;
#pragma unroll
      for (int c16 = 0; c16 < 4; ++c16) {
        const int col = col0 + 16 * c16;
        const f32x4 pa = (part[2 * c16 + 0] - partn[2 * c16 + 0]) * ma, pb = (part[2 * c16 + 1] - partn[2 * c16 + 1]) * mb;
        float4 va = make_float4(pa[0], pa[1], pa[2], pa[3]), vb = make_float4(pb[0], pb[1], pb[2], pb[3]);
        if (L.bias) {
          const float4 b4 = *reinterpret_cast<const float4*>(L.bias + col);
          va.x += b4.x; va.y += b4.y; va.z += b4.z; va.w += b4.w;
          vb.x += b4.x; vb.y += b4.y; vb.z += b4.z; vb.w += b4.w;
        }
        va = chain_act(va, L.act);
        vb = chain_act(vb, L.act);
        if (L.resid) {
          const float4 ra = *reinterpret_cast<const float4*>(L.resid + rca * L.ld_resid + col);
          const float4 rb = *reinterpret_cast<const float4*>(L.resid + rcb * L.ld_resid + col);
          va.x += ra.x; va.y += ra.y; va.z += ra.z; va.w += ra.w;
          vb.x += rb.x; vb.y += rb.y; vb.z += rb.z; vb.w += rb.w;
        }
        if (L.dact) {
          va = chain_deriv(va, *reinterpret_cast<const float4*>(L.dact + rca * L.ld_dact + col), L.dact_type);
          vb = chain_deriv(vb, *reinterpret_cast<const float4*>(L.dact + rcb * L.ld_dact + col), L.dact_type);
        }
        if (L.out) {
          float* oa = L.out + rca * L.ld_out + col;
          float* ob = L.out + rcb * L.ld_out + col;
          if (L.accumulate) {
            const float4 ua = *reinterpret_cast<const float4*>(oa), ub = *reinterpret_cast<const float4*>(ob);
            if (row_a < rows) *reinterpret_cast<float4*>(oa) = make_float4(ua.x + va.x, ua.y + va.y, ua.z + va.z, ua.w + va.w);
            if (row_b < rows) *reinterpret_cast<float4*>(ob) = make_float4(ub.x + vb.x, ub.y + vb.y, ub.z + vb.z, ub.w + vb.w);
          } else {
            if (row_a < rows) *reinterpret_cast<float4*>(oa) = va;
            if (row_b < rows) *reinterpret_cast<float4*>(ob) = vb;
          }
        }
        vals[0][half * 4 + c16] = f32x4{va.x, va.y, va.z, va.w};
        vals[1][half * 4 + c16] = f32x4{vb.x, vb.y, vb.z, vb.w};
      }
    }
    if (l + 1 < d.n_layers) {
      // ---- this layer's outputs become the next layer's rows: C layout -> B layout inside the four lanes of a row ----
      float sq[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        float m = 0.f;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
          for (int j = 0; j < 4; ++j) m = fmaxf(m, fabsf(vals[nb][b][j]));
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float iq;
        pow2_scale(m, sq[nb], iq);
        (nb ? rs_b : rs_a) = iq;
      }
      // through a wave-private LDS tile, one 64-column half at a time (16 b128 writes + 16 b128 reads per layer; the
      // first version used 128 ds_bpermute and was slower than separate launches).  LDS serves a wave's accesses in
      // order, so the write -> read -> overwrite sequence needs no barrier.
      float* xw = xch + wave * (32 * XP);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int c16 = 0; c16 < 4; ++c16) {
            const f32x4 t = vals[nb][4 * h + c16];
            *reinterpret_cast<float4*>(xw + (n16 + 16 * nb) * XP + 16 * c16 + 4 * kg) = make_float4(t[0], t[1], t[2], t[3]);
          }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int sl = 0; sl < 2; ++sl) {
            const float* rp = xw + (n16 + 16 * nb) * XP + 32 * sl + 8 * kg;
            const float4 lo = *reinterpret_cast<const float4*>(rp), hi4 = *reinterpret_cast<const float4*>(rp + 4);
            const float f = sq[nb];
            const float v[8] = {lo.x * f, lo.y * f, lo.z * f, lo.w * f, hi4.x * f, hi4.y * f, hi4.z * f, hi4.w * f};
            split2_x8_f16(v, q1[2 * (2 * h + sl) + nb], q2[2 * (2 * h + sl) + nb]);
            if (sl & 1) {                            // k-step s = 2 h + sl is odd
              q1[2 * (2 * h + sl) + nb] = neg_x8(q1[2 * (2 * h + sl) + nb]);
              q2[2 * (2 * h + sl) + nb] = neg_x8(q2[2 * (2 * h + sl) + nb]);
            }
          }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef CN_TLOAD
#undef CN_READ
#undef CN_MFMA1
#undef CN_MFMA
}

// ---------------------------------------------------------------------------------------
// The same chain in the 24-bit arithmetic modes ("f16x3c", "bf16x6"; round 4): every fp32 value split exactly into three
// bf16 pieces, a product = six v_mfma_f32_16x16x32_bf16 passes (mfma_bf16.h), no scales.  The chain is bound by the
// activations it moves and by its vector work (matrix-core utilisation 0.1 in the fp16 form), so the six passes cost
// little; what they need is registers: 96 for a wave's rows instead of 64.  The partial accumulators therefore cover a
// PAIR of 16-column blocks over the whole K instead of a 64-column half per k-step (8 + 8 registers... 16 + 16 instead
// of 32 + 32), i.e. the weight image is cut differently:
//   image of one 128 x 128 layer (prepare_W_x6_batch_kernel): chunk (half, cbp, kh) = 12 KB:
//       [s2 (2)][plane (3)][cb2 (2)][lane 64 x 16 B],  lane = 16 kg + c % 16 holds W[c][b = 64 kh + 32 s2 + 8 kg + j]
//   for the output column c = 64 half + 32 cbp + 16 cb2 + c % 16; eight chunks per layer in the order (half, cbp, kh).
// Everything else -- ring, layout exchange between layers, epilogue, negated odd k-steps into a second accumulator set
// (the bf16 matrix instruction's accumulator bias cancels) -- is mlp_chain128_kernel's.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prepare_W_x6_batch_kernel(WPrepBatch b, float* __restrict__ dst) {
  const int it = blockIdx.x, tid = threadIdx.x;
  const float* src = b.src[it];
  const long sb = b.sb[it], sc = b.sc[it];
  __bf16* img = reinterpret_cast<__bf16*>(dst + (size_t)it * WPREP_IMAGE_FLOATS_X6);
#pragma unroll 4
  for (int r = 0; r < 64; ++r) {
    const int i = r * 256 + tid;                  // thread order follows the fastest source stride
    int bb, c;
    if (sc == 1) { bb = i >> 7; c = i & 127; }
    else { c = i >> 7; bb = i & 127; }
    __bf16 x1, x2, x3;
    split3_bf16(src[bb * sb + c * sc], x1, x2, x3);
    const int half = c >> 6, cbp = (c >> 5) & 1, cb2 = (c >> 4) & 1, i16 = c & 15;
    const int kh = bb >> 6, s2 = (bb >> 5) & 1, kg = (bb >> 3) & 3, j = bb & 7;
    const long chunk = ((long)(half * 2 + cbp) * 2 + kh) * 6144;          // 12 KB = 6144 bf16
    const long in = ((long)cb2 * 64 + kg * 16 + i16) * 8 + j;
    img[chunk + ((s2 * 3 + 0) * 2) * 512 + in] = x1;
    img[chunk + ((s2 * 3 + 1) * 2) * 512 + in] = x2;
    img[chunk + ((s2 * 3 + 2) * 2) * 512 + in] = x3;
  }
}
int prepare_W_x6_batch_launch(const WPrepBatch& b, float* dst, hipStream_t stream) {
  if (b.n <= 0) return CGAT_OK;
  hipLaunchKernelGGL(prepare_W_x6_batch_kernel, dim3(b.n), dim3(256), 0, stream, b, dst);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// (the body as a device function: the single and the batched kernel below call it with their descriptor)
#if !defined(CGAT_DEV_ABLATIONS)   // the product build: the timing-only variants below do not exist, whatever -DCX_ABL says
#undef CX_ABL
#define CX_ABL 0
#elif !defined(CX_ABL)
#define CX_ABL 0   // timing-only ablations (wrong results): 1 no epilogue loads (bias, residual, derivative, accumulate), 2 no stores
#endif
__device__ __forceinline__ void mlp_chain128_x6_body(const ChainDesc& d) {
  constexpr int CH16 = 2 * 3 * 2 * 64;          // 16-byte pieces per chunk: two k-steps x three planes x two blocks x 64 lanes
  constexpr int XP = 36;                        // pitch (floats) of the layout-exchange tile: 32 columns + 4
  __shared__ uint4 smem[4 * CH16];
  __shared__ __attribute__((aligned(16))) float xch[4 * 32 * XP];   // per wave: 32 rows x 32 columns (one k-step) at a time
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kg = lane >> 4;
  const int rows = d.rows;
  const int row_w = blockIdx.x * 128 + wave * 32;
  const int row_a = row_w + n16, row_b = row_a + 16;
  const long rca = row_a < rows ? row_a : rows - 1, rcb = row_b < rows ? row_b : rows - 1;
  const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const unsigned wave_t = __builtin_amdgcn_readfirstlane(sbase + wave * 1024);
  const bf16x8* ring = reinterpret_cast<const bf16x8*>(smem) + lane;
  const unsigned t_off = (unsigned)tid * 16;
  const int n_chunks = d.n_layers * 8;

  // ---- the lane's two rows: q[plane][2 s + nb] holds x[row(nb), 32 s + 8 kg + 0..7]; odd k-steps negated (CX_MFMA) ----
  bf16x8 q1[8], q2[8], q3[8];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const long rr = nb ? rcb : rca;
    const bool live = (nb ? row_b : row_a) < rows;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int col = 32 * s + 8 * kg;
      const float4* xp = reinterpret_cast<const float4*>(d.x + rr * d.ldx + col);
      float4 t0 = xp[0], t1 = xp[1];
      if (d.in_dact) {                          // backward: rows = g * act'(saved activation)
        const float4* yp = reinterpret_cast<const float4*>(d.in_dact + rr * d.ld_in_dact + col);
        t0 = chain_deriv(t0, yp[0], d.in_dact_type);
        t1 = chain_deriv(t1, yp[1], d.in_dact_type);
      }
      if (d.in_store && live) {
        float4* sp = reinterpret_cast<float4*>(d.in_store + rr * d.ld_in_store + col);
        sp[0] = t0; sp[1] = t1;
      }
      const float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
      split3_x8(v, q1[2 * s + nb], q2[2 * s + nb], q3[2 * s + nb]);
      if (s & 1) { q1[2 * s + nb] = neg_x8(q1[2 * s + nb]); q2[2 * s + nb] = neg_x8(q2[2 * s + nb]); q3[2 * s + nb] = neg_x8(q3[2 * s + nb]); }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // chunk g = 8 l + 4 half + 2 cbp + kh of the whole chain -> ring slot g % 4; past the end the last chunk is re-loaded
#define CX_TLOAD(g_)                                                                           \
  {                                                                                            \
    const int gg = (g_) < n_chunks ? (g_) : n_chunks - 1;                                      \
    const uint4* tb = d.layer[gg >> 3].W + (long)(gg & 7) * CH16;                              \
    const unsigned dst = wave_t + (unsigned)((g_) & 3) * (CH16 * 16);                          \
    glds_b128(tb, t_off, dst);                                                                 \
    glds_b128(tb + 256, t_off, dst + 4096);                                                    \
    glds_b128(tb + 512, t_off, dst + 8192);                                                    \
  }
  CX_TLOAD(0);
  CX_TLOAD(1);
  CX_TLOAD(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  bf16x8 fa1, fa2, fa3, fb1, fb2, fb3;
  // group gq = 2 s2 + cb2 of the chunk in ring slot slot_: the three planes of block cb2 at k-step s2
#define CX_READ(F1_, F2_, F3_, slot_, gq_)                                                     \
  {                                                                                            \
    const bf16x8* fp = ring + (slot_) * (CH16) + ((((gq_) >> 1) * 3) * 2 + ((gq_) & 1)) * 64;  \
    F1_ = fp[0];                                                                               \
    F2_ = fp[2 * 64];                                                                          \
    F3_ = fp[4 * 64];                                                                          \
  }
#define CX_MFMA1(F1_, F2_, F3_, qi_, P_)                                                       \
  {                                                                                            \
    P_ = mma16<false>(F3_, q1[qi_], P_);                                                       \
    P_ = mma16<false>(F1_, q3[qi_], P_);                                                       \
    P_ = mma16<false>(F2_, q2[qi_], P_);                                                       \
    P_ = mma16<false>(F2_, q1[qi_], P_);                                                       \
    P_ = mma16<false>(F1_, q2[qi_], P_);                                                       \
    P_ = mma16<false>(F1_, q1[qi_], P_);                                                       \
  }
  // k-step s_ = 2 kh + s2; odd k-steps (negated rows) go to the second accumulator set, subtracted in the epilogue
#define CX_MFMA(F1_, F2_, F3_, s_, cb2_)                                                       \
  {                                                                                            \
    if ((s_) & 1) {                                                                            \
      CX_MFMA1(F1_, F2_, F3_, 2 * (s_) + 0, partn[2 * (cb2_) + 0])                             \
      CX_MFMA1(F1_, F2_, F3_, 2 * (s_) + 1, partn[2 * (cb2_) + 1])                             \
    } else {                                                                                   \
      CX_MFMA1(F1_, F2_, F3_, 2 * (s_) + 0, part[2 * (cb2_) + 0])                              \
      CX_MFMA1(F1_, F2_, F3_, 2 * (s_) + 1, part[2 * (cb2_) + 1])                              \
    }                                                                                          \
  }
  CX_READ(fa1, fa2, fa3, 0, 0);
  f32x4 part[4], partn[4];
  f32x4 vals[2][8];                              // vals[nb][b] = this layer's out[row(nb), 16 b + 4 kg .. + 3]
  for (int l = 0; l < d.n_layers; ++l) {
    const ChainLayer L = d.layer[l];
#pragma unroll
    for (int hc = 0; hc < 4; ++hc) {             // hc = 2 half + cbp: output columns 32 hc .. 32 hc + 31
#pragma unroll
      for (int i = 0; i < 4; ++i) { part[i] = f32x4{0.f, 0.f, 0.f, 0.f}; partn[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      // The epilogue's per-row operand of these 32 columns -- the saved activation for the derivative, the residual, or
      // the output that is accumulated into: at most one of them in the chains the library builds (pre[] below; a layer
      // with two of them loads the others in the epilogue as before) -- and the bias are requested HERE, two ring steps
      // before their use.  Loaded where they are used (rounds 3-5) every epilogue paid their full latency: the compiler's
      // wait for them is vmcnt(0), which also drains the LDS-DMA pieces in flight (it cannot see those), and the ring
      // steps behind it stalled in turn -- 0.46 of the chains' 1.22 ms per step (CX_ABL=1).
      float4 pre[4], pbias[2];
      const float* pre_src = L.dact ? L.dact : (L.resid ? L.resid : ((L.out && L.accumulate) ? L.out : nullptr));
      const long pre_ld = L.dact ? L.ld_dact : (L.resid ? L.ld_resid : L.ld_out);
      if (pre_src && !(CX_ABL & 1)) {
#pragma unroll
        for (int cb2 = 0; cb2 < 2; ++cb2) {
          const int col = 16 * (2 * hc + cb2) + 4 * kg;
          pre[2 * cb2 + 0] = *reinterpret_cast<const float4*>(pre_src + rca * pre_ld + col);
          pre[2 * cb2 + 1] = *reinterpret_cast<const float4*>(pre_src + rcb * pre_ld + col);
        }
      }
      if (L.bias && !(CX_ABL & 1)) {
#pragma unroll
        for (int cb2 = 0; cb2 < 2; ++cb2) pbias[cb2] = *reinterpret_cast<const float4*>(L.bias + 16 * (2 * hc + cb2) + 4 * kg);
      }
      __builtin_amdgcn_sched_barrier(0);         // the requests stay HERE
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {           // chunk (l, hc, kh) sits in ring slot (2 hc + kh) % 4
        const int slot = (2 * hc + kh) & 3;
        CX_TLOAD(l * 8 + hc * 2 + kh + 3);
        // four groups (s2, cb2), fragments one group ahead (sets A / B alternate)
        CX_READ(fb1, fb2, fb3, slot, 1);
        __builtin_amdgcn_sched_barrier(0);
        CX_MFMA(fa1, fa2, fa3, 2 * kh + 0, 0);
        CX_READ(fa1, fa2, fa3, slot, 2);
        __builtin_amdgcn_sched_barrier(0);
        CX_MFMA(fb1, fb2, fb3, 2 * kh + 0, 1);
        CX_READ(fb1, fb2, fb3, slot, 3);
        __builtin_amdgcn_sched_barrier(0);
        CX_MFMA(fa1, fa2, fa3, 2 * kh + 1, 0);
        CX_READ(fa1, fa2, fa3, (slot + 1) & 3, 0);
        __builtin_amdgcn_sched_barrier(0);
        CX_MFMA(fb1, fb2, fb3, 2 * kh + 1, 1);
        // chunk g + 2 (issued one chunk ago) must have landed; younger than it: this chunk's three loads.  The epilogue's
        // ordinary loads and stores are older than the next chunk's loads, so the counted wait can only be stricter
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      // ---- epilogue of the 32 columns ----
#pragma unroll
      for (int cb2 = 0; cb2 < 2; ++cb2) {
        const int b16 = 2 * hc + cb2;            // 16-column block of the layer's output
        const int col = 16 * b16 + 4 * kg;
        const f32x4 pa = part[2 * cb2 + 0] - partn[2 * cb2 + 0], pb = part[2 * cb2 + 1] - partn[2 * cb2 + 1];
        float4 va = make_float4(pa[0], pa[1], pa[2], pa[3]), vb = make_float4(pb[0], pb[1], pb[2], pb[3]);
        if (L.bias && !(CX_ABL & 1)) {
          const float4 b4 = pbias[cb2];
          va.x += b4.x; va.y += b4.y; va.z += b4.z; va.w += b4.w;
          vb.x += b4.x; vb.y += b4.y; vb.z += b4.z; vb.w += b4.w;
        }
        va = chain_act(va, L.act);
        vb = chain_act(vb, L.act);
        if (L.resid && !(CX_ABL & 1)) {
          const bool hoisted = !L.dact;            // (uniform) the residual is the hoisted operand unless a derivative is
          const float4 ra = hoisted ? pre[2 * cb2 + 0] : *reinterpret_cast<const float4*>(L.resid + rca * L.ld_resid + col);
          const float4 rb = hoisted ? pre[2 * cb2 + 1] : *reinterpret_cast<const float4*>(L.resid + rcb * L.ld_resid + col);
          va.x += ra.x; va.y += ra.y; va.z += ra.z; va.w += ra.w;
          vb.x += rb.x; vb.y += rb.y; vb.z += rb.z; vb.w += rb.w;
        }
        if (L.dact && !(CX_ABL & 1)) {
          va = chain_deriv(va, pre[2 * cb2 + 0], L.dact_type);
          vb = chain_deriv(vb, pre[2 * cb2 + 1], L.dact_type);
        }
        if (L.out) {
          float* oa = L.out + rca * L.ld_out + col;
          float* ob = L.out + rcb * L.ld_out + col;
          if (L.accumulate && !(CX_ABL & 1)) {
            const bool hoisted = !L.dact && !L.resid;
            const float4 ua = hoisted ? pre[2 * cb2 + 0] : *reinterpret_cast<const float4*>(oa);
            const float4 ub = hoisted ? pre[2 * cb2 + 1] : *reinterpret_cast<const float4*>(ob);
            if (row_a < rows) *reinterpret_cast<float4*>(oa) = make_float4(ua.x + va.x, ua.y + va.y, ua.z + va.z, ua.w + va.w);
            if (row_b < rows) *reinterpret_cast<float4*>(ob) = make_float4(ub.x + vb.x, ub.y + vb.y, ub.z + vb.z, ub.w + vb.w);
          } else if (!(CX_ABL & 2)) {
            if (row_a < rows) *reinterpret_cast<float4*>(oa) = va;
            if (row_b < rows) *reinterpret_cast<float4*>(ob) = vb;
          }
        }
        vals[0][b16] = f32x4{va.x, va.y, va.z, va.w};
        vals[1][b16] = f32x4{vb.x, vb.y, vb.z, vb.w};
      }
    }
    if (l + 1 < d.n_layers) {
      // ---- this layer's outputs become the next layer's rows: C layout -> B layout inside the four lanes of a row,
      // through a wave-private LDS tile, 32 columns at a time (see mlp_chain128_kernel; LDS serves a wave's accesses in
      // order, so write -> read -> overwrite needs no barrier) ----
      float* xw = xch + wave * (32 * XP);
#pragma unroll
      for (int sq = 0; sq < 4; ++sq) {           // columns 32 sq .. 32 sq + 31 = k-step sq of the next layer
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int c16 = 0; c16 < 2; ++c16) {
            const f32x4 t = vals[nb][2 * sq + c16];
            *reinterpret_cast<float4*>(xw + (n16 + 16 * nb) * XP + 16 * c16 + 4 * kg) = make_float4(t[0], t[1], t[2], t[3]);
          }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const float* rp = xw + (n16 + 16 * nb) * XP + 8 * kg;
          const float4 lo = *reinterpret_cast<const float4*>(rp), hi4 = *reinterpret_cast<const float4*>(rp + 4);
          const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
          const int qi = 2 * sq + nb;
          split3_x8(v, q1[qi], q2[qi], q3[qi]);
          if (sq & 1) { q1[qi] = neg_x8(q1[qi]); q2[qi] = neg_x8(q2[qi]); q3[qi] = neg_x8(q3[qi]); }   // odd k-step
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef CX_TLOAD
#undef CX_READ
#undef CX_MFMA1
#undef CX_MFMA
}

// floats of one prepared weight image in the current arithmetic mode (0: no fused chain in this mode)
size_t wprep_image_floats() {
  const int m = bilinear_mode();
  return m == 2 ? (size_t)WPREP_IMAGE_FLOATS : ((m == 4 || m == 6) ? (size_t)WPREP_IMAGE_FLOATS_X6 : 0);
}
int prepare_W_batch_launch(const WPrepBatch& b, float* dst, hipStream_t stream) {
  if (bilinear_mode() == 2) return prepare_W_f16_batch_launch(b, dst, stream);
  if (bilinear_mode() == 4 || bilinear_mode() == 6) return prepare_W_x6_batch_launch(b, dst, stream);
  return CGAT_ERR_UNSUPPORTED;
}

__global__ __launch_bounds__(256, 2) void mlp_chain128_x6_kernel(ChainDesc d) { mlp_chain128_x6_body(d); }
// several independent chains over the same row count in ONE launch (blockIdx.y = chain): the four predicted layers' trunks of
// a hypernetwork forward all read the hyper input.  At 83 340 rows a chain is 652 workgroups on 512 slots -- two rounds for
// 1.27 rounds of work; four chains together are six rounds instead of eight, and at 64 crystals one launch instead of four
struct ChainBatch { ChainDesc d[CHAIN_BATCH_MAX]; };
__global__ __launch_bounds__(256, 2) void mlp_chain128_x6_batch_kernel(ChainBatch b) { mlp_chain128_x6_body(b.d[blockIdx.y]); }

bool mlp_chain128_fast(const ChainDesc& d) {
  if (wprep_image_floats() == 0 || d.n_layers < 1 || d.n_layers > CHAIN_MAX) return false;
  uintptr_t bits = (uintptr_t)d.x | (uintptr_t)d.in_dact | (uintptr_t)d.in_store;
  long lds = d.ldx | d.ld_in_dact | d.ld_in_store;
  for (int l = 0; l < d.n_layers; ++l) {
    const ChainLayer& L = d.layer[l];
    bits |= (uintptr_t)L.W | (uintptr_t)L.bias | (uintptr_t)L.dact | (uintptr_t)L.resid | (uintptr_t)L.out;
    lds |= L.ld_dact | L.ld_resid | L.ld_out;
  }
  return (bits & 15) == 0 && (lds & 3) == 0;
}

int mlp_chain128_batch_launch(const ChainDesc* d, int n, hipStream_t stream) {
  if (n <= 0) return CGAT_OK;
  bool same = n <= CHAIN_BATCH_MAX && bilinear_mode() != 2;
  for (int i = 1; i < n; ++i) same = same && d[i].rows == d[0].rows;
  if (!same || n == 1) {
    for (int i = 0; i < n; ++i) CGAT_TRY(mlp_chain128_launch(d[i], stream));
    return CGAT_OK;
  }
  if (d[0].rows <= 0) return CGAT_OK;
  ChainBatch b;
  for (int i = 0; i < n; ++i) {
    CGAT_CHECK_ARG(mlp_chain128_fast(d[i]), "mlp_chain128: needs a split arithmetic mode, 1..%d layers and 16-byte aligned rows", CHAIN_MAX);
    b.d[i] = d[i];
  }
  for (int i = n; i < CHAIN_BATCH_MAX; ++i) b.d[i] = d[0];
  CGAT_PROF("mlp_chain", stream);
  hipLaunchKernelGGL(mlp_chain128_x6_batch_kernel, dim3(cdiv(d[0].rows, 128), n), dim3(256), 0, stream, b);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}
int mlp_chain128_launch(const ChainDesc& d, hipStream_t stream) {
  if (d.rows <= 0) return CGAT_OK;
  CGAT_CHECK_ARG(mlp_chain128_fast(d), "mlp_chain128: needs a split arithmetic mode, 1..%d layers and 16-byte aligned rows", CHAIN_MAX);
  CGAT_PROF("mlp_chain", stream);
  if (bilinear_mode() == 2) hipLaunchKernelGGL(mlp_chain128_kernel, dim3(cdiv(d.rows, 128)), dim3(256), 0, stream, d);
  else hipLaunchKernelGGL(mlp_chain128_x6_kernel, dim3(cdiv(d.rows, 128)), dim3(256), 0, stream, d);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}
