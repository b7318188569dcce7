// Small-row dense-layer programs: a whole G-row / Nc-row network (the output head ResidualNetwork, reference
// CGAT/message_changed.py:81-138; Roost's gate / message SimpleNetworks, CGAT/roost_message.py:88-156, 324-355; the
// per-crystal networks of MHAttention, CGAT/CGAT.py:14-62) forward or backward in ONE launch.
//
// At the batch the reference harness ships (64 crystals: 64 head rows, 200-450 composition rows, 1 280 atoms) every one
// of these layers is a few hundred kFLOP: what they cost is kernel boundaries (product + split-K reduction + bias sum +
// operand image per layer on the generic engine: 260 launches per step).  Here a network is a PROGRAM of products
//
//     out[m,n] = act( sum_k A'(m,k) B0(n,k) + bias[n] )  [-> h_out]  +  sum_k A(m,k) B1(n,k)  +  resid[m,n]   (+ out[m,n])
//     A'(m,k)  = A(m,k) * act'(dact(m,k))        (dact: saved activation values; none: A' = A)
//     rowsum[m] = sum_k A'(m,k)                                     (bias gradients)
//
// with arbitrary element strides on A, dact, B0, B1 (so the same op is a forward layer with its residual product, an
// input gradient g_pre W + g_y R, or a weight gradient g_pre^T x), grouped in PHASES: the ops of a phase are independent
// of each other, phase p reads what phases < p wrote.  One persistent launch walks the phases; between phases the
// workgroups meet at a grid barrier (one device-scope counter, release before the arrival, acquire after the wait:
// MI355X_MICROARCH.md, inter-workgroup visibility), so the hidden rows of a network never cross a kernel boundary.
//
// Arithmetic: fp32 operands converted (exactly) to fp64, products and sums on the fp64 matrix instruction
// (v_mfma_f64_16x16x4_f64), ONE rounding to fp32 per output -- a correctly rounded fp32 dot product up to the final
// activation: at least as accurate as the reference's fp32 sequence, and independent of the summation order to fp32
// resolution.  (An fp32-accumulating form -- v_mfma_f32_16x16x4_f32, a k-ordered fmaf chain -- runs at twice the rate but
// put 27 instead of 3 tensors of the cancellation-heavy sin-filled fixture `net_mean` beyond 1e-4 |ref|: the weight
// gradients of the gate networks are sums that cancel over every segment, and the six-pass split products these layers
// ran on before round 6 summed 32 exact products per rounding.)  No operand splitting, no prepared weight images: at
// <= 2048 rows the weights are read once, straight from HBM into registers (a 16 x 64 wave tile needs no LDS staging,
// cdna_hip_programming.md "M <= 16 decode weights" row).  Fixed task -> wave mapping: bitwise reproducible.
// What bounds a LARGE product here is the fp64 matrix rate itself: tools/mfma_f64_rate.hip measures 146 cycles per
// v_mfma_f64_16x16x4_f64 and SIMD = 34 TFLOP/s for the chip (the f32-input 16x16x4 form: 104-111), and a 1 280 x 128 x
// 2 560 product (0.84 GFLOP: the input gradient of a 5-head first layer at 64 crystals) takes 53 us = half of that
// whatever the load schedule (tools/rowprog_longk_probe.py: deeper k splits, batched requests: 48-65 us).  Programs are
// for products whose cost is their kernel boundaries; rowprog_gemm_ok() bounds rows and K accordingly.
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "../../include/cgat_hip.h"
#include "common.h"
#include "kernels.h"

typedef float v4f __attribute__((ext_vector_type(4)));
typedef double v4d __attribute__((ext_vector_type(4)));

#define RP_MAX_OPS CGAT_ROWPROG_MAX_OPS
#define RP_WG_THREADS 256
#define RP_MAX_WGS 128              // every workgroup must be resident for the grid barrier: 128 x 4 waves on 256 CUs
#define RP_BAR_SLOTS 1024           // barrier counters (one 64-byte line each); launch i uses slot i mod RP_BAR_SLOTS
#define RP_BAR_STRIDE 16            // unsigneds per slot
#define RP_SPIN_LIMIT (1u << 22)    // polls before a barrier gives up (~ seconds): sets the timeout word, never hangs

#define RP_VEC_A 1
#define RP_VEC_D 2
#define RP_VEC_B0 4
#define RP_VEC_B1 8

struct RowOp {
  const float* A;
  const float* dact;
  const float* B0;
  const float* B1;
  const float* bias;
  const float* resid;
  float* out;
  float* h_out;
  float* rowsum;
  int a_rs, a_ks, d_rs, d_ks, b0_rs, b0_ks, b1_rs, b1_ks, ld_resid, ldo, ld_h;
  int M, N, K;
  float alpha, beta;   // out = act(alpha * acc0 + bias) + acc1 + resid + beta * out
  unsigned char act, dact_type;
  unsigned char nb;      // 16-column tiles per wave tile (1, 2, 4)
  unsigned char ksplit;  // waves of a workgroup that share one tile, each taking every ksplit-th k group (1, 2, 4; nb == 1 if > 1)
  unsigned char vec;     // RP_VEC_*: operand rows are k-contiguous and 16-byte aligned -> dwordx4 loads
  unsigned char pad_[3];
  int tiles;             // wave tiles of the op
  int task_off;    // first WORKGROUP task of this op inside its phase (a workgroup task = 4 / ksplit wave tiles)
  int tiles_m;
};

struct RowProgK {
  int n_ops, n_phases;
  unsigned* bar;       // barrier counter of this launch (zero on entry, zero again on exit)
  unsigned* timeout;   // sticky give-up word of the device
  unsigned char phase_end[RP_MAX_OPS];   // one past the last op of phase p
  int phase_tasks[RP_MAX_OPS];
  RowOp op[RP_MAX_OPS];
};
static_assert(sizeof(RowProgK) <= 4096, "the program travels as a kernel argument");
static_assert(CGAT_ROWPROG_SYNC_WORDS == (RP_BAR_SLOTS + 1) * RP_BAR_STRIDE, "header and kernel disagree on the sync buffer");

__device__ __forceinline__ float rp_act(float v, int act) {
  switch (act) {
    case CGAT_ACT_TANH: return tanhf(v);
    case CGAT_ACT_LEAKY: return v > 0.f ? v : 0.01f * v;
    case CGAT_ACT_RELU: return v > 0.f ? v : 0.f;
    default: return v;
  }
}
// g * act'(.) in terms of the saved activation VALUE y (LeakyReLU / ReLU: sign(y) = sign(pre); tanh: 1 - y^2)
__device__ __forceinline__ float rp_dact(float g, float y, int type) {
  switch (type) {
    case CGAT_ACT_TANH: return g * (1.f - y * y);
    case CGAT_ACT_LEAKY: return y > 0.f ? g : 0.01f * g;
    case CGAT_ACT_RELU: return y > 0.f ? g : 0.f;
    default: return g;
  }
}

// elements k0 .. k0+3 of a row (k0 a multiple of 4), element stride ks; FULL: all four are below K
template <bool FULL>
__device__ __forceinline__ v4f rp_ld4(const float* __restrict__ row, int ks, bool vec, int k0, int K) {
  v4f r;
  if (FULL) {
    if (vec) return *reinterpret_cast<const v4f*>(row + k0);
    const float* p = row + (long)k0 * ks;
    r.x = p[0]; r.y = p[ks]; r.z = p[2 * (long)ks]; r.w = p[3 * (long)ks];
    return r;
  }
  const float* p = row + (long)k0 * ks;
  r.x = k0 < K ? p[0] : 0.f;
  r.y = k0 + 1 < K ? p[ks] : 0.f;
  r.z = k0 + 2 < K ? p[2 * (long)ks] : 0.f;
  r.w = k0 + 3 < K ? p[3 * (long)ks] : 0.f;
  return r;
}

// Operand registers of one k group (64 / NB k values: NF dwordx4 per operand row).  EXT: the op has a derivative
// operand and / or a second product; plain products carry neither and afford a deeper ring.
template <int NB, bool EXT>
struct RpFrag {
  static constexpr int NF = 4 / NB;
  v4f a[NF], w0[NB][NF];
  v4f d[EXT ? NF : 1], w1[EXT ? NB : 1][EXT ? NF : 1];
};

// One wave tile = 16 x (16 NB) outputs.  MFMA 16x16x4 f64: lane l supplies A(row l%16, k = l/16) and B(col l%16,
// k = l/16), one double each; a lane's dwordx4 along k (k = 4 (l/16) + j) feeds four instructions, both operands alike
// (fp32 -> fp64 conversion is exact).
// ksplit > 1 (NB == 1): the tile's k groups are dealt round-robin to `ksplit` waves of the workgroup (kpart = this
// wave's share), partial sums meet in LDS in wave order and the first wave of the set finishes the tile -- a layer
// with few output tiles and a long K (the head's 1024-wide layers at 64 rows) then keeps every wave busy with a
// quarter of the chain.  `active` = this wave has a tile (all waves of a workgroup pass the same barriers).
// The loads of DEPTH - 1 groups are in flight behind the group being multiplied: a group is 16 (32) matrix
// instructions = 0.45 (0.9) us, an L2 / HBM round trip 1-2 us, and a wave is alone on its SIMD.
template <int NB, bool EXT>
__device__ __forceinline__ void rp_tile(const RowOp& op, int task, int lane, int kpart, int ksplit, bool active,
                                        double* __restrict__ red, int wave) {
  typedef RpFrag<NB, EXT> Frag;
  constexpr int NF = Frag::NF;
  constexpr int GK = 16 * NF;
  constexpr int DEPTH = EXT ? 3 : 5;
  const int i = lane & 15, q = lane >> 4;
  const int tiles_m = op.tiles_m;
  const int nt = task / tiles_m, mt = task - nt * tiles_m;
  const int m0 = mt * 16, n0 = nt * 16 * NB;
  const int M = op.M, N = op.N, K = op.K;
  const int arow = min(m0 + i, M - 1);
  const float* __restrict__ Ap = op.A + (long)arow * op.a_rs;
  const bool has_d = EXT && op.dact != nullptr, two = EXT && op.B1 != nullptr;
  const float* __restrict__ Dp = has_d ? op.dact + (long)arow * op.d_rs : Ap;
  const float* __restrict__ Bp0[NB];
  const float* __restrict__ Bp1[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int brow = min(n0 + 16 * b + i, N - 1);
    Bp0[b] = op.B0 + (long)brow * op.b0_rs;
    Bp1[b] = two ? op.B1 + (long)brow * op.b1_rs : Bp0[b];
  }
  const int a_ks = op.a_ks, d_ks = op.d_ks, b0_ks = op.b0_ks, b1_ks = op.b1_ks;
  const bool va = op.vec & RP_VEC_A, vd = op.vec & RP_VEC_D, vb0 = op.vec & RP_VEC_B0, vb1 = op.vec & RP_VEC_B1;
  const int dtype = op.dact_type;

  v4d acc0[NB], acc1[EXT ? NB : 1];
#pragma unroll
  for (int b = 0; b < NB; ++b) acc0[b] = v4d{0., 0., 0., 0.};
#pragma unroll
  for (int b = 0; b < (EXT ? NB : 1); ++b) acc1[b] = v4d{0., 0., 0., 0.};
  double rs = 0.;

  auto load = [&](Frag& f, int g) {
    const int kb = g * GK + 4 * q;
    if (g * GK + GK <= K) {
#pragma unroll
      for (int t = 0; t < NF; ++t) {
        const int k0 = kb + 16 * t;
        f.a[t] = rp_ld4<true>(Ap, a_ks, va, k0, K);
        if (EXT && has_d) f.d[t] = rp_ld4<true>(Dp, d_ks, vd, k0, K);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          f.w0[b][t] = rp_ld4<true>(Bp0[b], b0_ks, vb0, k0, K);
          if (EXT && two) f.w1[b][t] = rp_ld4<true>(Bp1[b], b1_ks, vb1, k0, K);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < NF; ++t) {
        const int k0 = kb + 16 * t;
        f.a[t] = rp_ld4<false>(Ap, a_ks, false, k0, K);
        if (EXT && has_d) f.d[t] = rp_ld4<false>(Dp, d_ks, false, k0, K);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          f.w0[b][t] = rp_ld4<false>(Bp0[b], b0_ks, false, k0, K);
          if (EXT && two) f.w1[b][t] = rp_ld4<false>(Bp1[b], b1_ks, false, k0, K);
        }
      }
    }
  };
  auto compute = [&](const Frag& f) {
#pragma unroll
    for (int t = 0; t < NF; ++t) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float av = f.a[t][j];
        const float at = (EXT && has_d) ? rp_dact(av, f.d[t][j], dtype) : av;
        const double ad = (double)at;
        rs += ad;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          acc0[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad, (double)f.w0[b][t][j], acc0[b], 0, 0, 0);
          if (EXT && two) acc1[b] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av, (double)f.w1[b][t][j], acc1[b], 0, 0, 0);
        }
      }
    }
  };

  // this wave's k groups: kpart, kpart + ksplit, ...; the loads of the next DEPTH - 1 groups are issued before the one
  // being multiplied.  What the ISA shows (round 6): the loads sit under uniform conditions, so hipcc waits vmcnt(0)
  // before every group's first matrix instruction -- the groups requested together share ONE round trip, the steady state
  // pays one per group.  Measured alternatives, none better at these sizes: unconditional requests with clamped indices
  // (counted waits, but a K = 128 product re-requests its last group 7 times: single products 8 -> 22 us); a condition-
  // free steady loop with a conditional drain (300-600 spilled registers across the inlined variants); batches of DEPTH
  // groups requested, awaited and multiplied together (the same times within 5 %).
  const int ng_all = (K + GK - 1) / GK;
  const int ng = active ? (ng_all - kpart + ksplit - 1) / ksplit : 0;
  auto gidx = [&](int g) { return kpart + g * ksplit; };
  Frag ring[DEPTH];
#pragma unroll
  for (int s = 0; s < DEPTH - 1; ++s)
    if (s < ng) load(ring[s], gidx(s));
  for (int g = 0; g < ng; g += DEPTH) {
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
      if (g + s < ng) {
        if (g + s + DEPTH - 1 < ng) load(ring[(s + DEPTH - 1) % DEPTH], gidx(g + s + DEPTH - 1));
        compute(ring[s]);
      }
    }
  }
  if (NB == 1 && ksplit > 1) {
    // partial sums of the set's other waves -> LDS; the set's first wave adds them in wave order
    double* mine = red + (size_t)wave * (9 * 64);
    if (kpart > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        mine[r * 64 + lane] = acc0[0][r];
        if (EXT) mine[(4 + r) * 64 + lane] = acc1[0][r];
      }
      mine[8 * 64 + lane] = rs;
    }
    __syncthreads();
    if (kpart == 0) {
      for (int w = 1; w < ksplit; ++w) {
        const double* other = red + (size_t)(wave + w) * (9 * 64);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc0[0][r] += other[r * 64 + lane];
          if (EXT) acc1[0][r] += other[(4 + r) * 64 + lane];
        }
        rs += other[8 * 64 + lane];
      }
    }
    __syncthreads();          // the slots are free again for the workgroup's next task
    if (kpart > 0) return;
  }
  if (!active) return;

  // epilogue.  f64 16x16x4 C/D map: lane holds column n0 + 16 b + i of rows m0 + q + 4 r (r = 0..3); the fp64 sums
  // are rounded to fp32 ONCE, after alpha and the bias
  const int act = op.act;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int n = n0 + 16 * b + i;
    if (n >= N) continue;
    const double bv = op.bias ? (double)op.bias[n] : 0.;
    const double alpha = (double)op.alpha;
    const float beta = op.beta;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + q + 4 * r;
      if (m >= M) continue;
      float v = rp_act((float)(alpha * acc0[b][r] + bv), act);
      if (op.h_out) op.h_out[(long)m * op.ld_h + n] = v;
      float* o = op.out + (long)m * op.ldo + n;
      if (two || op.resid || beta != 0.f) {
        double w = (double)v;
        if (EXT && two) w += acc1[b][r];
        if (op.resid) w += (double)op.resid[(long)m * op.ld_resid + n];
        if (beta != 0.f) w += (double)beta * (double)*o;
        v = (float)w;
      }
      *o = v;
    }
  }
  if (op.rowsum && nt == 0) {
    rs += __shfl_xor(rs, 16, 64);
    rs += __shfl_xor(rs, 32, 64);
    if (q == 0 && m0 + i < M) op.rowsum[m0 + i] = (float)rs;
  }
}

// All workgroups of the launch; `target` arrivals complete the barrier.  Producer side: every wave drains its stores,
// the workgroup meets, one lane releases (L2 write-back) and arrives; consumer side: the same lane polls the counter
// (L1-bypassing load), acquires (L1 invalidate), and the workgroup meets again before anyone loads.
__device__ __forceinline__ void rp_grid_barrier(unsigned* ctr, unsigned target, unsigned* timeout) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins >= RP_SPIN_LIMIT ||
          ((spins & 1023u) == 0 && __hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
        __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // results are void; no hang
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

__global__ __launch_bounds__(RP_WG_THREADS) void rowprog_kernel(const RowProgK P) {
  __shared__ double red[(RP_WG_THREADS / 64) * 9 * 64];      // k-split partial sums: 4 waves x (2 x 4 + 1) x 64 lanes
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int op_lo = 0;
  for (int ph = 0; ph < P.n_phases; ++ph) {
    const int op_hi = P.phase_end[ph];
    const int ntasks = P.phase_tasks[ph];
    for (int t = blockIdx.x; t < ntasks; t += gridDim.x) {       // workgroup tasks: uniform over the workgroup
      int o = op_lo;
      while (o + 1 < op_hi && t >= P.op[o + 1].task_off) ++o;
      const RowOp& op = P.op[o];
      const int ks = op.ksplit;
      const int tile = (t - op.task_off) * ((RP_WG_THREADS / 64) / ks) + wave / ks;
      const bool active = tile < op.tiles;
      const bool ext = op.dact != nullptr || op.B1 != nullptr;
      switch (op.nb) {
        case 4:
          if (active) { if (ext) rp_tile<4, true>(op, tile, lane, 0, 1, true, red, wave); else rp_tile<4, false>(op, tile, lane, 0, 1, true, red, wave); }
          break;
        case 2:
          if (active) { if (ext) rp_tile<2, true>(op, tile, lane, 0, 1, true, red, wave); else rp_tile<2, false>(op, tile, lane, 0, 1, true, red, wave); }
          break;
        default:
          if (ext) rp_tile<1, true>(op, active ? tile : 0, lane, wave % ks, ks, active, red, wave);
          else rp_tile<1, false>(op, active ? tile : 0, lane, wave % ks, ks, active, red, wave);
          break;
      }
    }
    op_lo = op_hi;
    if (ph + 1 < P.n_phases) rp_grid_barrier(P.bar, (unsigned)(ph + 1) * gridDim.x, P.timeout);
  }
  if (P.n_phases > 1 && threadIdx.x == 0) {
    // leave the counter at zero for the slot's next user: the workgroup whose exit arrival is the last one resets it
    const unsigned total = (unsigned)P.n_phases * gridDim.x;
    if (__hip_atomic_fetch_add(P.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1u)
      __hip_atomic_store(P.bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- host side ----
static std::mutex g_rp_mu;
static unsigned g_rp_next = 0;

static bool rp_fits_int(int64_t v) { return v >= -(1ll << 30) && v <= (1ll << 30); }
static bool rp_vec_ok(const float* p, int64_t rs, int64_t ks) {
  return p && ks == 1 && rs % 4 == 0 && ((uintptr_t)p & 15) == 0;
}

// sync_words may be null for a single-phase program (no barrier)
int rowprog_launch(const cgat_rowprog_op* ops, int n_ops, uint32_t* sync_words, hipStream_t s) {
  CGAT_CHECK_ARG(ops && n_ops >= 1 && n_ops <= RP_MAX_OPS, "rowprog: 1..%d ops", RP_MAX_OPS);
  RowProgK K;
  memset(&K, 0, sizeof(K));
  K.n_ops = n_ops;
  bool multi = false;
  for (int o = 1; o < n_ops; ++o) multi = multi || ops[o].phase != ops[0].phase;
  // waves the program can keep busy: the resident grid of a program with barriers; two rounds of one wave per SIMD otherwise
  const int total_waves = multi ? RP_MAX_WGS * (RP_WG_THREADS / 64) : 2048;
  int phase = -1, n_phases = 0, max_tasks = 0;
  for (int o = 0; o < n_ops; ++o) {
    const cgat_rowprog_op& in = ops[o];
    RowOp& op = K.op[o];
    CGAT_CHECK_ARG(in.M >= 1 && in.N >= 1 && in.K >= 0, "rowprog op %d: empty shape %d x %d x %d", o, in.M, in.N, in.K);
    CGAT_CHECK_ARG(in.A && in.B0 && in.out, "rowprog op %d: A, B0 and out are required", o);
    CGAT_CHECK_ARG(in.phase >= 0 && in.phase >= phase && in.phase <= phase + 1,
                   "rowprog op %d: phases must be non-decreasing without gaps (got %d after %d)", o, in.phase, phase);
    CGAT_CHECK_ARG(rp_fits_int(in.a_rs) && rp_fits_int(in.a_ks) && rp_fits_int(in.d_rs) && rp_fits_int(in.d_ks) &&
                       rp_fits_int(in.b0_rs) && rp_fits_int(in.b0_ks) && rp_fits_int(in.b1_rs) && rp_fits_int(in.b1_ks) &&
                       rp_fits_int(in.ld_resid) && rp_fits_int(in.ldo) && rp_fits_int(in.ld_h),
                   "rowprog op %d: strides beyond 2^30 elements", o);
    CGAT_CHECK_ARG(in.act >= 0 && in.act <= 3 && in.dact_type >= 0 && in.dact_type <= 3, "rowprog op %d: bad activation", o);
    op.A = in.A; op.dact = in.dact; op.B0 = in.B0; op.B1 = in.B1; op.bias = in.bias; op.resid = in.resid;
    op.out = in.out; op.h_out = in.h_out; op.rowsum = in.rowsum;
    op.a_rs = (int)in.a_rs; op.a_ks = (int)in.a_ks; op.d_rs = (int)in.d_rs; op.d_ks = (int)in.d_ks;
    op.b0_rs = (int)in.b0_rs; op.b0_ks = (int)in.b0_ks; op.b1_rs = (int)in.b1_rs; op.b1_ks = (int)in.b1_ks;
    op.ld_resid = (int)in.ld_resid; op.ldo = (int)in.ldo; op.ld_h = (int)in.ld_h;
    op.M = in.M; op.N = in.N; op.K = in.K;
    op.act = (unsigned char)in.act; op.dact_type = (unsigned char)(in.dact ? in.dact_type : 0);
    op.alpha = in.alpha; op.beta = in.beta;
    op.vec = (unsigned char)((rp_vec_ok(in.A, in.a_rs, in.a_ks) ? RP_VEC_A : 0) | (rp_vec_ok(in.dact, in.d_rs, in.d_ks) ? RP_VEC_D : 0) |
             (rp_vec_ok(in.B0, in.b0_rs, in.b0_ks) ? RP_VEC_B0 : 0) | (rp_vec_ok(in.B1, in.b1_rs, in.b1_ks) ? RP_VEC_B1 : 0));
    op.tiles_m = cdiv(in.M, 16);
    // the widest tile that still gives every wave of the launch a tile (A fragments are shared by a tile's column
    // blocks); with fewer 16 x 16 tiles than waves, split the k range over 2 or 4 waves while a share keeps >= 64 k
    const int tn = cdiv(in.N, 16);
    int nb = 4;
    while (nb > 1 && (long)op.tiles_m * cdiv(tn, nb) < total_waves) nb >>= 1;
    int ks = 1;
    if (nb == 1)
      while (ks < 4 && (long)op.tiles_m * tn * ks * 2 <= total_waves && in.K >= 64 * ks * 2) ks <<= 1;
    op.nb = (unsigned char)nb;
    op.ksplit = (unsigned char)ks;
    op.tiles = op.tiles_m * cdiv(tn, nb);
    if (in.phase != phase) {
      phase = in.phase;
      ++n_phases;
    }
    op.task_off = K.phase_tasks[n_phases - 1];
    K.phase_tasks[n_phases - 1] += cdiv(op.tiles, (RP_WG_THREADS / 64) / ks);
    K.phase_end[n_phases - 1] = (unsigned char)(o + 1);
  }
  K.n_phases = n_phases;
  for (int p = 0; p < n_phases; ++p) max_tasks = K.phase_tasks[p] > max_tasks ? K.phase_tasks[p] : max_tasks;
  // a program with barriers needs every workgroup resident; a single phase takes as many workgroups as it has tasks
  int wgs = max_tasks;
  const int cap = n_phases > 1 ? RP_MAX_WGS : 4096;
  if (wgs > cap) wgs = cap;
  if (n_phases > 1) {
    CGAT_CHECK_ARG(sync_words && ((uintptr_t)sync_words & 63) == 0,
                   "rowprog: a program with more than one phase needs sync_words (64-byte aligned device memory)");
    std::lock_guard<std::mutex> lk(g_rp_mu);
    K.bar = sync_words + (size_t)(g_rp_next++ % RP_BAR_SLOTS) * RP_BAR_STRIDE;
    K.timeout = sync_words + (size_t)RP_BAR_SLOTS * RP_BAR_STRIDE;
  }
  CGAT_PROF("rowprog", s);
  hipLaunchKernelGGL(rowprog_kernel, dim3(wgs), dim3(RP_WG_THREADS), 0, s, K);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

extern "C" int cgat_rowprog_run(const cgat_rowprog* prog, uint32_t* sync_words, void* stream) {
  CGAT_CHECK_ARG(prog, "rowprog: null program");
  return rowprog_launch(prog->op, prog->n_ops, sync_words, (hipStream_t)stream);
}

// ---- the layer orchestrators' small-row products (layers.hip, Ctx::gemm) as single-op programs ----
static int g_rp_max_rows = -1;
int rowprog_max_rows() {
  if (g_rp_max_rows < 0) {
    const char* e = getenv("CGAT_ROWPROG_MAX_ROWS");
    g_rp_max_rows = e ? atoi(e) : 2048;
  }
  return g_rp_max_rows;
}
// (K bounded too: a reduction over the BATCH -- a bias or weight gradient, K = 83 340 atoms with M = 3 outputs -- is 8 wave
// tiles walking the whole batch: 0.72 ms in the 1 M-edge layer step where the split-K engine takes 0.05; found in the
// round's own kernel statistics)
bool rowprog_gemm_ok(const GemmParams& p) {
  return p.M >= 1 && p.N >= 1 && p.K >= 1 && p.M <= rowprog_max_rows() && p.N <= 4096 && p.K <= 2 * rowprog_max_rows() &&
         !p.a_rgather && !p.a_block && !p.b_kgather && !p.c_scatter && !p.add1 && !p.add2 && !p.a_outer && !p.b_outer;
}
void rowprog_op_from_gemm(const GemmParams& p, cgat_rowprog_op* o) {
  memset(o, 0, sizeof(*o));
  o->M = p.M; o->N = p.N; o->K = p.K;
  o->A = p.A;
  if (p.a_kmajor) { o->a_rs = 1; o->a_ks = p.lda; } else { o->a_rs = p.lda; o->a_ks = 1; }
  o->B0 = p.B;
  if (p.b_kmajor) { o->b0_rs = 1; o->b0_ks = p.ldb; } else { o->b0_rs = p.ldb; o->b0_ks = 1; }
  o->bias = p.bias; o->act = p.act;
  o->out = p.C; o->ldo = p.ldc;
  o->alpha = p.alpha; o->beta = p.beta;
}
int rowprog_gemm(const GemmParams& p, hipStream_t s) {
  cgat_rowprog_op o;
  rowprog_op_from_gemm(p, &o);
  return rowprog_launch(&o, 1, nullptr, s);
}
