// Public C ABI glue: error string, launch-timing registry, and the thin entry points that map
// one-to-one onto a kernel family (dense layer, segment softmax / sum, CSR plan).
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/cgat_hip.h"
#include "common.h"
#include "kernels.h"

// ---- error ----
static thread_local char g_err[1024] = "";
void cgat_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* cgat_last_error(void) { return g_err; }
extern "C" int cgat_abi_version(void) { return CGAT_ABI_VERSION; }

// ---- launch timing ----
struct ProfRec {
  std::string tag;
  hipEvent_t beg, end;
};
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
static bool g_prof_on = false;

ProfScope::ProfScope(const char* tag, hipStream_t s) : slot(-1), stream(s) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r;
  r.tag = tag;
  if (hipEventCreate(&r.beg) != hipSuccess) return;
  if (hipEventCreate(&r.end) != hipSuccess) { (void)hipEventDestroy(r.beg); return; }
  (void)hipEventRecord(r.beg, s);
  g_prof.push_back(r);
  slot = (int)g_prof.size() - 1;
}
ProfScope::~ProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (slot < (int)g_prof.size()) (void)hipEventRecord(g_prof[slot].end, stream);
}
unsigned long long g_cgat_launches = 0;
extern "C" uint64_t cgat_prof_launches(void) { return __atomic_load_n(&g_cgat_launches, __ATOMIC_RELAXED); }
extern "C" void cgat_prof_enable(int on) { g_prof_on = on != 0; }
extern "C" void cgat_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) {
    (void)hipEventDestroy(r.beg);
    (void)hipEventDestroy(r.end);
  }
  g_prof.clear();
}
extern "C" int cgat_prof_get(const char* tag, int* count, float* total_ms) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  int n = 0;
  float tot = 0.f;
  for (auto& r : g_prof) {
    if (r.tag != tag) continue;
    if (hipEventSynchronize(r.end) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.beg, r.end) == hipSuccess) {
      tot += ms;
      ++n;
    }
  }
  if (count) *count = n;
  if (total_ms) *total_ms = tot;
  return CGAT_OK;
}

// ---- CSR plan ----
size_t csr_ws_bytes(int S);
extern "C" size_t cgat_plan_workspace_bytes(int32_t E, int32_t N) { return plan_ws_bytes(E, N) + 256; }
extern "C" int cgat_plan_build(const int64_t* edge_index, int32_t E, int32_t N, int32_t* dst_rowptr, int32_t* dst_perm,
                               int32_t* dst_sorted, int32_t* src_sorted, int32_t* src_rowptr, int32_t* src_pos,
                               void* ws, size_t ws_bytes, void* stream) {
  CGAT_CHECK_ARG(E >= 0 && N >= 0, "plan_build: negative size");
  return plan_build_launch(edge_index, E, N, dst_rowptr, dst_perm, dst_sorted, src_sorted, src_rowptr, src_pos, ws,
                           ws_bytes, (hipStream_t)stream);
}
extern "C" size_t cgat_csr_workspace_bytes(int32_t S) { return csr_ws_bytes(S) + 256; }
extern "C" int cgat_csr_from_keys(const int32_t* keys, int32_t n, int32_t S, int32_t* rowptr, int32_t* perm, void* ws,
                                  size_t ws_bytes, void* stream) {
  return csr_from_keys_launch(keys, n, S, rowptr, perm, ws, ws_bytes, (hipStream_t)stream);
}

// ---- dense layer ----
// split-bf16 routes of the dense layer (default arithmetic mode): K == 128 -> any multiple of 128 outputs on the
// edge_z kernel (rows split once, weights through the LDS-DMA ring); K a multiple of 128 -> 128 outputs on the edge_ge
// kernel (rows split in the loop).  Everything else, and the f32 mode, runs on the generic f32 GEMM engine.
static bool linear_route_z(int K, int N, int act, long ldx, long ldy, const void* x, const void* y) {
  return (act == CGAT_ACT_NONE || act == CGAT_ACT_TANH || act == CGAT_ACT_LEAKY) && linear128_fast(K, N, ldx, ldy, x, y);
}
static bool linear_route_ge(int K, int N, int act, long ldx, long ldy, const void* x, const void* y) {
  return act == CGAT_ACT_NONE && N == 128 && K > 128 && edge_ge_fast(128, K, ldx, 128, ldy, x, y);
}
extern "C" size_t cgat_linear_forward_workspace_bytes(int32_t M, int32_t K, int32_t N) {
  (void)M;
  const size_t a = (K == 128 && N % 128 == 0) ? linear128_ws_bytes(N) : 0;
  const size_t b = (N == 128 && K % 128 == 0) ? edge_z_wq_floats(K) * sizeof(float) : 0;
  const int sk = gemm_pick_splits_skinny(M, N, K);
  const size_t c = sk > 1 ? ws_round((size_t)sk * M * N, 4) : 0;
  return (a > b ? (a > c ? a : c) : (b > c ? b : c)) + 256;
}
extern "C" int cgat_linear_forward(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias,
                                   float* y, int64_t ldy, int32_t M, int32_t K, int32_t N, int32_t act,
                                   const float* x_absmax, void* ws, size_t ws_bytes, void* stream) {
  CGAT_CHECK_ARG(M >= 0 && K >= 0 && N >= 0, "linear_forward: negative size");
  hipStream_t s = (hipStream_t)stream;
  const bool have_ws = ws && ws_bytes >= cgat_linear_forward_workspace_bytes(M, K, N);
  if (have_ws && (!bias || (((uintptr_t)bias) & 15) == 0)) {
    if (linear_route_z(K, N, act, ldx, ldy, x, y))
      return linear128_launch(x, ldx, w, ldw, 1, bias, act, 0, y, ldy, M, ws, s, N);
    if (linear_route_ge(K, N, act, ldx, ldy, x, y))
      return edge_ge_launch(x, ldx, 128, w, 1, ldw, (float*)ws, K, y, ldy, nullptr, M, 0, bias, s, x_absmax);
  }
  GemmParams g = gemm_params(M, N, K, x, ldx, w, ldw, y, ldy);
  g.bias = bias;
  g.act = act;
  if (have_ws) g.splits = gemm_pick_splits_skinny(M, N, K);
  return gemm_launch(g, g.splits > 1 ? ws : nullptr, g.splits > 1 ? ws_bytes : 0, s);
}
extern "C" size_t cgat_linear_backward_workspace_bytes(int32_t M, int32_t K, int32_t N) {
  int splits = gemm_pick_splits(N, K, M);
  size_t a = splits > 1 ? ws_round((size_t)splits * N * K, 4) : 0;
  size_t b = colsum_ws_bytes(M, N);
  // g_x = gpre W on the split-bf16 kernels: N == 128 inputs -> K outputs (edge_z route) or N > 128 -> 128 outputs
  size_t c = (N == 128 && K % 128 == 0) ? linear128_ws_bytes(K) : 0;
  if (K == 128 && N % 128 == 0 && edge_z_wq_floats(N) * sizeof(float) > c) c = edge_z_wq_floats(N) * sizeof(float);
  if (b > a) a = b;
  if (c > a) a = c;
  {  // g_x on the generic kernel with few output tiles: split over N
    const int sk = gemm_pick_splits_skinny(M, K, N);
    if (sk > 1 && ws_round((size_t)sk * M * K, 4) > a) a = ws_round((size_t)sk * M * K, 4);
  }
  if (K == 128 && N == 128 && rows_dw128_ws_bytes(M, 1) > a) a = rows_dw128_ws_bytes(M, 1);
  if (K % 128 == 0 && N % 128 == 0 && (K / 128) * (N / 128) <= DW_BATCH_MAX &&
      rows_dw128_batch_ws_bytes((K / 128) * (N / 128), M) > a)
    a = rows_dw128_batch_ws_bytes((K / 128) * (N / 128), M);
  return a + 256;
}

static int linear_backward_impl(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* y,
                                int64_t ldy, const float* g_y, int64_t ldgy, float* gpre, float* g_x,
                                int64_t ldgx, int32_t accumulate_gx, float* g_w, int64_t ldgw, float* g_b,
                                int32_t M, int32_t K, int32_t N, int32_t act, void* ws, size_t ws_bytes,
                                void* stream, const float* gx_dact, int64_t ld_dact, float* gx_absmax);
extern "C" int cgat_linear_backward(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* y,
                                    int64_t ldy, const float* g_y, int64_t ldgy, float* gpre, float* g_x,
                                    int64_t ldgx, int32_t accumulate_gx, float* g_w, int64_t ldgw, float* g_b,
                                    int32_t M, int32_t K, int32_t N, int32_t act, void* ws, size_t ws_bytes,
                                    void* stream) {
  return linear_backward_impl(x, ldx, w, ldw, y, ldy, g_y, ldgy, gpre, g_x, ldgx, accumulate_gx, g_w, ldgw, g_b, M, K, N, act,
                              ws, ws_bytes, stream, nullptr, 0, nullptr);
}
// The same with the input gradient multiplied by LeakyReLU'(0.01) at the sign of gx_dact[M, K] (row stride ld_dact) in
// the product's epilogue, and max |g_x| folded into gx_absmax[0] (caller-zeroed): when x IS a LeakyReLU output (the hidden
// layer of MultiHeadNetwork, CGAT.py:96) the result is the gradient of its PRE-activation -- what autograd computes with a
// separate elementwise pass over [M, K] -- ready for cgat_edge_hidden_backward(g_is_pre = 1).  Supported where g_x runs
// on the dense-layer kernel (N == 128, K a multiple of 128, split arithmetic modes); CGAT_ERR_UNSUPPORTED otherwise.
extern "C" int cgat_linear_backward_dact(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* g_y,
                                         int64_t ldgy, float* g_x, int64_t ldgx, const float* gx_dact, int64_t ld_dact,
                                         float* gx_absmax, float* g_w, int64_t ldgw, float* g_b, int32_t M, int32_t K,
                                         int32_t N, void* ws, size_t ws_bytes, void* stream) {
  CGAT_CHECK_ARG(g_x && gx_dact, "linear_backward_dact: g_x and gx_dact are required");
  if (!(ws && ws_bytes >= cgat_linear_backward_workspace_bytes(M, K, N) && N == 128 &&
        linear128_fast(N, K, ldgy, ldgx, g_y, g_x) && (ld_dact % 4) == 0 && (((uintptr_t)gx_dact) & 15) == 0)) {
    cgat_set_error("linear_backward_dact: needs the dense-layer route (N == 128, K %% 128 == 0, aligned operands, split mode)");
    return CGAT_ERR_UNSUPPORTED;
  }
  return linear_backward_impl(x, ldx, w, ldw, nullptr, N, g_y, ldgy, nullptr, g_x, ldgx, 0, g_w, ldgw, g_b, M, K, N,
                              CGAT_ACT_NONE, ws, ws_bytes, stream, gx_dact, ld_dact, gx_absmax);
}
// ---- the H per-head second layers of one MultiHeadNetwork as ONE call (reference CGAT/CGAT.py:97-98, 103-109: fc_out
// is a grouped Conv1d, i.e. H independent Linear(Hd, Co) on the H column blocks of the hidden matrix).  Head h works on
// x + h * s_x, w + h * s_w, bias + h * s_bias, y + h * s_y (strides in elements): with the operands of a head an affine
// function of h, the whole layer is three launches (batched weight maxima + planes, one product launch with grid.y = h)
// instead of 4-5 per head -- at the harness' shipped batch (15-30 k edge rows) those launches are 20-40 us each and
// fill a fraction of the chip.  Shapes the batched kernels do not take run head by head through cgat_linear_forward.
static bool heads_strides_ok(int64_t a, int64_t b, int64_t c, int64_t d) { return ((a | b | c | d) & 3) == 0; }
extern "C" size_t cgat_heads_linear_forward_workspace_bytes(int32_t M, int32_t K, int32_t N, int32_t H) {
  size_t a = cgat_linear_forward_workspace_bytes(M, K, N);
  if (N == 128 && K % 128 == 0 && H >= 1 && H <= TPREP_MAX) {
    const size_t b = ((size_t)H * edge_ge_heads_image_floats(K) + bilinear_prepare_T_batch_ws_floats(H)) * sizeof(float) + 256;
    if (b > a) a = b;
  }
  return a;
}
extern "C" int cgat_heads_linear_forward(const float* x, int64_t ldx, int64_t s_x, const float* w, int64_t ldw, int64_t s_w,
                                         const float* bias, int64_t s_bias, float* y, int64_t ldy, int64_t s_y, int32_t M,
                                         int32_t K, int32_t N, int32_t H, const float* x_absmax, void* ws, size_t ws_bytes,
                                         void* stream) {
  CGAT_CHECK_ARG(M >= 0 && K >= 0 && N >= 0 && H >= 0, "heads_linear_forward: negative size");
  hipStream_t s = (hipStream_t)stream;
  const bool have_ws = ws && ws_bytes >= cgat_heads_linear_forward_workspace_bytes(M, K, N, H);
  if (have_ws && H > 1 && N == 128 && K > 128 && heads_strides_ok(s_x, s_w, s_bias, s_y) &&
      (!bias || (((uintptr_t)bias) & 15) == 0) && edge_ge_heads_fast(H, K, ldx, ldy, ldw, x, w, y, x_absmax))
    return edge_ge_heads_launch(H, x, ldx, s_x, w, s_w, bias, s_bias, y, ldy, s_y, M, K, (float*)ws, s, x_absmax);
  for (int h = 0; h < H; ++h)
    CGAT_TRY(cgat_linear_forward(x + h * s_x, ldx, w + h * s_w, ldw, bias ? bias + h * s_bias : nullptr, y + h * s_y, ldy, M,
                                 K, N, CGAT_ACT_NONE, x_absmax, ws, ws_bytes, stream));
  return CGAT_OK;
}
// The backward of the same layer with LeakyReLU' of the hidden activations folded in (cgat_linear_backward_dact per
// head): all heads' input gradients in one launch pair, all heads' weight / bias gradients in one batched launch.
extern "C" size_t cgat_heads_linear_backward_dact_workspace_bytes(int32_t M, int32_t K, int32_t N, int32_t H) {
  size_t a = cgat_linear_backward_workspace_bytes(M, K, N);
  if (N == 128 && K % 128 == 0 && H >= 1 && H * (K / 128) <= DW_BATCH_MAX) {
    size_t b = (size_t)H * linear128_heads_image_floats(K) * sizeof(float);
    const size_t c = rows_dw128_batch_ws_bytes(H * (K / 128), M);
    if (c > b) b = c;
    if (b + 256 > a) a = b + 256;
  }
  return a;
}
extern "C" int cgat_heads_linear_backward_dact(const float* x, int64_t ldx, int64_t s_x, const float* w, int64_t ldw,
                                               int64_t s_w, const float* g_y, int64_t ldgy, int64_t s_gy, float* g_x,
                                               int64_t ldgx, int64_t s_gx, const float* gx_dact, int64_t ld_dact,
                                               int64_t s_dact, float* gx_absmax, float* g_w, int64_t ldgw, int64_t s_gw,
                                               float* g_b, int64_t s_gb, int32_t M, int32_t K, int32_t N, int32_t H,
                                               void* ws, size_t ws_bytes, void* stream) {
  CGAT_CHECK_ARG(M >= 0 && K >= 0 && N >= 0 && H >= 0, "heads_linear_backward_dact: negative size");
  CGAT_CHECK_ARG(g_x && gx_dact && g_w, "heads_linear_backward_dact: g_x, gx_dact and g_w are required");
  hipStream_t s = (hipStream_t)stream;
  const bool have_ws = ws && ws_bytes >= cgat_heads_linear_backward_dact_workspace_bytes(M, K, N, H);
  const int kb = K / 128;
  if (have_ws && H > 1 && M > 0 && (bilinear_mode() == 2 || bilinear_mode() == 4 || bilinear_mode() == 6) && N == 128 &&
      K % 128 == 0 && H * kb <= DW_BATCH_MAX &&
      heads_strides_ok(s_x, s_w, s_gy, s_gx) && heads_strides_ok(s_dact, s_gw, s_gb, ld_dact) &&
      linear128_fast(N, K, ldgy, ldgx, g_y, g_x) && (((uintptr_t)gx_dact) & 15) == 0) {
    DwBatchDesc b;
    memset(&b, 0, sizeof(b));
    b.rows = M; b.ldg = ldgy; b.ldx = ldx; b.ldo = ldgw;
    for (int h = 0; h < H; ++h)
      for (int j = 0; j < kb; ++j)
        b.it[b.n++] = {g_y + h * s_gy, x + h * s_x + 128 * j, g_w + h * s_gw + 128 * j,
                       (g_b && j == 0) ? g_b + h * s_gb : nullptr};
    if (rows_dw128_batch_fast(b)) {
      // weight seen as out(o = k) x in(n): element W_h[n * ldw + k]  ->  so = 1, sk = ldw
      CGAT_TRY(linear128_heads_launch(H, g_y, ldgy, s_gy, w, 1, ldw, s_w, nullptr, 0, CGAT_ACT_NONE, 0, g_x, ldgx, s_gx, M, ws,
                                      s, K, gx_dact, ld_dact, s_dact, gx_absmax));
      return rows_dw128_batch_launch(b, ws, ws_bytes, s);
    }
  }
  for (int h = 0; h < H; ++h)
    CGAT_TRY(cgat_linear_backward_dact(x + h * s_x, ldx, w + h * s_w, ldw, g_y + h * s_gy, ldgy, g_x + h * s_gx, ldgx,
                                       gx_dact + h * s_dact, ld_dact, gx_absmax, g_w + h * s_gw, ldgw,
                                       g_b ? g_b + h * s_gb : nullptr, M, K, N, ws, ws_bytes, stream));
  return CGAT_OK;
}
static int linear_backward_impl(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* y,
                                int64_t ldy, const float* g_y, int64_t ldgy, float* gpre, float* g_x,
                                int64_t ldgx, int32_t accumulate_gx, float* g_w, int64_t ldgw, float* g_b,
                                int32_t M, int32_t K, int32_t N, int32_t act, void* ws, size_t ws_bytes,
                                void* stream, const float* gx_dact, int64_t ld_dact, float* gx_absmax) {
  hipStream_t s = (hipStream_t)stream;
  CGAT_CHECK_ARG(M >= 0 && K >= 0 && N >= 0, "linear_backward: negative size");
  const float* gp = g_y;
  long ldgp = ldgy;
  if (act != CGAT_ACT_NONE) {
    CGAT_CHECK_ARG(gpre && ldy == N && ldgy == N, "linear_backward: activation backward needs dense y, g_y and a gpre buffer");
    CGAT_TRY(act_bwd_launch(y, g_y, gpre, (long)M * N, act, s));
    gp = gpre;
    ldgp = N;
  }
  if (g_x) {  // g_x = gpre @ W   (W [N, K]: "inputs" are the N columns of gpre, "outputs" the K columns of g_x)
    const bool have_ws = ws && ws_bytes >= cgat_linear_backward_workspace_bytes(M, K, N);
    if (have_ws && N == 128 && linear128_fast(N, K, ldgp, ldgx, gp, g_x)) {
      // weight seen as out(o = k) x in(n): element W[n * ldw + k]  ->  so = 1, sk = ldw
      CGAT_TRY(linear128_launch(gp, ldgp, w, 1, ldw, nullptr, CGAT_ACT_NONE, accumulate_gx, g_x, ldgx, M, ws, s, K, nullptr,
                                gx_dact, ld_dact, gx_absmax));
    } else if (have_ws && K == 128 && N > 128 && edge_ge_fast(128, N, ldgp, 128, ldgx, gp, g_x)) {
      // inputs n (N of them) -> 128 outputs k: element (col = n, out = k) at W[n * ldw + k]
      CGAT_TRY(edge_ge_launch(gp, ldgp, 128, w, ldw, 1, (float*)ws, N, g_x, ldgx, nullptr, M, accumulate_gx, nullptr, s));
    } else {
      GemmParams g = gemm_params(M, K, N, gp, ldgp, w, ldw, g_x, ldgx);
      g.b_kmajor = 1;
      g.beta = accumulate_gx ? 1.f : 0.f;
      if (have_ws) g.splits = gemm_pick_splits_skinny(M, K, N);
      CGAT_TRY(gemm_launch(g, g.splits > 1 ? ws : nullptr, g.splits > 1 ? ws_bytes : 0, s));
    }
  }
  if (g_w && K == 128 && N == 128 && M > 0 && ws && ws_bytes >= rows_dw128_ws_bytes(M, 1) &&
      rows_dw128_fast(gp, ldgp, x, ldx, nullptr, 0)) {
    // g_W = gpre^T @ x and g_b = column sums of gpre in one pass over both operands (rowsdw.hip)
    return rows_dw128_launch(gp, ldgp, x, ldx, g_w, ldgw, nullptr, 0, nullptr, 0, g_b, M, ws, ws_bytes, s);
  }
  // widths that are multiples of 128 on both sides (the per-head second layers of vector attention: 256 -> 128 at E
  // rows, 47 ms of split-K generic GEMMs per step of the Lightning-default network): the (N / 128) x (K / 128) blocks
  // of g_W as ONE batched launch of the rows kernel, bias gradient from the first block column
  {
    const int nb = N / 128, kb = K / 128;
    DwBatchDesc b;
    memset(&b, 0, sizeof(b));
    b.rows = M; b.ldg = ldgp; b.ldx = ldx; b.ldo = ldgw;
    if (g_w && M > 0 && N % 128 == 0 && K % 128 == 0 && nb * kb <= DW_BATCH_MAX && nb * kb > 1 && bilinear_mode() != 0) {
      for (int i = 0; i < nb; ++i)
        for (int j = 0; j < kb; ++j)
          b.it[b.n++] = {gp + 128 * i, x + 128 * j, g_w + (size_t)128 * i * ldgw + 128 * j,
                         (g_b && j == 0) ? g_b + 128 * i : nullptr};
      if (rows_dw128_batch_fast(b) && ws && ws_bytes >= rows_dw128_batch_ws_bytes(b.n, M))
        return rows_dw128_batch_launch(b, ws, ws_bytes, s);
    }
  }
  if (g_w) {  // g_W = gpre^T @ x
    GemmParams g = gemm_params(N, K, M, gp, ldgp, x, ldx, g_w, ldgw);
    g.a_kmajor = 1;
    g.b_kmajor = 1;
    g.splits = gemm_pick_splits(N, K, M);
    CGAT_TRY(gemm_launch(g, ws, ws_bytes, s));
  }
  if (g_b) CGAT_TRY(colsum_launch(gp, ldgp, M, N, g_b, 1.f, ws, ws_bytes, s));
  return CGAT_OK;
}

// ---- segment ops ----
extern "C" int cgat_segment_softmax_forward(const float* a, const float* mult, const int32_t* rowptr, int32_t S,
                                            int32_t F, float eps, float* alpha, void* stream) {
  return seg_softmax_fwd_launch(a, mult, rowptr, S, F, eps, alpha, nullptr, (hipStream_t)stream);
}
extern "C" int cgat_segment_softmax_backward(const float* alpha, const float* g_alpha, const float* mult,
                                             const int32_t* rowptr, int32_t S, int32_t F, float* g_a, float* g_mult,
                                             void* stream) {
  return seg_softmax_bwd_launch(alpha, g_alpha, nullptr, mult, rowptr, S, F, g_a, g_mult, (hipStream_t)stream);
}
extern "C" int cgat_segment_sum(const float* x, int64_t ldx, const int32_t* ridx, const int32_t* rowptr, int32_t S,
                                int32_t F, float* out, int64_t ldo, void* stream) {
  return seg_wsum_launch(x, ldx, ridx, nullptr, 0, 1, rowptr, S, F, CGAT_ACT_NONE, out, ldo, (hipStream_t)stream);
}

extern "C" int cgat_segment_attention_pool_forward(const float* a, int32_t aF, const float* mult, const float* m,
                                                   int64_t ldm, const int32_t* rowptr, const int32_t* ridx, int32_t S, int32_t F,
                                                   float eps, float* out, float* mx, float* inv, float* out_lo,
                                                   void* stream) {
  return seg_attnpool_fwd_launch(a, aF, mult, m, ldm, rowptr, ridx, S, F, eps, out, mx, inv, (hipStream_t)stream, out_lo);
}
extern "C" int cgat_segment_attention_pool_backward(const float* a, int32_t aF, const float* mult, const float* m,
                                                    int64_t ldm, const int32_t* rowptr, const int32_t* ridx, int32_t S, int32_t F,
                                                    const float* out, const float* mx, const float* inv,
                                                    const float* out_lo, const float* g_out, float* g_a, float* g_m,
                                                    int64_t ldgm, float* g_mult, void* stream) {
  return seg_attnpool_bwd_launch(a, aF, mult, m, ldm, rowptr, ridx, S, F, out, mx, inv, g_out, g_a, g_m, ldgm, g_mult,
                                 (hipStream_t)stream, out_lo);
}

// ---- dense-layer chain ----
extern "C" size_t cgat_mlp_chain_workspace_bytes(int32_t n_layers) {
  return (size_t)(n_layers > 0 ? n_layers : 0) * WPREP_IMAGE_FLOATS_MAX * sizeof(float) + 256;
}
extern "C" int cgat_mlp_chain(const cgat_chain_desc* d, void* ws, size_t ws_bytes, void* stream) {
  CGAT_CHECK_ARG(d && d->n_layers >= 1 && d->n_layers <= CHAIN_MAX && d->rows >= 0, "mlp_chain: bad descriptor");
  if (wprep_image_floats() == 0) {
    cgat_set_error("mlp_chain: the f32 arithmetic mode has no fused chain; run the layers one by one");
    return CGAT_ERR_UNSUPPORTED;
  }
  if (!ws || ws_bytes < cgat_mlp_chain_workspace_bytes(d->n_layers)) {
    cgat_set_error("mlp_chain: workspace too small");
    return CGAT_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  WPrepBatch wp;
  wp.n = d->n_layers;
  ChainDesc c;
  memset(&c, 0, sizeof(c));
  float* images = (float*)ws;
  for (int l = 0; l < d->n_layers; ++l) {
    const cgat_chain_layer& L = d->layer[l];
    wp.src[l] = L.W; wp.sb[l] = L.w_sk; wp.sc[l] = L.w_so;
    ChainLayer& o = c.layer[l];
    o.W = (const uint4*)(images + (size_t)l * wprep_image_floats());
    o.bias = L.bias; o.dact = L.dact; o.resid = L.resid; o.out = L.out;
    o.ld_dact = L.ld_dact; o.ld_resid = L.ld_resid; o.ld_out = L.ld_out;
    o.act = L.act; o.dact_type = L.dact_type; o.accumulate = L.accumulate;
  }
  c.n_layers = d->n_layers; c.rows = d->rows; c.x = d->x; c.ldx = d->ldx;
  c.in_dact = d->in_dact; c.ld_in_dact = d->ld_in_dact; c.in_dact_type = d->in_dact_type;
  c.in_store = d->in_store; c.ld_in_store = d->ld_in_store;
  CGAT_CHECK_ARG(mlp_chain128_fast(c), "mlp_chain: rows must be 16-byte aligned with leading dimensions that are multiples of 4");
  CGAT_TRY(prepare_W_batch_launch(wp, images, s));
  return mlp_chain128_launch(c, s);
}

// ---- batched dense-layer weight gradients ----
extern "C" size_t cgat_dense_wgrad_batch_workspace_bytes(int32_t n, int32_t rows) {
  return rows_dw128_batch_ws_bytes(n > 0 ? n : 1, rows > 0 ? rows : 1) + 256;
}
extern "C" int cgat_dense_wgrad_batch(int32_t n, const float* const* G, int64_t ldg, const float* const* X, int64_t ldx,
                                      float* const* out, int64_t ldo, float* const* bsum, int32_t rows, void* ws,
                                      size_t ws_bytes, void* stream) {
  CGAT_CHECK_ARG(n >= 0 && n <= DW_BATCH_MAX && rows >= 0 && G && X && out, "dense_wgrad_batch: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return CGAT_OK;
  if (rows == 0) {   // empty sums
    for (int i = 0; i < n; ++i) {
      for (int o = 0; o < 128; ++o)
        CGAT_TRY(fill_launch(out[i] + o * ldo, 0.f, 128, s));
      if (bsum && bsum[i]) CGAT_TRY(fill_launch(bsum[i], 0.f, 128, s));
    }
    return CGAT_OK;
  }
  DwBatchDesc d;
  memset(&d, 0, sizeof(d));
  d.n = n; d.rows = rows; d.ldg = ldg; d.ldx = ldx; d.ldo = ldo;
  for (int i = 0; i < n; ++i) d.it[i] = {G[i], X[i], out[i], bsum ? bsum[i] : nullptr};
  CGAT_CHECK_ARG(rows_dw128_batch_fast(d), "dense_wgrad_batch: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
  return rows_dw128_batch_launch(d, ws, ws_bytes, s);
}

// ---- kernel-level primitives ----
static GemmParams from_desc(const cgat_gemm_desc* d) {
  GemmParams g = gemm_params(d->M, d->N, d->K, d->A, d->lda, d->B, d->ldb, d->C, d->ldc);
  g.a_kmajor = d->a_kmajor; g.a_rgather = d->a_rgather; g.a_block = d->a_block;
  g.b_kmajor = d->b_kmajor; g.b_kgather = d->b_kgather;
  g.c_scatter = d->c_scatter;
  g.alpha = d->alpha; g.beta = d->beta; g.bias = d->bias;
  g.add1 = d->add1; g.add1_idx = d->add1_idx; g.add2 = d->add2; g.add2_idx = d->add2_idx; g.ld_add = d->ld_add;
  g.act = d->act;
  g.splits = d->splits > 0 ? d->splits : gemm_pick_splits(d->M, d->N, d->K);
  if (g.c_scatter || g.add1 || g.add2) g.splits = 1;
  return g;
}
extern "C" size_t cgat_gemm_workspace_bytes(const cgat_gemm_desc* d) {
  GemmParams g = from_desc(d);
  return gemm_ws_bytes(g) + 256;
}
extern "C" int cgat_gemm(const cgat_gemm_desc* d, void* ws, size_t ws_bytes, void* stream) {
  CGAT_CHECK_ARG(d, "gemm: null descriptor");
  CGAT_CHECK_ARG(!(d->a_kmajor && d->a_rgather), "gemm: a_rgather needs a_kmajor == 0");
  CGAT_CHECK_ARG(!(!d->b_kmajor && d->b_kgather), "gemm: b_kgather needs b_kmajor == 1");
  return gemm_launch(from_desc(d), ws, ws_bytes, (hipStream_t)stream);
}
extern "C" void cgat_set_bilinear_mode(int32_t mode) { bilinear_set_mode(mode); }
extern "C" int32_t cgat_get_bilinear_mode(void) { return bilinear_mode(); }
static size_t cgat_T_image_floats(int NA, int NB, int NC) { return bilinear_T_floats_max(NA, NB, NC); }
extern "C" size_t cgat_bilinear_dual_workspace_bytes(int32_t rows) {
  const size_t tq = ws_round(cgat_T_image_floats(128, 128, 128), 4);
  const size_t a = bilinear_dual_ws_bytes(rows), b = bilinear_rows_ws_bytes(rows, 128, 128, 128);
  return tq + (a > b ? a : b) + 256;
}
extern "C" int cgat_bilinear_dual(const float* p, int64_t ldp, const float* q, int64_t ldq, const float* zz, int64_t ldz,
                                  const float* T, const float* init1, int64_t ldi1, float* out1, int64_t ldo1,
                                  const float* init2, int64_t ldi2, float* out2, int64_t ldo2, int32_t rows, void* ws,
                                  size_t ws_bytes, void* stream) {
  CGAT_CHECK_ARG(rows >= 0, "bilinear_dual: rows < 0");
  if (ws_bytes < cgat_bilinear_dual_workspace_bytes(rows)) {
    cgat_set_error("bilinear_dual: workspace too small");
    return CGAT_ERR_WORKSPACE;
  }
  float* Tq = (float*)ws;
  const size_t off = ws_round(cgat_T_image_floats(128, 128, 128), 4);
  void* rest = (char*)ws + off;
  hipStream_t s = (hipStream_t)stream;
  if (bilinear_dual_fast(128, 128, 128)) {
    CGAT_TRY(bilinear_prepare_T(T, Tq, 128, 128, 128, 0, 1, 2, s));
    return bilinear_dual_launch(p, ldp, q, ldq, zz, ldz, Tq, init1, ldi1, out1, ldo1, init2, ldi2, out2, ldo2, rows, rest,
                                ws_bytes - off, s);
  }
  CGAT_TRY(bilinear_prepare_T(T, Tq, 128, 128, 128, 0, 1, 2, s));
  CGAT_TRY(bilinear_rows_launch(p, ldp, q, ldq, Tq, init1, ldi1, out1, ldo1, rows, 128, 128, 128, rest, ws_bytes - off, s));
  CGAT_TRY(bilinear_prepare_T(T, Tq, 128, 128, 128, 2, 1, 0, s));   // [c][b][a]: out2[n,a] = sum_{c,b} zz[c] q[b] T[a,b,c]
  return bilinear_rows_launch(zz, ldz, q, ldq, Tq, init2, ldi2, out2, ldo2, rows, 128, 128, 128, rest, ws_bytes - off, s);
}
extern "C" size_t cgat_bilinear_rows_workspace_bytes(int32_t rows, int32_t NA, int32_t NB, int32_t NC) {
  return ws_round(cgat_T_image_floats(NA, NB, NC), 4) + bilinear_rows_ws_bytes(rows, NA, NB, NC) + 256;
}
extern "C" int cgat_bilinear_rows(const float* p, int64_t ldp, const float* q, int64_t ldq, const float* T,
                                  const float* init, int64_t ldi, float* out, int64_t ldo, int32_t rows, int32_t NA,
                                  int32_t NB, int32_t NC, void* ws, size_t ws_bytes, void* stream) {
  if (ws_bytes < cgat_bilinear_rows_workspace_bytes(rows, NA, NB, NC)) {
    cgat_set_error("bilinear_rows: workspace too small");
    return CGAT_ERR_WORKSPACE;
  }
  float* Tq = (float*)ws;
  size_t off = ws_round(cgat_T_image_floats(NA, NB, NC), 4);
  CGAT_TRY(bilinear_prepare_T(T, Tq, NA, NB, NC, 0, 1, 2, (hipStream_t)stream));
  return bilinear_rows_launch(p, ldp, q, ldq, Tq, init, ldi, out, ldo, rows, NA, NB, NC, (char*)ws + off,
                              ws_bytes - off, (hipStream_t)stream);
}
extern "C" size_t cgat_bilinear_wgrad_workspace_bytes(int32_t rows, int32_t NA, int32_t NB, int32_t NC) {
  return bilinear_wgrad_ws_bytes(rows, NA, NB, NC) + 256;
}
extern "C" int cgat_bilinear_wgrad(const float* p, int64_t ldp, const float* q, int64_t ldq, const float* r,
                                   int64_t ldr, float* out, int32_t rows, int32_t NA, int32_t NB, int32_t NC, void* ws,
                                   size_t ws_bytes, void* stream) {
  return bilinear_wgrad_launch(p, ldp, q, ldq, r, ldr, out, rows, NA, NB, NC, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int cgat_layernorm_tanh_forward(const float* u, float* y, int32_t rows, int32_t W, float eps, void* stream) {
  return layernorm_tanh_fwd_launch(u, y, rows, W, eps, (hipStream_t)stream);
}
extern "C" int cgat_layernorm_tanh_backward(const float* u, const float* y, const float* g_y, float* g_u, int32_t rows,
                                            int32_t W, float eps, void* stream) {
  return layernorm_tanh_bwd_launch(u, y, g_y, g_u, rows, W, eps, (hipStream_t)stream);
}
