// Internal launch interface between the kernel files and the layer orchestrators.
#pragma once
#include "common.h"

#define CGAT_ACT_NONE 0
#define CGAT_ACT_TANH 1
#define CGAT_ACT_LEAKY 2  // LeakyReLU, slope 0.01 (nn.LeakyReLU() default; reference CGAT.py:95)
#define CGAT_ACT_RELU 3

struct GemmParams {
  int M, N, K;
  const float* A;
  long lda;
  int a_kmajor;          // 0: A(m,k)=A[row(m)*lda+k]   1: A(m,k)=A[k*lda+m]
  const int* a_rgather;  // a_kmajor==0 only: row(m)=a_rgather[m]
  long a_block;          // != 0: A is stored in 128-wide column blocks [cols/128][rows][128] with this block
                         // stride (elements); the blocked dimension is k (a_kmajor==0) or m (a_kmajor==1); lda=128
  const float* B;
  long ldb;
  int b_kmajor;          // 0: B(k,n)=B[n*ldb+k] (torch Linear weight)   1: B(k,n)=B[krow(k)*ldb+n]
  const int* b_kgather;  // b_kmajor==1 only: krow(k)=b_kgather[k]
  float* C;
  long ldc;
  const int* c_scatter;  // output row = c_scatter[m]
  float alpha, beta;     // C = act(alpha*acc + bias + adds) + beta*C
  const float* bias;     // [N]
  const float* add1;     // rows gathered by add1_idx[m], leading dim ld_add
  const int* add1_idx;
  const float* add2;
  const int* add2_idx;
  long ld_add;
  int act;
  int splits;            // >1: split the K range, partial slabs in workspace, then reduce
  // Row-wise outer-product (Khatri-Rao) operands: the bilinear contractions at widths other than 128 as plain products
  //   a_outer (a_kmajor == 0, b_kmajor == 1): A(m,k) = a_outer[m*ld_a_outer + k / outer_n] * A[m*lda + k % outer_n]
  //   b_outer (a_kmajor == 1, b_kmajor == 1): B(k,n) = b_outer[k*ld_b_outer + n / outer_n] * B[k*ldb + n % outer_n]
  const float* a_outer;
  long ld_a_outer;
  const float* b_outer;
  long ld_b_outer;
  int outer_n;
  // filled by gemm_launch
  int k_per_split, a_vec, b_vec, c_vec;
  float* slab;
};

static inline GemmParams gemm_params(int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C,
                                     long ldc) {
  GemmParams p = {};
  p.M = M; p.N = N; p.K = K;
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
  p.alpha = 1.f; p.beta = 0.f; p.splits = 1;
  return p;
}

int gemm_launch(GemmParams p, void* ws, size_t ws_bytes, hipStream_t stream);
// the same product on the bf16 matrix cores (exact three-plane split, six passes: gemmsplit.hip); called by gemm_launch
// in the split arithmetic modes with p's derived fields filled in
int gemm_split_launch(const GemmParams& p, unsigned grid, hipStream_t stream);
int gemm_pick_splits(int M, int N, int K);
// split count for products with <= 32 output tiles and K >= 512 (1 otherwise): see gemm.hip
int gemm_pick_splits_skinny(int M, int N, int K);
// C[m*ldc + n] = act(alpha * sum_z slab[z][m][n] + bias[n]) + beta * C   (fixed summation tree)
int splitk_reduce_launch(const float* slab, int splits, int M, int N, float* C, long ldc, hipStream_t stream,
                         float alpha = 1.f, float beta = 0.f, const float* bias = nullptr, int act = 0);
size_t gemm_ws_bytes(const GemmParams& p);

// ---- small-row products as single-op programs of rowprog.hip (exact fp32 MFMA, no workspace) ----
struct cgat_rowprog_op;
int rowprog_max_rows();                       // env CGAT_ROWPROG_MAX_ROWS, default 2048; 0: never
bool rowprog_gemm_ok(const GemmParams& p);    // plain strided product of at most rowprog_max_rows() rows
int rowprog_gemm(const GemmParams& p, hipStream_t s);
int rowprog_launch(const cgat_rowprog_op* ops, int n_ops, uint32_t* sync_words, hipStream_t s);

// ---- bilinear (hypernetwork) contractions, bilinear.hip ----
// out[n,c] = init[n,c] + sum_{a<NA,b<NB} p[n,a] q[n,b] T3[a,b,c], with T = bilinear_prepare_T(T3 source):
// a permuted copy whose columns are interleaved for the MFMA kernel when NB == NC == 128.
bool bilinear_T_interleaved(int NB, int NC);
int bilinear_mode();               // 0 f32 MFMA, 6 / 3: split-bf16 passes, 2: f16x3 (two fp16 planes), 4: f16x3c (h + l + t)
void bilinear_set_mode(int m);
size_t bilinear_T_floats(int NA, int NB, int NC);  // workspace floats of the prepared T
size_t bilinear_T_floats_max(int NA, int NB, int NC);   // ... in whichever arithmetic mode needs most (size queries)
// several tensors in two launches (f16x3 mode at width 128; CGAT_ERR_UNSUPPORTED otherwise -> prepare one by one)
#define TPREP_MAX 8
struct TPrepBatch {
  int n;
  const float* src[TPREP_MAX];
  void* dst[TPREP_MAX];
};
size_t bilinear_prepare_T_batch_ws_floats(int n);
int bilinear_prepare_T_batch(int n, const float* const* src, float* const* dst, int n0, int n1, int n2, int perm0,
                             int perm1, int perm2, float* part, hipStream_t stream, int alternate = 1);
int bilinear_prepare_T(const float* src, float* dst, int n0, int n1, int n2, int perm0, int perm1, int perm2,
                       hipStream_t stream);
size_t bilinear_rows_ws_bytes(int nrows, int NA, int NB, int NC);
int bilinear_rows_launch(const float* p, long ldp, const float* q, long ldq, const float* T, const float* init,
                         long ldi, float* out, long ldo, int nrows, int NA, int NB, int NC, void* ws, size_t ws_bytes,
                         hipStream_t stream, float* ln_out = nullptr, float ln_eps = 0.f);   // ln_out: tanh(LayerNorm(out))
// fused pair (bilinear.hip, bilinear_rows128_dual_kernel): T = bilinear_prepare_T of the [128,128,128] operand
bool bilinear_dual_fast(int NA, int NB, int NC);
size_t bilinear_dual_ws_bytes(int nrows);
int bilinear_dual_launch(const float* p, long ldp, const float* q, long ldq, const float* zz, long ldz, const float* T,
                         const float* init1, long ldi1, float* out1, long ldo1, const float* init2, long ldi2,
                         float* out2, long ldo2, int nrows, void* ws, size_t ws_bytes, hipStream_t stream);
// out[(a*NB+b)*NC + c] = sum_n p[n,a] q[n,b] r[n,c]      (workspace: slabs)
size_t bilinear_wgrad_ws_bytes(int nrows, int NA, int NB, int NC);
int bilinear_wgrad_launch(const float* p, long ldp, const float* q, long ldq, const float* r, long ldr, float* out,
                          int nrows, int NA, int NB, int NC, void* ws, size_t ws_bytes, hipStream_t stream,
                          int force_splits = 0);
// the same for n_layers (<= 8) operand triples in ONE launch (f16x3 mode: batched, software-pipelined kernel; other
// modes: one launch per layer).  max_wgs: grid size (0 = one workgroup per CU; 128 = half of the chip)
size_t bilinear_wgrad_batch_ws_bytes(int n_layers, int nrows, int NA, int NB, int NC);
int bilinear_wgrad_batch_launch(int n_layers, const float* const* p, long ldp, const float* const* q, long ldq,
                                const float* const* r, long ldr, float* const* out, int nrows, int NA, int NB, int NC,
                                void* ws, size_t ws_bytes, hipStream_t stream, int max_wgs = 0, bool prepared = false);
// one layer's operand preparation for that launch (slot of n_layers), issued separately -- e.g. early, on a side
// stream; CGAT_ERR_UNSUPPORTED when the batched form does not take the operands
int bilinear_wgrad_batch_prep(int slot, int n_layers, const float* p, long ldp, const float* q, long ldq, const float* r,
                              long ldr, int nrows, int NA, int NB, int NC, void* ws, size_t ws_bytes, hipStream_t stream);
// three bf16 planes of sgn(a) * src[a*sa + b*sb + c*sc] (a < NA; b, c < 128) in the ring kernels' fragment order
int prepare_T_bf16_launch(const float* src, void* dst, int NA, long sa, long sb, long sc, int alternate,
                          hipStream_t stream);
int prepare_T_bf16_heads_launch(const float* src, void* dst, int NA, long sa, long sb, long sc, int alternate, int heads,
                                long s_head, long image_floats, hipStream_t stream);
// f16x3c image (bilinear.hip, prepare_T_f16c_kernel) of a dense-layer weight: W2 output rows of 128 inputs, row stride ldw
size_t prepare_W_f16c_rows_floats(int W2);
int prepare_W_f16c_rows_launch(const float* W, long ldw, int W2, void* dst, hipStream_t stream);
// ---- fused edge pre-activations + attention logits, edgez.hip ----
bool edge_z_fast(int Ce, int W2, int H, int Hd, long lde, long ld_add, long ldz, const void* e, const void* Pi,
                 const void* Pj, const void* Z, const void* wA);
size_t edge_z_wq_floats(int W2);
// per-edge launch with the x_j projection folded in (f16x3 mode, C == Ce == 128): edgez.hip edge_zx_kernel
size_t edge_zx_wq_floats(int W2);
bool edge_zx_fast(int C, int Ce, int W2, int H, int Hd, long ld_add, long ldz, const void* e, const void* x,
                  const void* Pi, const void* Z, const void* wA);
int edge_zx_launch(const float* e, long lde, const int* perm, const float* x, long ldx, const float* We, const float* Wj,
                   long ldw, float* Wq, int W2, const float* Pi, const int* dsti, const int* srci, long ld_add, float* Z,
                   long ldz, int E, const float* wA, const float* bA, int H, int Hd, float* a_out, hipStream_t stream,
                   int z_bf16 = 0);   // z_bf16: Z stored as bf16 (ldz in elements either way)
int edge_z_launch(const float* e, long lde, const int* perm, const float* We, long ldw, float* Wq, int W2,
                  const float* Pi, const int* dsti, const float* Pj, const int* srci, long ld_add, float* Z, long ldz,
                  int E, const float* wA, const float* bA, int H, int Hd, float* a_out, hipStream_t stream,
                  int act = CGAT_ACT_NONE, float* omax = nullptr, int z_bf16 = 0,   // z_bf16: as edge_zx_launch
                  int n_add_rows = 0);   // rows of Pi / Pj (0 = unknown: the f16x3c kernel addresses them by 32-bit offsets)
int absmax_launch(const float* src, long n, float* out, hipStream_t stream);   // zeroes out[0] first
int absmax_rows128_launch(const float* t, long ld, int rows, float* out, hipStream_t stream);  // folds into out[0]
// fp16 form for weight operands: planes of 2^k(a) W[a], max |W[a]| in ((float*)dst)[NA * 16384 + a]
int prepare_W_f16_launch(const float* src, void* dst, int NA, long sa, long sb, long sc, hipStream_t stream);
// the same for `heads` weights of NA blocks each in ONE launch: head h reads src + h * s_head and writes the image
// (NA planes blocks, then NA maxima) at (float*)dst + h * image_floats
int prepare_W_f16_heads_launch(const float* src, void* dst, int NA, long sa, long sb, long sc, int heads, long s_head,
                               long image_floats, hipStream_t stream);
// Per-head offsets of a launch whose grid.y runs over the heads of a multi-head second layer (elements of each operand;
// w in 16-byte pieces): all zero = the single-operand launch
struct HeadBatch {
  long in, w, bias, out, dact;
  long add2;   // edge_z_kernel: offset of the second gathered table (column groups of the per-edge first layer)
  long ks0;    // edge_ge_kernel<., RC>: k-steps in front of a K group (the rebuilt rows are indexed by k-step, not by pointer);
               // edge_z_kernel with logits: column blocks in front of a column group (head index, fc_out_A's weight)
};
// column blocks of a [rows, 128 ncb] product dealt to grid.y groups when the row tiles alone leave CUs idle: the number
// of groups (a divisor of ncb; 1 = no split)
// (unit: column blocks that must stay in one group -- a head's blocks when the launch also forms that head's logits)
static inline int z_col_groups(int row_tiles, int ncb, int unit = 1) {
  int G = 1;
  if (row_tiles >= 512 || unit < 1) return 1;
  for (int g = 2; g <= ncb; ++g)
    if (ncb % g == 0 && (ncb / g) % unit == 0) { G = g; if ((long)row_tiles * G >= 512) break; }
  return G;
}
size_t linear128_heads_image_floats(int n_out);
int linear128_heads_launch(int heads, const float* in, long ldi, long s_in, const float* W, long so, long sk, long s_w,
                           const float* bias, long s_bias, int act, int accumulate, float* out, long ldo, long s_out, int rows,
                           void* ws, hipStream_t stream, int n_out, const float* dact, long ld_dact, long s_dact,
                           float* omax);
// batched form for 128 x 128 dense-layer weights W(o, k) = src[o * so + k * sk]: image i at dst + i * WPREP_IMAGE_FLOATS
#define WPREP_MAX 48
#define WPREP_IMAGE_FLOATS (16384 + 4)
#define WPREP_IMAGE_FLOATS_X6 24576             // the six-pass bf16 chain's image: three 2-byte planes (chain.hip)
#define WPREP_IMAGE_FLOATS_MAX 24576
size_t wprep_image_floats();                    // ... of the current arithmetic mode (0: it has no fused chain)
struct WPrepBatch {
  const float* src[WPREP_MAX];
  long sb[WPREP_MAX], sc[WPREP_MAX];   // strides of the k index and of the output index
  int n;
};
int prepare_W_f16_batch_launch(const WPrepBatch& b, float* dst, hipStream_t stream);
// three-bf16-plane images (the dense-layer kernel's operand in the 24-bit modes) of b.n 128 x 128 weights, 24576 floats each
int prepare_T_bf16_batch_launch(const WPrepBatch& b, float* dst, hipStream_t stream);
int prepare_T_bf16_rows_launch(const float* rows, long ld, const int* gather, int nrows, void* dst, int NA,
                               hipStream_t stream, const float* emax = nullptr);   // emax: fp16 form scaled by max |rows|
int prepare_T_f16_scaled_launch(const float* src, void* dst, int NA, long sa, long sb, long sc, const float* tmax,
                                hipStream_t stream, int alternate = 0);
// dense layer at width 128 on the split-bf16 kernel: out = act(in W^T + bias) (+ out),  W(o, k) = W[o*so + k*sk]
bool linear128_fast(int K, int N, long ldi, long ldo, const void* in, const void* out);
size_t linear128_ws_bytes(int n_out = 128);
int linear128_launch(const float* in, long ldi, const float* W, long so, long sk, const float* bias, int act, int accumulate,
                     float* out, long ldo, int rows, void* ws, hipStream_t stream, int n_out = 128,
                     const void* prepared = nullptr,    // prepared: image from prepare_W_f16_batch_launch (f16x3, n_out 128)
                     const float* dact = nullptr, long ld_dact = 0,   // out *= LeakyReLU'(sign of dact[row, col])
                     float* omax = nullptr);            // max |out| folded into omax[0] (not zeroed here)
// ---- a chain of width-128 dense layers in one launch (f16x3 mode), chain.hip ----
#define CHAIN_MAX 5
struct ChainLayer {
  const uint4* W;        // prepared image (prepare_W_f16_batch_launch: 8 chunks of 8 KB, its scale behind them)
  const float* bias;     // [128] or null
  const float* dact;     // null, or saved activation values y [rows,128]: the layer's result is multiplied by act'(y)
  const float* resid;    // null, or [rows,128] added to the (activated) result
  float* out;            // null, or where the result goes
  long ld_dact, ld_resid, ld_out;
  int act;               // activation of the layer (CGAT_ACT_*)
  int dact_type;         // activation whose derivative `dact` stands for
  int accumulate;        // out += result
};
struct ChainDesc {
  ChainLayer layer[CHAIN_MAX];
  int n_layers, rows;
  const float* x;        // input rows [rows,128]
  long ldx;
  const float* in_dact;  // null, or the input rows are multiplied by act'(in_dact) first (backward of an activation)
  long ld_in_dact;
  int in_dact_type;
  float* in_store;       // null, or where the (multiplied) input rows are stored
  long ld_in_store;
};
int prepare_W_batch_launch(const WPrepBatch& b, float* dst, hipStream_t stream);   // the current mode's chain images
bool mlp_chain128_fast(const ChainDesc& d);
int mlp_chain128_launch(const ChainDesc& d, hipStream_t stream);
#define CHAIN_BATCH_MAX 4
// n independent chains over the same rows in one launch (24-bit modes; otherwise one launch each)
int mlp_chain128_batch_launch(const ChainDesc* d, int n, hipStream_t stream);
// ---- split-bf16 backward products over gZ, edgebwd.hip ----
// The pre-activation gradient of the scalar-attention layer is never stored when its consumers can rebuild it
// (edge_seg_bwd_kernel step 3, layers.hip): for destination-sorted slot t and column col of the stacked hidden layer
//     gZ[t, col] = (ga[t,h]    * wA[col])                 * d      col <  HHd  (attention network, h = col / Hd)
//                = (alpha[t,h] * gS[dst[t], col - HHd])   * d      col >= HHd  (message network,  h = (col-HHd) / Hd)
// with d = 1 if Z[t,col] > 0 else 0.01 (LeakyReLU').  Per edge that is 2 H scalars, one bit per column and a row of a
// per-NODE matrix shared by all edges of the destination: 216 bytes + cache hits instead of 6 KB of HBM per read.
struct EdgeRC {
  const unsigned* mask;   // [E][W2 / 32]: bit (col & 31) of word (col >> 5) = (Z[t, col] > 0)
  const float* ga;        // [E][H]
  const float* alpha;     // [E][H]
  const float* gS;        // [N][HHd]
  const float* wA;        // [HHd]
  const int* dst;         // [E] destination node of slot t
  int H, Hd, HHd, nw;     // nw = W2 / 32
};
// edge storage mode 3 ("bf16-mma", layers.hip): the per-edge products over the rebuilt gZ rows run ONE bf16 pass
bool edge_mma_bf16();
// shapes the rebuilding kernels take: whole 128-column blocks inside one head, 256-column groups in the mask writer
static inline bool edge_rc_shape(int Ce, int H, int Hd) { return Ce == 128 && H > 0 && Hd > 0 && Hd % 128 == 0; }
// operand element (t, 128 a + j) at gZ[t * ldg + a * gzb + j]: (128, E*128) = column-blocked, (W2, 128) = row-major;
// with rc != null the operand is rebuilt from *rc instead (gZ, ldg, gzb unused)
bool edge_gw_fast(int Ce, int W2, long ldg, long gzb, const void* gZ);
size_t edge_gw_ws_floats(int E, int W2);
int edge_gw_launch(const float* gZ, long ldg, long gzb, const float* e, long lde, const int* perm, int E, int W2,
                   float* ws, float* out, long ldo, hipStream_t stream, const float* gmax = nullptr,
                   const float* emax = nullptr,            // device maxima of |gZ| and |e| -> fp16 form in the f16x3 mode
                   const EdgeRC* rc = nullptr);
bool edge_ge_fast(int Ce, int W2, long ldg, long gzb, long ldo, const void* gZ, const void* out);
size_t edge_ge_heads_image_floats(int W2);
bool edge_ge_heads_fast(int heads, int W2, long ldx, long ldy, long ldw, const void* x, const void* w, const void* y,
                        const float* amax);
int edge_ge_heads_launch(int heads, const float* x, long ldx, long s_x, const float* W, long s_w, const float* bias,
                         long s_bias, float* y, long ldy, long s_y, int E, int W2, float* ws, hipStream_t stream,
                         const float* amax);
int edge_ge_ksplit_groups(int E, int W2);   // K-split form for few row tiles: groups a launch takes (1: use edge_ge_launch)
int edge_ge_ksplit_launch(const float* gZ, long ldg, long gzb, const float* We, long s_col, long s_out, float* Wq, int W2,
                          float* slabs, const int* scatter, int E, int S, hipStream_t stream,
                          const EdgeRC* rc = nullptr);   // S slabs [E,128]; caller sums
int edge_ge_prepared_launch(const float* x, long ldx, const void* Wq, int ncb, float* out, long ldo, int rows,
                            int accumulate, hipStream_t stream);   // six-pass image made by the caller (odd blocks negated)
int edge_ge_launch(const float* gZ, long ldg, long gzb, const float* We, long s_col, long s_out, float* Wq, int W2,
                   float* out, long ldo, const int* scatter, int E, int accumulate, const float* bias,
                   hipStream_t stream,
                   const float* amax = nullptr,            // amax: device max |gZ| -> fp16 form in the f16x3 mode
                   const EdgeRC* rc = nullptr);
// Gj[n, :] = sum over the edges leaving n of the rebuilt gZ rows (src_rowptr / src_pos: slots grouped by source)
int edge_gj_launch(const EdgeRC& rc, const int* src_rowptr, const int* src_pos, int N, int W2, float* Gj, long ldo,
                   hipStream_t stream, float* gjmax = nullptr);   // gjmax: max |Gj| folded in (zeroed by the caller)
// dst[(a*d1 + b)*d2 + c] = src[...] under an index permutation of a [n0,n1,n2] tensor
int permute3_launch(const float* src, float* dst, int n0, int n1, int n2, int perm0, int perm1, int perm2,
                    int interleave, hipStream_t stream);

// ---- width-128 weight gradients over rows, rowsdw.hip: out_k[o][i] = sum_n G[n,o] X_k[n,i], bsum[o] = sum_n G[n,o] ----
bool rows_dw128_fast(const float* G, long ldg, const float* X1, long ldx1, const float* X2, long ldx2);
size_t rows_dw128_ws_bytes(int rows, int nx);
int rows_dw128_launch(const float* G, long ldg, const float* X1, long ldx1, float* out1, long ldo1, const float* X2,
                      long ldx2, float* out2, long ldo2, float* bsum, int rows, void* ws, size_t ws_bytes,
                      hipStream_t stream);
// many such products in one launch (every item: one right operand; common row count and leading dimensions)
#define DW_BATCH_MAX 32
struct DwBatchItem {
  const float* G;    // [rows, 128], row stride ldg
  const float* X;    // [rows, 128], row stride ldx
  float* out;        // [128][ldo]: G^T X
  float* bsum;       // [128] column sums of G, or null
};
struct DwBatchDesc {
  int n, rows, splits, rows_per_unit;   // splits / rows_per_unit are set by the launch
  long ldg, ldx, ldo;
  DwBatchItem it[DW_BATCH_MAX];
};
size_t rows_dw128_batch_ws_bytes(int n_items, int rows);
bool rows_dw128_batch_fast(const DwBatchDesc& d);
int rows_dw128_batch_launch(DwBatchDesc d, void* ws, size_t ws_bytes, hipStream_t stream);

// ---- elementwise / row kernels, rowops.hip ----
int layernorm_tanh_fwd_launch(const float* u, float* y, int rows, int W, float eps, hipStream_t s);
int layernorm_tanh_bwd_launch(const float* u, const float* y, const float* gy, float* gu, int rows, int W, float eps,
                              hipStream_t s);
int act_bwd_launch(const float* y, const float* gy, float* gpre, long n, int act, hipStream_t s);  // in terms of post-activation y
int act_bwd_leaky_max_launch(const float* y, const float* gy, float* gpre, long n, float* gmax, hipStream_t s, bool* used);
int colsum_launch(const float* x, long ldx, int rows, int cols, float* out, float alpha, void* ws, size_t ws_bytes,
                  hipStream_t s);
size_t colsum_ws_bytes(int rows, int cols);
int mix_launch(const float* a, const float* b, const float* d, float* out, long n, hipStream_t s);  // out = d*a + (1-d)*b
int mix_bwd_launch(const float* g, const float* a, const float* b, const float* d, float* ga, float* gb_accum,
                   float* gd, long n, void* ws, size_t ws_bytes, hipStream_t s);
int scale_launch(const float* x, float alpha, float* out, long n, hipStream_t s);   // out = alpha * x
int axpy_launch(float* y, const float* x, float alpha, long n, hipStream_t s);  // y += alpha*x
int sum_slabs_launch(const float* slabs, int n, long stride, float* out, long count, hipStream_t s);   // out = 0.f + slab 0 + slab 1 + ...
int copy2d_launch(const float* src, long lds, float* dst, long ldd, int rows, int cols, hipStream_t s);
struct Copy2DJob { const float* src; long lds; float* dst; long ldd; int rows, cols; };
struct Copy2DJobs { Copy2DJob job[4]; int n; };
int copy2d_multi_launch(const Copy2DJobs& j, hipStream_t s);   // n <= 4 copies in one launch
int fill_launch(float* p, float v, long n, hipStream_t s);

// ---- segment kernels (rows sorted by segment, rowptr[S+1]), segment.hip ----
int seg_softmax_fwd_launch(const float* a, const float* mult, const int* rowptr, int S, int F, float eps, float* alpha,
                           float* ssum, hipStream_t s);
int seg_softmax_bwd_launch(const float* alpha, const float* galpha, const float* gssum, const float* mult,
                           const int* rowptr, int S, int F, float* ga, float* gmult, hipStream_t s);
// out[s, f] = sum_{r in seg s} w[r, f / fw] * act(x[r or ridx[r], f])      (w nullable, fw = features per weight)
int seg_wsum_launch(const float* x, long ldx, const int* ridx, const float* w, int wF, int fw, const int* rowptr, int S,
                    int F, int act, float* out, long ldo, hipStream_t s, long xblock = 0, int x_bf16 = 0);   // x_bf16: x holds bf16 (ldx, xblock in elements)
// softmax-weighted segment sum (attention pooling) in one pass per direction, see segment.hip
bool seg_attnpool_fast(int aF, int F, long ldm, const void* a, const void* m, const void* out);
int seg_attnpool_fwd_launch(const float* a, int aF, const float* mult, const float* m, long ldm, const int* rowptr,
                            const int* ridx, int S, int F, float eps, float* out, float* mx, float* inv, hipStream_t s,
                            float* out_lo = nullptr);     // out_lo: low part of the fp64 sum, for backward (may be null)
int seg_attnpool_bwd_launch(const float* a, int aF, const float* mult, const float* m, long ldm, const int* rowptr,
                            const int* ridx, int S, int F, const float* out, const float* mx, const float* inv, const float* g_out, float* g_a,
                            float* g_m, long ldgm, float* g_mult, hipStream_t s, const float* out_lo = nullptr);
// per-row, per-head dot:  out[r,h] = sum_j act(x[r, h*Hd+j]) * v[(vrow(r)) * ldv + h*Hd + j] + bias[h] (+ addv[vrow(r)*H + h])
int rowdot_launch(const float* x, long ldx, int act, const float* v, long ldv, const int* vrow, const float* bias,
                  const float* addv, int rows, int H, int Hd, float* out, hipStream_t s);

// ---- CSR plan, plan.hip ----
size_t plan_ws_bytes(int E, int N);
int plan_build_launch(const int64_t* edge_index, int E, int N, int* dst_rowptr, int* dst_perm, int* dst_sorted,
                      int* src_sorted, int* src_rowptr, int* src_pos, void* ws, size_t ws_bytes, hipStream_t s);
int csr_from_keys_launch(const int* keys, int n, int S, int* rowptr, int* perm, void* ws, size_t ws_bytes, hipStream_t s);
