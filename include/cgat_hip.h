/* libcgat_hip -- C ABI of the MI355X (gfx950) edge-attention hot path of hyllios/CGAT.
 *
 * Every pointer is a DEVICE pointer to contiguous fp32 / int32 / int64 data unless stated
 * otherwise; `stream` is a hipStream_t passed as void*.  Functions return 0 on success; on
 * failure cgat_last_error() describes the problem.  No function allocates device memory or
 * synchronises the host: outputs, saved-for-backward buffers and workspaces are provided by
 * the caller (size queries below), so every call is hipGraph-capturable.
 *
 * The reference (pure Python) has no FFI; each entry point names the reference operator it
 * replaces (paths relative to the reference repository root).
 */
#ifndef CGAT_HIP_H
#define CGAT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGAT_ABI_VERSION 3   /* 2: cgat_linear_forward / cgat_edge_hidden_forward / _backward carry tensor maxima (round 3)
                              * 3: cgat_segment_attention_pool_* carry out_lo; arithmetic mode 4 = "f16x3c" (round 4) */
#define CGAT_MAX_FC 8      /* Linear+Tanh layers per hypernetwork trunk */
#define CGAT_MAX_HYPER 8   /* predicted layers per hypernetwork */

int cgat_abi_version(void);
const char* cgat_last_error(void);

/* ---- launch timing (HIP events on the launch stream) --------------------------------- */
void cgat_prof_enable(int on);
void cgat_prof_reset(void);
/* sums all recorded launches of kernels tagged `tag` ("bilinear_rows", "bilinear_wgrad",
 * "gemm_f32", ...); synchronises the recorded events. */
int cgat_prof_get(const char* tag, int* count, float* total_ms);
/* kernel launches issued by the library since the process started (every launch is counted, on any stream) */
uint64_t cgat_prof_launches(void);

/* ---- CSR plan: PyG propagate's implicit segment structure, built once per batch --------
 * replaces the index handling of torch_geometric MessagePassing.propagate / utils.softmax /
 * torch_scatter.scatter_add as used at CGAT/CGAT.py:313-326 (aggregation keyed by
 * edge_index[1]).  dst_perm: edge ids sorted (stably) by edge_index[1]; dst_rowptr[N+1];
 * dst_sorted/src_sorted: endpoints in that order; (src_rowptr, src_pos): sorted positions
 * grouped by edge_index[0]. */
typedef struct cgat_plan {
  int32_t N, E;
  const int32_t* dst_rowptr;
  const int32_t* dst_perm;
  const int32_t* dst_sorted;
  const int32_t* src_sorted;
  const int32_t* src_rowptr;
  const int32_t* src_pos;
} cgat_plan;
size_t cgat_plan_workspace_bytes(int32_t E, int32_t N);
int cgat_plan_build(const int64_t* edge_index /* [2,E] */, int32_t E, int32_t N, int32_t* dst_rowptr /* [N+1] */,
                    int32_t* dst_perm /* [E] */, int32_t* dst_sorted /* [E] */, int32_t* src_sorted /* [E] */,
                    int32_t* src_rowptr /* [N+1] */, int32_t* src_pos /* [E] */, void* ws, size_t ws_bytes,
                    void* stream);
/* generic CSR from int32 keys in [0,S): rowptr[S+1], perm[n] (ids ascending inside a segment) */
size_t cgat_csr_workspace_bytes(int32_t S);
int cgat_csr_from_keys(const int32_t* keys, int32_t n, int32_t S, int32_t* rowptr, int32_t* perm, void* ws,
                       size_t ws_bytes, void* stream);

/* ---- batch collation on the device (the step before the hot path) -----------
 * replaces, per training step, CGAT/data.py:61-144 (CompositionData.__getitem__ for every crystal of the batch),
 * PyG Batch.from_data_list (CGAT/lightning_module.py:200) and CGAT/roost_message.py:400-458 (collate_batch).
 * The dataset lives on the device in packed int32 form (built once): per-atom element ids into the embedding table,
 * the neighbour tables already sliced to max_nbr columns, the per-crystal composition (unique elements in
 * first-appearance order and their weights) and y = target * n_atoms (or the plain target for 'volume'). */
typedef struct cgat_packed_dataset {
  int32_t n_graphs, fea, max_nbr, n_elem;
  const float* table;        /* [n_elem, fea] element embedding */
  const int32_t* atom_ptr;   /* [n_graphs+1] */
  const int32_t* atom_elem;  /* [atoms] row of `table` */
  const int32_t* shell;      /* [atoms, max_nbr] edge_attr ids */
  const int32_t* self_idx;   /* [atoms, max_nbr] crystal-local centre atom */
  const int32_t* nbr_idx;    /* [atoms, max_nbr] crystal-local neighbour atom */
  const int32_t* comp_ptr;   /* [n_graphs+1] */
  const int32_t* comp_elem;  /* [unique elements] row of `table` */
  const float* comp_weight;  /* [unique elements] count / n_atoms */
  const float* y_val;        /* [n_graphs] */
} cgat_packed_dataset;
typedef struct cgat_collated {   /* outputs, caller-allocated; dtypes as the reference's tensors */
  float* x;                  /* [N, fea] */
  int64_t* edge_index;       /* [2, E], E = N * max_nbr */
  int64_t* edge_attr;        /* [E] */
  float* y;                  /* [B] */
  int64_t* batch;            /* [N] */
  float* comp_weight;        /* [Nc] (the reference views it as [Nc,1]) */
  float* comp_fea;           /* [Nc, fea] */
  int64_t* comp_self;        /* [Ec] */
  int64_t* comp_nbr;         /* [Ec] */
  int64_t* comp_crystal;     /* [Nc] */
} cgat_collated;
/* ids[B]: crystals of the batch in order; node_off/comp_off/cedge_off[B]: exclusive prefix sums of their atom
 * counts, unique-element counts and u*(u-1) composition edges (all device pointers). */
int cgat_collate_batch(const cgat_packed_dataset* ds, const int32_t* ids, const int32_t* node_off,
                       const int32_t* comp_off, const int32_t* cedge_off, int32_t B, int64_t E_total, int64_t Ec_total,
                       const cgat_collated* out, void* stream);

/* Gradient of a small embedding table looked up once per edge (reference CGAT/CGAT.py: nbr_embedding =
 * nn.Embedding(neighbor_number + 1, nbr_embedding_size), CGAtNet.forward): g_table[k, :] = sum_{t: idx[t] == k} g[t, :].
 * Deterministic (no atomics).  idx: int64 [rows]; C <= 128, K * C <= 8192. */
size_t cgat_embedding_backward_workspace_bytes(int32_t K, int32_t C);
int cgat_embedding_backward(const float* g, int64_t ldg, const int64_t* idx, int64_t rows, int32_t K, int32_t C,
                            float* g_table, void* ws, size_t ws_bytes, void* stream);

/* ---- fused optimiser steps and robust losses (after the hot path) ------------
 * One launch over all parameter tensors.  `table[n_tensors]` (device) holds the tensors; the chunk list cuts them
 * into pieces of cgat_mt_chunk_elems() elements: chunk c covers table[chunk_tensor[c]] from element chunk_off[c];
 * the chunks of a tensor are consecutive and first_chunk[n_tensors+1] indexes them (LAMB's per-tensor norms).
 *   cgat_adamw_step : torch.optim.AdamW as the reference constructs it (CGAT/lightning_module.py:328-331):
 *                     p *= 1 - lr*wd;  m = lerp(m, g, 1-b1);  v = b2 v + (1-b2) g^2;
 *                     p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 *   cgat_lamb_step  : CGAT/lambs.py:155-181 lamb_kernel (JITLamb.step 226-262): no bias correction, weight norm
 *                     clamped to [0,10], trust ratio |w|/(|s|+eps) (1 when either norm is 0), p -= lr*ratio*s,
 *                     s = m/(sqrt(v)+eps) + wd*p.  ws: 2*n_chunks + n_tensors floats.
 *   cgat_robust_loss: CGAT/utils.py:30-47 RobustL1 (kind 1) / RobustL2 (kind 2): per-row terms and the gradients
 *                     of the terms wrt output and log_std (the caller takes the mean). */
typedef struct cgat_mt_tensor {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
} cgat_mt_tensor;
int32_t cgat_mt_chunk_elems(void);
int cgat_adamw_step(const cgat_mt_tensor* table, const int32_t* chunk_tensor, const int64_t* chunk_off, int32_t n_chunks,
                    float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, void* stream);
int cgat_lamb_step(const cgat_mt_tensor* table, const int32_t* chunk_tensor, const int64_t* chunk_off, int32_t n_chunks,
                   const int32_t* first_chunk, int32_t n_tensors, float lr, float beta1, float beta2, float eps,
                   float weight_decay, float* ws, void* stream);
int cgat_robust_loss(const float* output, const float* log_std, const float* target, int32_t n, int32_t kind,
                     float* loss_terms, float* g_output, float* g_log_std, void* stream);

/* ---- GATConvNodes message + softmax + aggregate (scalar attention) ----------------------
 * replaces CGAT/CGAT.py:319-329: m=cat[x_i,edge_attr,x_j]; alpha=softmax_dst(MH_A(m));
 * aggr = scatter_add(MH_M(m)*alpha).mean(heads), with MH_* = MultiHeadNetwork (CGAT.py:65-112).
 * Parameter tensors are exactly the reference state_dict tensors (Conv1d weights
 * [H*Hd, D, 1] / [H*out, Hd, 1] are passed as the same contiguous memory). */
typedef struct cgat_attn_params {
  int32_t C, Ce, H, Hd; /* D = 2C + Ce, columns of *_in_w ordered [x_i | edge_attr | x_j] */
  const float* A_in_w;  /* [H*Hd, D]  MH_A.fc_in.weight  */
  const float* A_in_b;  /* [H*Hd]                         */
  const float* A_out_w; /* [H, Hd]    MH_A.fc_out.weight */
  const float* A_out_b; /* [H]                            */
  const float* M_in_w;  /* [H*Hd, D]  MH_M.fc_in.weight  */
  const float* M_in_b;  /* [H*Hd]                         */
  const float* M_out_w; /* [H*C, Hd]  MH_M.fc_out.weight */
  const float* M_out_b; /* [H*C]                          */
} cgat_attn_params;
typedef struct cgat_attn_grads {
  float *A_in_w, *A_in_b, *A_out_w, *A_out_b, *M_in_w, *M_in_b, *M_out_w, *M_out_b;
} cgat_attn_grads;
/* saved-for-backward: `saved` holds Z[E,2*H*Hd] | alpha[E,H] | S[N,H*Hd] | ssum[N,H] */
size_t cgat_nodes_attention_saved_floats(int32_t N, int32_t E, int32_t H, int32_t Hd);
size_t cgat_nodes_attention_forward_workspace_bytes(const cgat_plan* plan, const cgat_attn_params* p);
size_t cgat_nodes_attention_backward_workspace_bytes(const cgat_plan* plan, const cgat_attn_params* p);
int cgat_nodes_attention_forward(const cgat_plan* plan, const cgat_attn_params* p, const float* x /* [N,C] */,
                                 const float* edge_attr /* [E,Ce], original edge order */,
                                 float* aggr /* out [N,C] = head-mean of the aggregated messages */, float* saved,
                                 void* ws, size_t ws_bytes, void* stream);
int cgat_nodes_attention_backward(const cgat_plan* plan, const cgat_attn_params* p, const float* x,
                                  const float* edge_attr, const float* saved, const float* g_aggr /* [N,C] */,
                                  float* g_x /* [N,C] */, float* g_edge_attr /* [E,Ce] */, const cgat_attn_grads* g,
                                  void* ws, size_t ws_bytes, void* stream);

/* Debug / parity instrumentation (tests only, not on the hot path): the LeakyReLU derivative pattern the backward of
 * cgat_nodes_attention_forward will use -- mask[e, c] = (Z[slot(e), c] > 0) for the pre-activations of MH_A (columns
 * [0, H*Hd)) and MH_M ([H*Hd, 2*H*Hd)) of CGAT/CGAT.py:96,105-108, in ORIGINAL edge order.  LeakyReLU's derivative
 * jumps at 0 (CGAT.py:95, slope 0.01), so two correct fp32 evaluations can disagree on elements with |z| ~ 1e-7 max|z|
 * and then differ in every upstream gradient by a finite amount; the parity tests force the oracle's derivative
 * pattern to THIS one and compare at the flat tolerance.  fp32 edge storage only. */
int cgat_debug_nodes_attention_signs(const cgat_plan* plan, const cgat_attn_params* p, const float* saved,
                                     uint8_t* mask /* out [E, 2*H*Hd] */, void* stream);

/* ---- first layer of the message networks alone (vector-attention variants) ---------------
 * hidden[t, :] = LeakyReLU(w_in [x_i ; edge_attr ; x_j] + b_in), t = destination-sorted edge slot (plan.dst_perm),
 * for the stacked first-layer weights w_in [W2, 2C+Ce] of any number of heads / networks: MultiHeadNetwork's
 * repeat + grouped Conv1d + LeakyReLU (CGAT/CGAT.py:96,105-108) on m = cat[x_i, edge_attr, x_j] (CGAT.py:316-318),
 * computed with the operand split and never materialising m.  The channel-wise attention of
 * vector_attention=True (CGAT.py:286-290) applies the second layers and the softmax to `hidden`. */
size_t cgat_edge_hidden_forward_workspace_bytes(const cgat_plan* plan, int32_t C, int32_t Ce, int32_t W2);
size_t cgat_edge_hidden_backward_workspace_bytes(const cgat_plan* plan, int32_t C, int32_t Ce, int32_t W2);
int cgat_edge_hidden_forward(const cgat_plan* plan, int32_t C, int32_t Ce, int32_t W2, const float* w_in /* [W2,2C+Ce] */,
                             const float* b_in /* [W2] */, const float* x /* [N,C] */, const float* edge_attr /* [E,Ce] */,
                             float* hidden /* [E,W2] */, float* hidden_absmax /* out [1], optional: max |hidden| */,
                             void* ws, size_t ws_bytes, void* stream);
/* g_is_pre != 0: g_hidden already is the gradient of the pre-activation (made by cgat_linear_backward_dact, which folds
 * LeakyReLU' into the product) and gpre_absmax[0] (device, optional) its maximum: no elementwise pass over [E, W2] */
int cgat_edge_hidden_backward(const cgat_plan* plan, int32_t C, int32_t Ce, int32_t W2, const float* w_in, const float* x,
                              const float* edge_attr, const float* hidden, const float* g_hidden, int32_t g_is_pre,
                              const float* gpre_absmax, float* g_x, float* g_edge_attr, float* g_w_in, float* g_b_in,
                              void* ws, size_t ws_bytes, void* stream);

/* ---- H_Net_0 / H_Net: hypernetwork Pooling_NN -------------------------------------------
 * replaces CGAT/Hypernetworksmp.py:257-313 (HyperFC of n_hyper predicted layers, each with its
 * own FCBlock trunk of n_fc Linear+Tanh and a Linear(W -> W*W+W) head; LayerNorm(no affine,
 * eps 1e-5)+Tanh after every predicted layer but the last).  All widths equal W, as
 * CGAT.py:301-305 constructs them.  damping == NULL selects H_Net_0 (hyper input = h0);
 * otherwise hyper input = d*h0 + (1-d)*v with d = *damping (already clamped by the caller,
 * Hypernetworksmp.py:310-311). */
typedef struct cgat_hyperlinear_params {
  const float* fc_w[CGAT_MAX_FC]; /* [W,W]      hypo_params.net.{s}.net.0.weight */
  const float* fc_b[CGAT_MAX_FC]; /* [W]                                          */
  const float* head_w;            /* [W*W+W, W] hypo_params.net.{n_fc}.weight    */
  const float* head_b;            /* [W*W+W]                                      */
} cgat_hyperlinear_params;
typedef struct cgat_hnet_params {
  int32_t W, n_fc, n_hyper;
  cgat_hyperlinear_params layer[CGAT_MAX_HYPER];
  const float* damping; /* device scalar or NULL */
} cgat_hnet_params;
typedef struct cgat_hyperlinear_grads {
  float* fc_w[CGAT_MAX_FC];
  float* fc_b[CGAT_MAX_FC];
  float* head_w;
  float* head_b;
} cgat_hyperlinear_grads;
typedef struct cgat_hnet_grads {
  cgat_hyperlinear_grads layer[CGAT_MAX_HYPER];
  float* damping; /* [1] or NULL */
} cgat_hnet_grads;
size_t cgat_hnet_saved_floats(int32_t rows, const cgat_hnet_params* p);
size_t cgat_hnet_forward_workspace_bytes(int32_t rows, const cgat_hnet_params* p);
size_t cgat_hnet_backward_workspace_bytes(int32_t rows, const cgat_hnet_params* p);
int cgat_hnet_forward(int32_t rows, const cgat_hnet_params* p, const float* h0 /* [rows,W] */,
                      const float* v /* [rows,W] */, float* y /* [rows,W] */, float* saved, void* ws, size_t ws_bytes,
                      void* stream);
int cgat_hnet_backward(int32_t rows, const cgat_hnet_params* p, const float* h0, const float* v, const float* saved,
                       const float* g_y, float* g_h0, float* g_v, const cgat_hnet_grads* g, void* ws, size_t ws_bytes,
                       void* stream);

/* Same backward with the four weight-gradient contractions (grad of head_w[0 : W*W]) issued on `side_stream`: they
 * feed nothing else in the pass and are matrix-core bound, so they can run beside the HBM-bound kernels that follow
 * on `stream`.  The caller owns `side_ws` (cgat_hnet_backward_side_workspace_bytes) and must keep it, `saved`, `v`,
 * `g_y` and the head_w gradient buffers alive -- and must not read those gradients -- until `side_stream` has drained. */
size_t cgat_hnet_backward_side_workspace_bytes(int32_t rows, const cgat_hnet_params* p);
int cgat_hnet_backward_overlapped(int32_t rows, const cgat_hnet_params* p, const float* h0, const float* v,
                                  const float* saved, const float* g_y, float* g_h0, float* g_v, const cgat_hnet_grads* g,
                                  void* ws, size_t ws_bytes, void* stream, void* side_ws, size_t side_ws_bytes,
                                  void* side_stream);

/* ---- dense layer  y = act(x W^T + b)  (nn.Linear / 1x1 Conv1d group + activation) --------
 * replaces the Linear/LeakyReLU/ReLU/Tanh pairs of CGAT/message_changed.py:58-63,124-130,
 * CGAT/roost_message.py:348-352 and one head of MultiHeadNetwork (CGAT/CGAT.py:103-109).
 * act: 0 none, 1 tanh, 2 LeakyReLU(0.01), 3 ReLU.  ldx/ldw/ldy are row strides in floats. */
size_t cgat_linear_forward_workspace_bytes(int32_t M, int32_t K, int32_t N);
size_t cgat_linear_backward_workspace_bytes(int32_t M, int32_t K, int32_t N);
/* ws may be NULL (then the generic f32 GEMM engine is used for every shape).  x_absmax (optional, device pointer to
 * max |x| over the tensor x is a slice of, e.g. cgat_edge_hidden_forward's hidden_absmax): lets the K > 128 -> 128 route
 * run in the f16x3 form (three matrix passes) instead of the scale-free six-pass bf16 form */
int cgat_linear_forward(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y,
                        int64_t ldy, int32_t M, int32_t K, int32_t N, int32_t act, const float* x_absmax, void* ws,
                        size_t ws_bytes, void* stream);
/* The backward of a bias-free-activation nn.Linear whose INPUT x is itself a LeakyReLU(0.01) output (the hidden layer
 * of MultiHeadNetwork, CGAT/CGAT.py:96-98): g_x = (g_y W) * LeakyReLU'(sign of gx_dact), i.e. the gradient of the
 * pre-activation behind x, with max |g_x| folded into gx_absmax[0] (device, caller-zeroed, optional); g_w, g_b as in
 * cgat_linear_backward.  Needs N == 128 and K a multiple of 128 in a split arithmetic mode (CGAT_ERR_UNSUPPORTED
 * otherwise: use cgat_linear_backward and an elementwise pass). */
int cgat_linear_backward_dact(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* g_y, int64_t ldgy,
                              float* g_x, int64_t ldgx, const float* gx_dact, int64_t ld_dact, float* gx_absmax,
                              float* g_w, int64_t ldgw, float* g_b, int32_t M, int32_t K, int32_t N, void* ws,
                              size_t ws_bytes, void* stream);
/* The H per-head second layers of one MultiHeadNetwork in ONE call (CGAT/CGAT.py:97-98, 103-109: fc_out is a grouped
 * Conv1d = H independent Linear(K, N) on the H column blocks of the hidden matrix): head h works on x + h * s_x,
 * w + h * s_w, bias + h * s_bias, y + h * s_y (strides in elements).  Same results as H calls of cgat_linear_forward /
 * cgat_linear_backward_dact, which is also what runs for shapes the batched kernels do not take; batched (f16x3 mode,
 * N == 128, K a multiple of 128, contiguous per-head weights) the layer is 3 launches instead of 4-5 per head. */
size_t cgat_heads_linear_forward_workspace_bytes(int32_t M, int32_t K, int32_t N, int32_t H);
int cgat_heads_linear_forward(const float* x, int64_t ldx, int64_t s_x, const float* w, int64_t ldw, int64_t s_w,
                              const float* bias, int64_t s_bias, float* y, int64_t ldy, int64_t s_y, int32_t M, int32_t K,
                              int32_t N, int32_t H, const float* x_absmax, void* ws, size_t ws_bytes, void* stream);
size_t cgat_heads_linear_backward_dact_workspace_bytes(int32_t M, int32_t K, int32_t N, int32_t H);
int cgat_heads_linear_backward_dact(const float* x, int64_t ldx, int64_t s_x, const float* w, int64_t ldw, int64_t s_w,
                                    const float* g_y, int64_t ldgy, int64_t s_gy, float* g_x, int64_t ldgx, int64_t s_gx,
                                    const float* gx_dact, int64_t ld_dact, int64_t s_dact, float* gx_absmax, float* g_w,
                                    int64_t ldgw, int64_t s_gw, float* g_b, int64_t s_gb, int32_t M, int32_t K, int32_t N,
                                    int32_t H, void* ws, size_t ws_bytes, void* stream);
/* gpre = g_y * act'(y) is written to `gpre` [M,N] (caller buffer); g_x += or = per accumulate_gx */
int cgat_linear_backward(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* y, int64_t ldy,
                         const float* g_y, int64_t ldgy, float* gpre /* [M,N] dense */, float* g_x, int64_t ldgx,
                         int32_t accumulate_gx, float* g_w, int64_t ldgw, float* g_b, int32_t M, int32_t K, int32_t N,
                         int32_t act, void* ws, size_t ws_bytes, void* stream);

/* ---- segment softmax / segment sums over CSR-ordered rows --------------------------------
 * replaces torch_geometric.utils.softmax (CGAT.py:59,323; eps 1e-16), the weighted softmax of
 * roost_message.py:307-311 (mult = w**pow, eps 1e-13) and torch_scatter.scatter_add
 * (CGAT.py:60, roost_message.py:315).  Rows must be ordered by segment (rowptr[S+1]). */
int cgat_segment_softmax_forward(const float* a /* [R,F] */, const float* mult /* [R] or NULL */,
                                 const int32_t* rowptr, int32_t S, int32_t F, float eps, float* alpha /* [R,F] */,
                                 void* stream);
int cgat_segment_softmax_backward(const float* alpha, const float* g_alpha, const float* mult, const int32_t* rowptr,
                                  int32_t S, int32_t F, float* g_a, float* g_mult /* [R] or NULL, F==1 */,
                                  void* stream);
/* out[s,f] = sum_{r in seg s} x[ridx ? ridx[r] : r, f] */
int cgat_segment_sum(const float* x, int64_t ldx, const int32_t* ridx, const int32_t* rowptr, int32_t S, int32_t F,
                     float* out, int64_t ldo, void* stream);

/* Softmax-weighted segment sum ("attention pooling") in one pass per direction, rows in CSR order:
 *   out[s,f] = sum_{r in seg s} alpha[r, f/fw] * m[r,f],  fw = F / aF,
 *   alpha[r,c] = mult[r] * exp(a[r,c] - max_seg a[.,c]) / (sum_seg mult * exp(..) + eps)      (mult may be NULL)
 * = torch_geometric softmax (+1e-16) -> multiply -> scatter_add of reference CGAT.py:323-329 (vector attention:
 * aF = F), CGAT.py:59-61 (MHAttention: aF = heads) and roost_message.py:305-317 (WeightedAttention: aF = 1,
 * mult = weights ** pow, eps = 1e-13).  mx / inv [S, aF] (segment maximum, the denominator sum + eps: alpha is ONE division, as in the reference) are what backward needs
 * instead of alpha; out_lo [S, F] (may be NULL) receives the low part of the sum, which is accumulated in fp64: backward
 * centres every row on out + out_lo, so that the rounding of `out` does not enter g_a as an error common to all rows of
 * a segment.  Row r of a segment is row ridx[r] of a / mult / m (and of the gradients) when ridx != NULL.  Needs F % 4 == 0 and aF == F (aF % 4 == 0) or F / aF in {4, 8, .., 256} (a power of two). */
int cgat_segment_attention_pool_forward(const float* a, int32_t aF, const float* mult, const float* m, int64_t ldm,
                                        const int32_t* rowptr, const int32_t* ridx, int32_t S, int32_t F, float eps, float* out, float* mx,
                                        float* inv, float* out_lo, void* stream);
/* g_m[r,f] = alpha * g_out[s,f];  g_a[r,c] = sum_{f in c} alpha * g_out * (m - out - out_lo);  g_mult[r] = g_a[r,0] / mult[r]
 * (aF == 1; g_m / g_mult / out_lo may be NULL) */
int cgat_segment_attention_pool_backward(const float* a, int32_t aF, const float* mult, const float* m, int64_t ldm,
                                         const int32_t* rowptr, const int32_t* ridx, int32_t S, int32_t F, const float* out, const float* mx,
                                         const float* inv, const float* out_lo, const float* g_out, float* g_a, float* g_m,
                                         int64_t ldgm, float* g_mult, void* stream);

/* A chain of up to 5 dense layers of width 128 in ONE launch (the split arithmetic modes f16x3, f16x3c, bf16x6;
 * CGAT_ERR_UNSUPPORTED in the f32 mode; the workspace holds one prepared weight image of 24576 floats per layer):
 *   r_0 = x                      (times act'(in_dact) if in_dact != NULL; stored to in_store if != NULL)
 *   r_(l+1) = act_l(r_l W_l^T + bias_l) (+ resid_l) (* dact_type_l'(dact_l)),   W_l(o,k) = W[o*w_so + k*w_sk]
 * every r_(l+1) is written to out_l when out_l != NULL (+= when accumulate).  Forward of the hypernetwork trunks
 * (reference Hypernetworksmp.py:36-83: four Linear+Tanh, then the linear terms of the predicted layer), their backward
 * (the same chain on the transposed weights with the tanh derivatives as in_dact / dact), and the edge update's
 * two-layer network with its residual (CGAT.py:226-229, 580-585).  Rows stay in registers between layers; only the
 * requested outputs touch HBM.  All row pointers 16-byte aligned, leading dimensions multiples of 4. */
typedef struct cgat_chain_layer {
  const float* W;
  int64_t w_so, w_sk;
  const float* bias;      /* [128] or NULL */
  const float* dact;      /* NULL, or saved activation values [rows,128]: result *= act'(dact) with act = dact_type */
  int64_t ld_dact;
  const float* resid;     /* NULL, or [rows,128] added to the activated result */
  int64_t ld_resid;
  float* out;             /* NULL, or destination [rows,128] */
  int64_t ld_out;
  int32_t act;            /* 0 none, 1 tanh, 2 LeakyReLU(0.01), 3 ReLU */
  int32_t dact_type;
  int32_t accumulate;
} cgat_chain_layer;
typedef struct cgat_chain_desc {
  int32_t n_layers, rows;
  const float* x;
  int64_t ldx;
  const float* in_dact;
  int64_t ld_in_dact;
  int32_t in_dact_type;
  float* in_store;
  int64_t ld_in_store;
  cgat_chain_layer layer[5];
} cgat_chain_desc;
size_t cgat_mlp_chain_workspace_bytes(int32_t n_layers);
int cgat_mlp_chain(const cgat_chain_desc* d, void* ws, size_t ws_bytes, void* stream);

/* Weight (and bias) gradients of many width-128 dense layers in ONE launch: for item i < n
 *     out[i][o][k] = sum_r G[i][r,o] X[i][r,k]   (128 x 128, row stride ldo),   bsum[i][o] = sum_r G[i][r,o]  (or NULL)
 * i.e. what autograd accumulates into nn.Linear.weight / .bias of the hypernetwork's trunk layers and linear terms
 * (Hypernetworksmp.py:24-60, 77-83) and of the per-head second layers (CGAT.py:103-109): exact fp32 products
 * (f32-input MFMA), fixed summation order.  All items share rows, ldg, ldx, ldo; n <= 32; G, X 16-byte aligned. */
size_t cgat_dense_wgrad_batch_workspace_bytes(int32_t n, int32_t rows);
int cgat_dense_wgrad_batch(int32_t n, const float* const* G, int64_t ldg, const float* const* X, int64_t ldx,
                           float* const* out, int64_t ldo, float* const* bsum, int32_t rows, void* ws, size_t ws_bytes,
                           void* stream);

/* ---- small-row dense-layer programs: a whole G-row / Nc-row network per launch ------------------------------------
 * At the batch the reference harness ships (--batch-size 64, CGAT/lightning_module.py:468-473) the output head
 * (ResidualNetwork, CGAT/message_changed.py:81-138: per layer act(fc(x)) + res_fc(x), then fc_out), Roost's gate and
 * message networks (SimpleNetwork, CGAT/roost_message.py:137-153, 324-355) and the per-crystal networks of MHAttention
 * (CGAT/CGAT.py:14-62) are products over 64 ... 2 048 rows: bound by kernel boundaries, not by flops.  A program is a
 * list of products
 *     out[m,n] = act( alpha sum_k A'(m,k) B0(n,k) + bias[n] ) [-> h_out[m,n]]  +  sum_k A(m,k) B1(n,k)  +  resid[m,n]  +  beta out[m,n]
 *     A'(m,k)  = A(m,k) * dact_type'(dact(m,k))     (dact = saved activation VALUES; NULL: A' = A)
 *     rowsum[m] = sum_k A'(m,k)                       (optional: a bias gradient)
 * with X(r,k) = X[r * x_rs + k * x_ks] for A, dact, B0, B1 -- so one op is a forward layer with its residual product
 * (A = x, B0 = fc.weight, B1 = res_fc.weight), an input gradient (A = g_y, dact = h, B0 = W^T, B1 = R^T) or a weight
 * gradient (A = g_y^T, dact = h^T, B0 = x^T, rowsum = the bias gradient) -- run by ONE persistent launch phase by
 * phase: the ops of a phase must not depend on each other, phase p may read what phases < p wrote (a grid barrier with
 * device-scope release / acquire separates them).  Exact fp32 products with fp32 accumulation (f32-input MFMA), fixed
 * summation order; no workspace, no operand images.  act / dact_type: 0 none, 1 tanh, 2 LeakyReLU(0.01), 3 ReLU.
 * B1, bias, resid, h_out, rowsum, dact may be NULL.  Meant for M, N <= a few thousand.
 * sync_words: CGAT_ROWPROG_SYNC_WORDS uint32 of device memory, 64-byte aligned, zero-filled by the caller ONCE and then
 * left to the library for the life of the process (barrier counters that every launch returns to zero, and one sticky
 * word -- index CGAT_ROWPROG_SYNC_WORDS - 16 -- that is set to 1 if a barrier ever gave up instead of hanging: the
 * results of that launch are void). */
#define CGAT_ROWPROG_MAX_OPS 24
#define CGAT_ROWPROG_SYNC_WORDS ((1024 + 1) * 16)
typedef struct cgat_rowprog_op {
  int32_t phase;            /* non-decreasing over the op list, starting at 0, no gaps */
  int32_t M, N, K;
  const float* A;     int64_t a_rs, a_ks;
  const float* dact;  int64_t d_rs, d_ks;  int32_t dact_type;
  const float* B0;    int64_t b0_rs, b0_ks;
  const float* B1;    int64_t b1_rs, b1_ks;
  const float* bias;  /* [N] */
  int32_t act;
  const float* resid; int64_t ld_resid;
  float* out;         int64_t ldo;
  float alpha, beta;  /* beta == 0: out is not read */
  float* h_out;       int64_t ld_h;
  float* rowsum;      /* [M] */
} cgat_rowprog_op;
typedef struct cgat_rowprog {
  int32_t n_ops;
  cgat_rowprog_op op[CGAT_ROWPROG_MAX_OPS];
} cgat_rowprog;
int cgat_rowprog_run(const cgat_rowprog* prog, uint32_t* sync_words, void* stream);

/* Storage of the per-edge intermediates of cgat_nodes_attention_*: the pre-activations Z saved by forward, and their
 * gradient gZ inside backward -- which at the benchmark widths is NOT stored at all but rebuilt by its consumers from
 * one sign bit per element and per-node rows (same values to fp32 rounding).
 *   0 = fp32 Z (default);
 *   1 = bf16 Z ("bf16 activations" of BASELINE configs[4]): halves the bytes of the Z-sized passes; attention logits,
 *       softmax statistics, sums and every matrix product stay as they are (fp32 accumulation); tolerance of that mode
 *       1e-2 max-norm relative.  Exists for the scalar-attention node layer at C = Ce = 128, Hd a multiple of 128, in every
 *       split arithmetic mode (f16x3c, bf16x6, f16x3); cgat_nodes_attention_forward / _backward return
 *       CGAT_ERR_UNSUPPORTED for a layer without that form (other widths, the f32 mode) instead of running it in fp32
 *       storage under the bf16 label (round 4's library ignored the switch there);
 *   2 = fp32 Z with gZ stored in backward (round 1's path, 6 KB more workspace per edge): the A/B reference of the tests.
 * A backward call must run under the mode its forward ran under.  Env CGAT_EDGE_STORAGE = bf16 | f32+gz sets the start
 * value. */
void cgat_set_edge_storage(int32_t mode);
int32_t cgat_get_edge_storage(void);

/* ---- kernel-level primitives (what the layer entry points above are composed of) --------- */
/* C = act(alpha * A.B + bias + add1[add1_idx[m]] + add2[add2_idx[m]]) + beta * C on the fp32 matrix
 * cores.  A(m,k) = A[row(m)*lda + k] (a_kmajor=0, row(m)=a_rgather?a_rgather[m]:m) or A[k*lda + m]
 * (a_kmajor=1); B(k,n) = B[n*ldb + k] (b_kmajor=0, i.e. a torch Linear weight) or
 * B[krow(k)*ldb + n] (b_kmajor=1, krow(k)=b_kgather?b_kgather[k]:k); output row = c_scatter?c_scatter[m]:m.
 * splits > 1 splits K over workgroups (deterministic slab reduction; needs workspace). */
typedef struct cgat_gemm_desc {
  int32_t M, N, K;
  const float* A;
  int64_t lda;
  int32_t a_kmajor;
  const int32_t* a_rgather;
  const float* B;
  int64_t ldb;
  int32_t b_kmajor;
  const int32_t* b_kgather;
  float* C;
  int64_t ldc;
  const int32_t* c_scatter;
  float alpha, beta;
  const float* bias;
  const float* add1;
  const int32_t* add1_idx;
  const float* add2;
  const int32_t* add2_idx;
  int64_t ld_add;
  int32_t act;
  int32_t splits; /* 0 = choose automatically */
  int64_t a_block; /* != 0: A stored as 128-wide column blocks [cols/128][rows][128], block stride in floats;
                      the blocked dimension is k (a_kmajor=0) or m (a_kmajor=1); lda must be 128 */
} cgat_gemm_desc;
size_t cgat_gemm_workspace_bytes(const cgat_gemm_desc* d);
int cgat_gemm(const cgat_gemm_desc* d, void* ws, size_t ws_bytes, void* stream);
/* out[n,c] = init[n,c] + sum_{a<NA,b<NB} p[n,a] q[n,b] T[(a*NB+b)*NC + c]   (init may be NULL or == out).
 * The workspace holds the kernel-side re-layout of T and the partial slabs of the a-split. */
size_t cgat_bilinear_rows_workspace_bytes(int32_t rows, int32_t NA, int32_t NB, int32_t NC);
int cgat_bilinear_rows(const float* p, int64_t ldp, const float* q, int64_t ldq, const float* T, const float* init,
                       int64_t ldi, float* out, int64_t ldo, int32_t rows, int32_t NA, int32_t NB, int32_t NC,
                       void* ws, size_t ws_bytes, void* stream);
/* Two gradients of the hypernetwork's trilinear form from ONE contraction (backward of reference
 * Hypernetworksmp.py:77-83, HyperLinear.forward; widths fixed at 128):
 *   out1[n,c] = init1[n,c] + sum_{a,b} p[n,a] q[n,b] T[(a*128+b)*128 + c]
 *   out2[n,a] = init2[n,a] + sum_{b,c} zz[n,c] q[n,b] T[(a*128+b)*128 + c]
 * (init1/init2 may be NULL or alias their outputs).  In the f32 arithmetic mode it runs as two contractions. */
size_t cgat_bilinear_dual_workspace_bytes(int32_t rows);
int cgat_bilinear_dual(const float* p, int64_t ldp, const float* q, int64_t ldq, const float* zz, int64_t ldz,
                       const float* T, const float* init1, int64_t ldi1, float* out1, int64_t ldo1, const float* init2,
                       int64_t ldi2, float* out2, int64_t ldo2, int32_t rows, void* ws, size_t ws_bytes, void* stream);
/* Arithmetic of the width-128 matrix-core kernels: 2 (default, "f16x3") = operands scaled by a power of two and
 * split into two fp16 pieces, three fp16-MFMA passes with fp32 accumulation (measured at the error of an fp32 product
 * chain); 6 ("bf16x6") = three bf16 pieces, six bf16-MFMA passes (same accuracy, twice the matrix work); 3 = three bf16
 * passes (~4e-6 relative, fails the parity tests: diagnostic only); 0 = f32-input MFMA (exact fp32 fmaf chains) and the
 * f32 GEMM engine for the edge products.  Env CGAT_BILINEAR_MODE = f16x3 | bf16x6 | bf16x3 | f32 sets the start value. */
void cgat_set_bilinear_mode(int32_t mode);
int32_t cgat_get_bilinear_mode(void);
/* out[(a*NB+b)*NC + c] = sum_n p[n,a] q[n,b] r[n,c] */
size_t cgat_bilinear_wgrad_workspace_bytes(int32_t rows, int32_t NA, int32_t NB, int32_t NC);
int cgat_bilinear_wgrad(const float* p, int64_t ldp, const float* q, int64_t ldq, const float* r, int64_t ldr,
                        float* out, int32_t rows, int32_t NA, int32_t NB, int32_t NC, void* ws, size_t ws_bytes,
                        void* stream);
/* y = tanh(LayerNorm(u)) without affine, biased variance (Hypernetworksmp.py:103-107) and its backward */
int cgat_layernorm_tanh_forward(const float* u, float* y, int32_t rows, int32_t W, float eps, void* stream);
int cgat_layernorm_tanh_backward(const float* u, const float* y, const float* g_y, float* g_u, int32_t rows, int32_t W,
                                 float eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CGAT_HIP_H */
