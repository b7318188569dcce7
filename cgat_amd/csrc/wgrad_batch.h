// Descriptors shared by the batched weight-gradient contractions (bilinear.hip: f16x3 form; wgradc.hip: f16x3c form).
#pragma once
#include "common.h"

#define WGB_MAX 8
struct WgradBatchDesc {
  float* out[WGB_MAX];   // final [NA][128][128] outputs (splits == 1)
  float* slab;           // [layer][split][NA][128][128] partial sums (splits > 1)
  long sT, sR;           // per-layer strides: floats of pT / qT (qF); uint4 of Rq (f16x3) or bytes of the r stream (f16x3c)
  int n_layers, splits, npairs, NA, rows_pad, rows_per_split;
};
struct WgradPrepDesc {
  const float* p[WGB_MAX];
  const float* q[WGB_MAX];
  const float* r[WGB_MAX];
};

// ---- f16x3c form (wgradc.hip) ----
int wgradc_pick(int n_layers, int nrows, int NA, int* rps_out);
size_t wgradc_ws_bytes(int n_layers, int nrows, int NA);
// operand preparation of layers [l0, l0 + n) of an n_layers batch
int wgradc_prep(int l0, int n, int n_layers, const float* const* p, long ldp, const float* const* q, long ldq,
                const float* const* r, long ldr, int nrows, int NA, void* ws, size_t ws_bytes, hipStream_t stream);
int wgradc_launch(int n_layers, const float* const* p, long ldp, const float* const* q, long ldq, const float* const* r,
                  long ldr, float* const* out, int nrows, int NA, void* ws, size_t ws_bytes, hipStream_t stream,
                  int max_wgs, bool prepared);
