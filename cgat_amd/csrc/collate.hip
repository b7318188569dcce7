// Batch collation on the device: builds, for a list of crystal ids, exactly the tensors the reference
// assembles on the host every step with per-graph Python loops --
//   CGAT/data.py:61-144          CompositionData.__getitem__  (embedding rows per atom, neighbour tables sliced to
//                                max_neighbor_number and flattened atom-major, composition graph, y = target * n_atoms)
//   PyG Batch.from_data_list     (CGAT/lightning_module.py:200: concatenate, offset edge_index, `batch` vector)
//   CGAT/roost_message.py:400-458 collate_batch  (offset composition indices, crystal index per composition node)
// from a packed, device-resident int32 form of the dataset (cgat_packed_dataset, built once by
// cgat_amd.collate.PackedDataset).  One workgroup per crystal; integer/index work and exact fp32 row gathers, so the
// outputs are bit-identical to the reference's.  HBM-bound: 800 B of embedding row per atom dominates.
#include "../../include/cgat_hip.h"
#include "common.h"
#include "kernels.h"

__global__ __launch_bounds__(256) void collate_kernel(cgat_packed_dataset ds, const int32_t* __restrict__ ids,
                                                      const int32_t* __restrict__ node_off,
                                                      const int32_t* __restrict__ comp_off,
                                                      const int32_t* __restrict__ cedge_off, int B, long E_total,
                                                      long Ec_total, cgat_collated out) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int g = ids[b];
  const int a0 = ds.atom_ptr[g], na = ds.atom_ptr[g + 1] - a0;
  const int n0 = node_off[b];
  const int F = ds.fea, K = ds.max_nbr;
  // x rows: gather of embedding rows, 16 bytes per thread step
  const int F4 = F >> 2;
  if ((F & 3) == 0) {
    for (int i = tid; i < na * F4; i += 256) {
      const int a = i / F4, c = i - a * F4;
      const float4 v = reinterpret_cast<const float4*>(ds.table + (long)ds.atom_elem[a0 + a] * F)[c];
      reinterpret_cast<float4*>(out.x + (long)(n0 + a) * F)[c] = v;
    }
  } else {
    for (int i = tid; i < na * F; i += 256) {
      const int a = i / F, c = i - a * F;
      out.x[(long)(n0 + a) * F + c] = ds.table[(long)ds.atom_elem[a0 + a] * F + c];
    }
  }
  for (int a = tid; a < na; a += 256) out.batch[n0 + a] = b;
  // edges: atom-major flatten of the [n_atoms, K] tables; global index = local + nodes before this crystal
  const long e0 = (long)n0 * K;
  for (int i = tid; i < na * K; i += 256) {
    const long src = (long)a0 * K + i;
    out.edge_index[e0 + i] = (int64_t)ds.self_idx[src] + n0;
    out.edge_index[E_total + e0 + i] = (int64_t)ds.nbr_idx[src] + n0;
    out.edge_attr[e0 + i] = (int64_t)ds.shell[src];
  }
  if (tid == 0) out.y[b] = ds.y_val[g];
  // composition graph: unique elements in first-appearance order, fully connected without self loops
  const int u0 = ds.comp_ptr[g], nu = ds.comp_ptr[g + 1] - u0;
  const int c0 = comp_off[b];
  if ((F & 3) == 0) {
    for (int i = tid; i < nu * F4; i += 256) {
      const int u = i / F4, c = i - u * F4;
      const float4 v = reinterpret_cast<const float4*>(ds.table + (long)ds.comp_elem[u0 + u] * F)[c];
      reinterpret_cast<float4*>(out.comp_fea + (long)(c0 + u) * F)[c] = v;
    }
  } else {
    for (int i = tid; i < nu * F; i += 256) {
      const int u = i / F, c = i - u * F;
      out.comp_fea[(long)(c0 + u) * F + c] = ds.table[(long)ds.comp_elem[u0 + u] * F + c];
    }
  }
  for (int u = tid; u < nu; u += 256) {
    out.comp_weight[c0 + u] = ds.comp_weight[u0 + u];
    out.comp_crystal[c0 + u] = b;
  }
  const long ce0 = cedge_off[b];
  for (int i = tid; i < nu * (nu - 1); i += 256) {   // pair (u, j-th other element)
    const int u = i / (nu - 1), j = i - u * (nu - 1);
    out.comp_self[ce0 + i] = (int64_t)(c0 + u);
    out.comp_nbr[ce0 + i] = (int64_t)(c0 + (j < u ? j : j + 1));
  }
  (void)B; (void)Ec_total;
}

extern "C" int cgat_collate_batch(const cgat_packed_dataset* ds, const int32_t* ids, const int32_t* node_off,
                                  const int32_t* comp_off, const int32_t* cedge_off, int32_t B, int64_t E_total,
                                  int64_t Ec_total, const cgat_collated* out, void* stream) {
  CGAT_CHECK_ARG(ds && out, "collate_batch: null dataset/outputs");
  CGAT_CHECK_ARG(B >= 0 && ds->fea > 0 && ds->max_nbr >= 0, "collate_batch: bad sizes");
  if (B == 0) return CGAT_OK;
  CGAT_PROF("collate", (hipStream_t)stream);
  hipLaunchKernelGGL(collate_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, *ds, ids, node_off, comp_off,
                     cedge_off, B, (long)E_total, (long)Ec_total, *out);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}

// ---------------------------------------------------------------------------------------
// Gradient of a small embedding table (reference CGAT.py: nbr_embedding = nn.Embedding(neighbor_number + 1, Ce),
// looked up once per EDGE): g_table[k, :] = sum over rows t with idx[t] == k of g[t, :].  torch's backward sorts the
// indices and scatters with atomics (7.6 ms per step of the full stack at E = 1M for a 13-row table, and not
// deterministic); here every workgroup owns a contiguous range of rows and two private [K][C] tables in LDS (one per
// half of its threads, a thread owns one column: read-modify-write without races), the per-workgroup tables are then
// summed in fixed order.  K * C <= 8192, C <= 128.
// ---------------------------------------------------------------------------------------
#define EMB_PARTS 1024   // workgroups (4 per CU: the 13 KB tables leave room) = partial tables
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const float* __restrict__ g, long ldg,
                                                            const long* __restrict__ idx, long rows, int K, int C,
                                                            float* __restrict__ partial) {
  extern __shared__ float tab[];                 // [2][K][C]
  const int tid = threadIdx.x, half = tid >> 7, c = tid & 127;
  for (int i = tid; i < 2 * K * C; i += 256) tab[i] = 0.f;
  __syncthreads();
  const long per = (rows + gridDim.x - 1) / gridDim.x;
  const long r0 = (long)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
  float* mine = tab + (long)half * K * C;
  if (c < C) {
    // eight rows' loads in flight, then the eight read-modify-writes in row order (same sums as one row at a time,
    // which ran at 0.6 TB/s: every iteration waited for its own two loads)
    long t = r0 + half;
    for (; t + 14 < r1; t += 16) {
      long k[8];
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { k[u] = idx[t + 2 * u]; v[u] = g[(t + 2 * u) * ldg + c]; }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k[u] >= 0 && k[u] < K) mine[k[u] * C + c] += v[u];
    }
    for (; t < r1; t += 2) {
      const long k = idx[t];
      if (k >= 0 && k < K) mine[k * C + c] += g[t * ldg + c];
    }
  }
  __syncthreads();
  for (int i = tid; i < K * C; i += 256) partial[(long)blockIdx.x * K * C + i] = tab[i] + tab[K * C + i];
}
// fixed summation tree: 8 threads per output each sum a contiguous eighth of the partial tables in order, the eight
// sums are added in order (one thread per output over 512 tables was a 512-long chain of dependent loads: 119 us)
__global__ __launch_bounds__(256) void embedding_bwd_reduce_kernel(const float* __restrict__ partial, int nparts, int n,
                                                                   float* __restrict__ out) {
  __shared__ float part[8][32];
  const int o = threadIdx.x & 31, zg = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + o;
  const int per = (nparts + 7) / 8;
  const int p0 = zg * per, p1 = min(nparts, p0 + per);
  float s = 0.f;
  if (i < n) {
#pragma unroll 8
    for (int p = p0; p < p1; ++p) s += partial[(long)p * n + i];
  }
  part[zg][o] = s;
  __syncthreads();
  if (zg != 0 || i >= n) return;
  s = part[0][o];
#pragma unroll
  for (int g = 1; g < 8; ++g) s += part[g][o];
  out[i] = s;
}

extern "C" size_t cgat_embedding_backward_workspace_bytes(int32_t K, int32_t C) { return (size_t)EMB_PARTS * K * C * sizeof(float) + 256; }
extern "C" int cgat_embedding_backward(const float* g, int64_t ldg, const int64_t* idx, int64_t rows, int32_t K, int32_t C,
                                       float* g_table, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  CGAT_CHECK_ARG(K > 0 && C > 0 && C <= 128 && (long)K * C <= 8192, "embedding_backward: needs C <= 128 and K * C <= 8192");
  CGAT_CHECK_ARG(ws && ws_bytes >= cgat_embedding_backward_workspace_bytes(K, C), "embedding_backward: workspace too small");
  const int parts = rows <= 0 ? 0 : (rows < (long)EMB_PARTS * 64 ? (int)((rows + 63) / 64) : EMB_PARTS);
  if (parts > 0) {
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(parts), dim3(256), (size_t)2 * K * C * sizeof(float), s, g, (long)ldg,
                       (const long*)idx, (long)rows, K, C, (float*)ws);
    CGAT_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(embedding_bwd_reduce_kernel, dim3(cdiv((long)K * C, 32)), dim3(256), 0, s, (const float*)ws, parts, K * C,
                     g_table);
  CGAT_LAUNCH_CHECK();
  return CGAT_OK;
}
