// CSR plans: the destination-sorted edge order every layer of a batch reuses, forward and
// backward.  PyG's propagate keys both the softmax and the scatter-add on edge_index[1]
// (reference CGAT.py:313-326, node_dim=0, flow source_to_target); sorting once by that key
// turns the scatter into contiguous, atomics-free segment reductions.  A second CSR over the
// sorted positions keyed by edge_index[0] serves the transposed accumulation (grad wrt x_j).
//
// csr_from_keys: histogram (integer atomics) -> exclusive scan -> cursor fill -> per-segment
// ascending sort of the filled ids.  The final sort makes the order *stable* (ids ascending
// inside a segment), hence independent of atomic arrival order: the plan, and every
// floating-point sum that walks it, is deterministic.
#include "common.h"
#include "kernels.h"

__global__ void hist_kernel(const int* __restrict__ keys, int n, int S, int* __restrict__ count) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int k = keys[i];
  if (k >= 0 && k < S) atomicAdd(&count[k], 1);
}

// single-workgroup exclusive scan with a running carry; count[S] -> rowptr[S+1]
__global__ void exscan_kernel(const int* __restrict__ count, int S, int* __restrict__ rowptr) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < S; base += 1024) {
    int i = base + tid;
    int v = i < S ? count[i] : 0;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    int incl = x + woff + carry;
    if (i < S) rowptr[i] = incl - v;
    __syncthreads();
    if (tid == 1023) carry = incl;
    __syncthreads();
  }
  if (tid == 0) rowptr[S] = carry;
}

__global__ void fill_kernel_csr(const int* __restrict__ keys, int n, int S, const int* __restrict__ rowptr,
                                int* __restrict__ cursor, int* __restrict__ perm) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int k = keys[i];
  if (k < 0 || k >= S) return;
  int pos = atomicAdd(&cursor[k], 1);
  perm[rowptr[k] + pos] = i;
}

// ascending insertion sort of each segment's ids (segments are short: in-degree of an atom)
__global__ void sort_segments_kernel(const int* __restrict__ rowptr, int S, int* __restrict__ perm) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  int r0 = rowptr[s], r1 = rowptr[s + 1];
  for (int i = r0 + 1; i < r1; ++i) {
    int v = perm[i];
    int j = i - 1;
    while (j >= r0 && perm[j] > v) {
      perm[j + 1] = perm[j];
      --j;
    }
    perm[j + 1] = v;
  }
}

size_t csr_ws_bytes(int S) { return 2 * ws_round((size_t)S + 1, 4); }

int csr_from_keys_launch(const int* keys, int n, int S, int* rowptr, int* perm, void* ws, size_t ws_bytes,
                         hipStream_t s) {
  CGAT_CHECK_ARG(S >= 0 && n >= 0, "csr: negative size");
  Workspace w(ws, ws_bytes);
  int* count = w.take<int>((size_t)S + 1);
  int* cursor = w.take<int>((size_t)S + 1);
  if (!w.ok) {
    cgat_set_error("csr: workspace too small (%zu < %zu)", ws_bytes, w.off);
    return CGAT_ERR_WORKSPACE;
  }
  CGAT_HIP(hipMemsetAsync(count, 0, ((size_t)S + 1) * 4, s));
  CGAT_HIP(hipMemsetAsync(cursor, 0, ((size_t)S + 1) * 4, s));
  if (n > 0) {
    hipLaunchKernelGGL(hist_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, keys, n, S, count);
    CGAT_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(exscan_kernel, dim3(1), dim3(1024), 0, s, count, S, rowptr);
  CGAT_LAUNCH_CHECK();
  if (n > 0) {
    hipLaunchKernelGGL(fill_kernel_csr, dim3(cdiv(n, 256)), dim3(256), 0, s, keys, n, S, rowptr, cursor, perm);
    CGAT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sort_segments_kernel, dim3(cdiv(S, 256)), dim3(256), 0, s, rowptr, S, perm);
    CGAT_LAUNCH_CHECK();
  }
  return CGAT_OK;
}

__global__ void split_edge_index_kernel(const int64_t* __restrict__ ei, int E, int* __restrict__ src,
                                        int* __restrict__ dst) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E) return;
  src[i] = (int)ei[i];
  dst[i] = (int)ei[(long)E + i];
}

__global__ void gather_int_kernel(const int* __restrict__ v, const int* __restrict__ idx, int n, int* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = v[idx[i]];
}

size_t plan_ws_bytes(int E, int N) { return 2 * ws_round((size_t)E, 4) + csr_ws_bytes(N); }

int plan_build_launch(const int64_t* edge_index, int E, int N, int* dst_rowptr, int* dst_perm, int* dst_sorted,
                      int* src_sorted, int* src_rowptr, int* src_pos, void* ws, size_t ws_bytes, hipStream_t s) {
  Workspace w(ws, ws_bytes);
  int* src = w.take<int>((size_t)E);
  int* dst = w.take<int>((size_t)E);
  if (!w.ok) {
    cgat_set_error("plan: workspace too small (%zu < %zu)", ws_bytes, plan_ws_bytes(E, N));
    return CGAT_ERR_WORKSPACE;
  }
  void* rest = (char*)ws + w.off;
  size_t rest_bytes = ws_bytes - w.off;
  if (E > 0) {
    hipLaunchKernelGGL(split_edge_index_kernel, dim3(cdiv(E, 256)), dim3(256), 0, s, edge_index, E, src, dst);
    CGAT_LAUNCH_CHECK();
  }
  CGAT_TRY(csr_from_keys_launch(dst, E, N, dst_rowptr, dst_perm, rest, rest_bytes, s));
  if (E > 0) {
    hipLaunchKernelGGL(gather_int_kernel, dim3(cdiv(E, 256)), dim3(256), 0, s, dst, dst_perm, E, dst_sorted);
    CGAT_LAUNCH_CHECK();
    hipLaunchKernelGGL(gather_int_kernel, dim3(cdiv(E, 256)), dim3(256), 0, s, src, dst_perm, E, src_sorted);
    CGAT_LAUNCH_CHECK();
  }
  CGAT_TRY(csr_from_keys_launch(src_sorted, E, N, src_rowptr, src_pos, rest, rest_bytes, s));
  return CGAT_OK;
}
