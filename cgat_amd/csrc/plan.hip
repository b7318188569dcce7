// CSR plans: the destination-sorted edge order every layer of a batch reuses, forward and
// backward.  PyG's propagate keys both the softmax and the scatter-add on edge_index[1]
// (reference CGAT.py:313-326, node_dim=0, flow source_to_target); sorting once by that key
// turns the scatter into contiguous, atomics-free segment reductions.  A second CSR over the
// sorted positions keyed by edge_index[0] serves the transposed accumulation (grad wrt x_j).
//
// Keys outside [0, S) are dropped: rowptr[S] then ends below n, which is how the caller detects an invalid
// edge_index / batch index (cgat_amd/ops.py raises IndexError, as the reference's index_select would).
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

// ascending sort of each segment's ids.  Short segments (the in-degree of an atom) by one thread's insertion sort;
// longer ones are queued (long_list, counter long_n) for sort_long_segments_kernel, so that a hub atom or a large
// crystal does not leave one lane with an O(d^2) loop.
#define SEG_SHORT 32
__global__ void sort_segments_kernel(const int* __restrict__ rowptr, int S, int* __restrict__ perm,
                                     int* __restrict__ long_list, int* __restrict__ long_n) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  int r0 = rowptr[s], r1 = rowptr[s + 1];
  if (r1 - r0 > SEG_SHORT) {
    long_list[atomicAdd(long_n, 1)] = s;   // order of the queue is irrelevant: every segment is sorted independently
    return;
  }
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

// One 1024-thread workgroup per queued segment: rank sort (ids are distinct, so rank = number of smaller ids is a
// permutation): every thread keeps up to 8 ids in registers, the segment passes through LDS in 1024-id tiles, then
// the ids are written to their ranks -- d^2 / 1024 comparisons per thread, no temporary array.
#define SEG_LONG_PER_THREAD 8
__global__ __launch_bounds__(1024) void sort_long_segments_kernel(const int* __restrict__ rowptr, int* __restrict__ perm,
                                                                  const int* __restrict__ long_list,
                                                                  const int* __restrict__ long_n) {
  __shared__ int tile[1024];
  const int tid = threadIdx.x;
  const int nl = *long_n;
  for (int li = blockIdx.x; li < nl; li += gridDim.x) {
    const int s = long_list[li];
    const int r0 = rowptr[s], d = rowptr[s + 1] - r0;
    if (d > 1024 * SEG_LONG_PER_THREAD) {   // beyond the register budget: one lane, correct but slow
      if (tid == 0) {
        for (int i = r0 + 1; i < r0 + d; ++i) {
          int v = perm[i];
          int j = i - 1;
          while (j >= r0 && perm[j] > v) { perm[j + 1] = perm[j]; --j; }
          perm[j + 1] = v;
        }
      }
      __syncthreads();
      continue;
    }
    int v[SEG_LONG_PER_THREAD], rank[SEG_LONG_PER_THREAD];
#pragma unroll
    for (int k = 0; k < SEG_LONG_PER_THREAD; ++k) {
      const int i = tid + 1024 * k;
      v[k] = i < d ? perm[r0 + i] : 0x7fffffff;
      rank[k] = 0;
    }
    const int tiles = (d + 1023) / 1024;
    for (int t = 0; t < tiles; ++t) {
      __syncthreads();
      int mine = 0x7fffffff;
#pragma unroll
      for (int k = 0; k < SEG_LONG_PER_THREAD; ++k)
        if (k == t) mine = v[k];
      tile[tid] = mine;
      __syncthreads();
      const int lim = min(1024, d - 1024 * t);
      for (int j = 0; j < lim; ++j) {
        const int x = tile[j];
#pragma unroll
        for (int k = 0; k < SEG_LONG_PER_THREAD; ++k) rank[k] += x < v[k] ? 1 : 0;
      }
    }
    __syncthreads();   // every id is in registers before the first one is overwritten
#pragma unroll
    for (int k = 0; k < SEG_LONG_PER_THREAD; ++k)
      if (tid + 1024 * k < d) perm[r0 + rank[k]] = v[k];
    __syncthreads();
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
  CGAT_TRY(fill_launch(reinterpret_cast<float*>(count), 0.f, (long)S + 1, s));   // (0.f and int 0 share their bits)
  CGAT_TRY(fill_launch(reinterpret_cast<float*>(cursor), 0.f, (long)S + 1, s));
  // keys outside [0, S) leave the tail of perm unfilled; the gathers that consume it before the caller has looked at
  // rowptr[S] must still read valid indices
  if (n > 0) CGAT_TRY(fill_launch(reinterpret_cast<float*>(perm), 0.f, (long)n, s));
  if (n > 0) {
    hipLaunchKernelGGL(hist_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, keys, n, S, count);
    CGAT_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(exscan_kernel, dim3(1), dim3(1024), 0, s, count, S, rowptr);
  CGAT_LAUNCH_CHECK();
  if (n > 0 && S > 0) {
    hipLaunchKernelGGL(fill_kernel_csr, dim3(cdiv(n, 256)), dim3(256), 0, s, keys, n, S, rowptr, cursor, perm);
    CGAT_LAUNCH_CHECK();
    // `count` is free once the scan has consumed it: it becomes the queue of long segments, cursor[S] its length
    hipLaunchKernelGGL(sort_segments_kernel, dim3(cdiv(S, 256)), dim3(256), 0, s, rowptr, S, perm, count, cursor + S);
    CGAT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sort_long_segments_kernel, dim3(S < 256 ? (S > 0 ? S : 1) : 256), dim3(1024), 0, s, rowptr, perm,
                       (const int*)count, (const int*)(cursor + S));
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
