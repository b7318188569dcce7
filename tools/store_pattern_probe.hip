// Dev tool: HBM write rate of Z [E, 1536] fp32 (6.1 GB at E = 1 000 080) under the store patterns of the per-edge forward:
//   A  edge_zc's: a 512-thread workgroup owns 256 rows and walks 48 chunks of 32 columns; per chunk a wave writes its
//      32 rows x 128 B as four 1-KB instructions (16 rows x 64 B each);
//   B  the same tile walk with 64-column (256-B) pieces;  C  128-column (512-B) pieces;
//   D  row-contiguous: a wave writes whole 6-KB rows.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/spp tools/store_pattern_probe.hip && /tmp/spp
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int COLS>   // columns per piece: 32, 64, 128
__global__ __launch_bounds__(512) void tile_walk(float* Z, int E) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n16 = lane & 15, kg = lane >> 4;
  const int row_w = blockIdx.x * 256 + wave * 32;
  const int ra = min(row_w + n16, E - 1), rb = min(row_w + 16 + n16, E - 1);
  for (int c0 = 0; c0 < 1536; c0 += COLS) {
#pragma unroll
    for (int cc = 0; cc < COLS; cc += 16) {
      const float4 v = make_float4(c0, cc, lane, wave);
      *reinterpret_cast<float4*>(Z + (long)ra * 1536 + c0 + cc + 4 * kg) = v;
      *reinterpret_cast<float4*>(Z + (long)rb * 1536 + c0 + cc + 4 * kg) = v;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(512) void row_contig(float* Z, int E) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = blockIdx.x * 256 + wave * 32; r < min(E, blockIdx.x * 256 + wave * 32 + 32); ++r)
    for (int c = 0; c < 1536; c += 256) *reinterpret_cast<float4*>(Z + (long)r * 1536 + c + 4 * lane) = make_float4(r, c, lane, 0.f);
}
int main() {
  const int E = 1000080;
  float* Z;
  hipMalloc(&Z, (size_t)E * 1536 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = (E + 255) / 256;
  for (int k = 0; k < 4; ++k) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (k == 0) hipLaunchKernelGGL(tile_walk<32>, dim3(grid), dim3(512), 0, 0, Z, E);
      if (k == 1) hipLaunchKernelGGL(tile_walk<64>, dim3(grid), dim3(512), 0, 0, Z, E);
      if (k == 2) hipLaunchKernelGGL(tile_walk<128>, dim3(grid), dim3(512), 0, 0, Z, E);
      if (k == 3) hipLaunchKernelGGL(row_contig, dim3(grid), dim3(512), 0, 0, Z, E);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const char* names[4] = {"A tile walk, 128-B pieces", "B tile walk, 256-B pieces", "C tile walk, 512-B pieces", "D row-contiguous"};
    printf("%-28s %.3f ms  %.2f TB/s\n", names[k], best, (double)E * 1536 * 4 / best / 1e9);
  }
  return 0;
}
