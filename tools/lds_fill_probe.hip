// Dev tool: how fast can a CU fill LDS from an L2-resident buffer on gfx950, and what does the fill cost concurrent LDS reads?
//   mode 0  global_load_lds_dwordx4 (LDS-DMA), all 8 waves, DEPTH pieces of 1 KB in flight per wave
//   mode 1  global_load_dwordx4 into registers + ds_write_b128, all 8 waves, DEPTH pieces in flight per wave
//   mode 2  ds_read_b128 only, all 8 waves (the LDS read rate)
//   mode 3  waves 0-3 LDS-DMA, waves 4-7 ds_read_b128 (6 x the iterations)     mode 4  the same with registers + ds_write
//   modes 5 / 7 / 6  the two fills and the reads of modes 3 / 4 alone: contention = mixed time against the longer of the two
// One 512-thread workgroup per CU, every workgroup streams the same source (128 KB: L2-resident; then 3 - 50 MB) in order
// into a 64-KB LDS window.
// Prints GB/s per CU of the fill and of the reads.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lds_fill_probe tools/lds_fill_probe.hip ; run: tools/lds_fill_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds_b128(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void probe(const uint4* __restrict__ src, float* __restrict__ sink, int iters, int src_kb) {
  __shared__ uint4 lds[4096];                        // 64 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  const bool filler = MODE == 0 || MODE == 1 || ((MODE == 3 || MODE == 4 || MODE == 5 || MODE == 7) && wave_u < 4);
  const bool reader = MODE == 2 || ((MODE == 3 || MODE == 4 || MODE == 6) && wave_u >= 4);
  constexpr int RMUL = 6;   // the readers' share of iterations in the mixed modes (a read is that much faster than a fill)
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < 4096; i += 512) lds[i] = make_uint4(i, i, i, i);
  __syncthreads();
  if (filler) {
    if (MODE == 0 || MODE == 3 || MODE == 5) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          const int piece = (it * DEPTH + d) * 8 + wave_u;           // 1-KB pieces, round-robin over the waves
          const uint4* s = src + (size_t)(piece % src_kb) * 64;       // src_kb KB of source, streamed in order by every CU
          const unsigned dst = __builtin_amdgcn_readfirstlane(sbase + (unsigned)(piece & 63) * 1024);
          glds_b128(s, (unsigned)lane * 16, dst);
        }
        wait_vmcnt<DEPTH>();                                          // the previous iteration's pieces have landed
      }
      wait_vmcnt<0>();
    } else {
      for (int it = 0; it < iters; ++it) {
        uint4 r[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          const int piece = (it * DEPTH + d) * 8 + wave_u;
          r[d] = src[(size_t)(piece % src_kb) * 64 + lane];
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          const int piece = (it * DEPTH + d) * 8 + wave_u;
          lds[(piece & 63) * 64 + lane] = r[d];
        }
      }
    }
  }
  if (reader) {
    const f32x4* lp = reinterpret_cast<const f32x4*>(lds) + lane;
    for (int it = 0; it < iters * (MODE == 2 ? 1 : RMUL); ++it) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int piece = (it * DEPTH + d) * 8 + wave_u;
        const f32x4 v = lp[(piece & 63) * 64];
        acc += v;
      }
    }
  }
  __syncthreads();
  sink[blockIdx.x * 512 + tid] = acc[0] + acc[1] + acc[2] + acc[3] + (float)lds[tid].x;
}

template <int MODE, int DEPTH>
static void run(const char* name, const uint4* src, float* sink, int src_kb = 128) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE, DEPTH>), dim3(256), dim3(512), 0, 0, src, sink, 100, src_kb);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<MODE, DEPTH>), dim3(256), dim3(512), 0, 0, src, sink, iters, src_kb);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const int fill_waves = (MODE == 0 || MODE == 1) ? 8 : ((MODE == 2 || MODE == 6) ? 0 : 4);
  const double read_waves = MODE == 2 ? 8 : ((MODE == 3 || MODE == 4 || MODE == 6) ? 4 * 6.0 : 0);
  const double per_wave = (double)iters * DEPTH * 1024.0;
  printf("%-44s src %6d KB depth %2d: %7.3f ms   fill %7.1f GB/s per CU   reads %7.1f GB/s per CU\n", name, src_kb, DEPTH, ms,
         fill_waves * per_wave / (ms * 1e-3) / 1e9, read_waves * per_wave / (ms * 1e-3) / 1e9);
}
int main() {
  uint4* src; float* sink;
  hipMalloc(&src, 64 << 20); hipMalloc(&sink, 256 * 512 * 4);
  hipMemset(src, 1, 64 << 20);
  run<0, 1>("LDS-DMA, 8 waves", src, sink);  run<0, 4>("LDS-DMA, 8 waves", src, sink);  run<0, 8>("LDS-DMA, 8 waves", src, sink);
  run<1, 1>("registers + ds_write_b128, 8 waves", src, sink);  run<1, 4>("registers + ds_write_b128, 8 waves", src, sink);
  run<1, 8>("registers + ds_write_b128, 8 waves", src, sink);
  run<2, 8>("ds_read_b128 only, 8 waves", src, sink);
  run<3, 4>("LDS-DMA (waves 0-3) + ds_read_b128 (4-7)", src, sink);  run<3, 8>("LDS-DMA (waves 0-3) + ds_read_b128 (4-7)", src, sink);
  run<4, 4>("registers + ds_write (0-3) + ds_read (4-7)", src, sink);  run<4, 8>("registers + ds_write (0-3) + ds_read (4-7)", src, sink);
  run<5, 4>("LDS-DMA, waves 0-3 alone", src, sink);  run<7, 4>("registers + ds_write, waves 0-3 alone", src, sink);
  run<6, 4>("ds_read_b128, waves 4-7 alone (6 x iters)", src, sink);
  // a source larger than one XCD's 4-MB L2, streamed in order by all CUs (the prepared T of one predicted layer: 12.8 MB)
  run<0, 4>("LDS-DMA, 8 waves", src, sink, 3 * 1024);  run<0, 4>("LDS-DMA, 8 waves", src, sink, 4300);
  run<0, 4>("LDS-DMA, 8 waves", src, sink, 12800);  run<0, 4>("LDS-DMA, 8 waves", src, sink, 51200);
  run<1, 4>("registers + ds_write_b128, 8 waves", src, sink, 12800);
  return 0;
}
