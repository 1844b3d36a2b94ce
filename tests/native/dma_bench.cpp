// Micro-benchmark: how many bytes per second one CU can pull from L2 into LDS / registers, by access pattern.
// Test infrastructure (informs the staging design of conv_tile.hip / conv_wgrad_kernel; numbers quoted in DESIGN.md).
//   hipcc --offload-arch=gfx950 -O3 tests/native/dma_bench.cpp -o tests/native/dma_bench && tests/native/dma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// MODE 0: global_load_lds, 16 rows x 64 B per wave instruction (row pitch ROWB bytes)   [conv_tile half-K tiles]
// MODE 1: global_load_lds, 8 rows x 128 B
// MODE 2: global_load_lds, 4 rows x 256 B                                                [wgrad: pixel rows of 128 channels]
// MODE 3: global_load_lds, 1 KiB contiguous
// MODE 4: global_load_dwordx4 into registers (16 rows x 64 B), no LDS
// MODE 5: global_load_dwordx4 into registers, 1 KiB contiguous
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void pull_kernel(const char* __restrict__ src, size_t window_bytes, int iters, int rowb,
                                                          float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // every CU (work-group) walks its own window, so that the data stays in its XCD's L2
  const char* base = src + (size_t)blockIdx.x * window_bytes;
  const int rows_per_instr = MODE == 0 || MODE == 4 ? 16 : MODE == 1 ? 8 : MODE == 2 ? 4 : 1;
  const int seg = 1024 / rows_per_instr;                 // contiguous bytes per row piece
  const int lpr = seg / 16;                              // lanes per row
  const size_t lane_off = (MODE == 3 || MODE == 5) ? (size_t)lane * 16 : (size_t)(lane / lpr) * rowb + (size_t)(lane % lpr) * 16;
  const size_t instr_stride = (MODE == 3 || MODE == 5) ? 1024 : (size_t)rows_per_instr * rowb;
  const size_t instrs_in_window = window_bytes / instr_stride;
  float acc = 0.f;
  size_t k = (size_t)w;                                  // instruction index, interleaved over the waves
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const char* p = base + (k % instrs_in_window) * instr_stride + lane_off;
      k += WAVES;
      if constexpr (MODE <= 3) {
        __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(smem + ((w * 8 + u) & 63) * 1024), 16, 0, 0);
      } else {
        const float4 v = *reinterpret_cast<const float4*>(p);
        acc += v.x;
      }
    }
    if constexpr (MODE <= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  if constexpr (MODE <= 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc = smem[lane]; }
  if (acc == 123456.f) sink[0] = acc;
}

template <int MODE, int WAVES>
static void run(const char* name, const char* src, size_t window, int rowb, float* sink) {
  const int iters = 4096 / WAVES;  // 8 instructions x 1 KiB per iteration per wave -> 32 MiB per CU
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  for (int groups_per_cu = 1; groups_per_cu <= 2; ++groups_per_cu) {
    const int grid = 256 * groups_per_cu;
    hipLaunchKernelGGL((pull_kernel<MODE, WAVES>), dim3(grid), dim3(WAVES * 64), 64 * 1024, 0, src, window, iters, rowb, sink);
    HIPCHK(hipEventRecord(e0));
    hipLaunchKernelGGL((pull_kernel<MODE, WAVES>), dim3(grid), dim3(WAVES * 64), 64 * 1024, 0, src, window, iters, rowb, sink);
    HIPCHK(hipEventRecord(e1)); HIPCHK(hipEventSynchronize(e1));
    float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)grid * WAVES * iters * 8 * 1024;
    printf("%-46s waves/group %d groups/CU %d window %5zu KiB: %7.2f TB/s  (%6.1f GB/s per CU)\n", name, WAVES, groups_per_cu,
           window >> 10, bytes / ms * 1e-9, bytes / ms * 1e-6 / 256);
  }
}

int main() {
  const size_t total = (size_t)1 << 30;
  char* src; float* sink;
  HIPCHK(hipMalloc(&src, total + (1 << 20))); HIPCHK(hipMemset(src, 1, total)); HIPCHK(hipMalloc(&sink, 64));
  for (size_t window : {(size_t)64 << 10, (size_t)1 << 20}) {   // 64 KiB per CU: L2-resident; 1 MiB per CU x 512: MALL / HBM
    run<0, 8>("glds 16 rows x 64 B (pitch 512)", src, window, 512, sink);
    run<1, 8>("glds 8 rows x 128 B (pitch 512)", src, window, 512, sink);
    run<2, 8>("glds 4 rows x 256 B (pitch 512)", src, window, 512, sink);
    run<3, 8>("glds 1 KiB contiguous", src, window, 512, sink);
    run<4, 8>("global_load_dwordx4 -> VGPR, 16 rows x 64 B", src, window, 512, sink);
    run<5, 8>("global_load_dwordx4 -> VGPR, 1 KiB contiguous", src, window, 512, sink);
    run<0, 4>("glds 16 rows x 64 B (pitch 512)", src, window, 512, sink);
    run<3, 4>("glds 1 KiB contiguous", src, window, 512, sink);
  }
  return 0;
}
