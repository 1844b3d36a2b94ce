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


// ---- store side (round 3): what an output tile costs.  One work-group of 8 waves per CU writes 128 KiB tiles (256 pixels x
// 512 B = 256 bf16 channels) the way the conv epilogue does - 16 x 16-byte stores per lane and tile - in four wave-instruction
// shapes, optionally followed by NM MFMAs per wave and tile (the next tile's K loop) and optionally by s_waitcnt vmcnt(0)
// (what an in-order counted wait behind the stores amounts to).
//   SMODE 0: 16 pixels x 64 B per instruction (the 16x16x32 MFMA D layout, two 8-channel runs per lane)
//   SMODE 1: 8 pixels x 128 B (full cache lines)       SMODE 2: 32 pixels x 32 B (the 32x32x16 D layout)
//   SMODE 3: 1 KiB contiguous
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
template <int SMODE, bool NT, bool DRAIN>
__global__ __launch_bounds__(512) void push_kernel(char* __restrict__ dst, int tiles, int nm, int do_store, float* sink) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  f32x4_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  s16x8_t fa, fb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { fa[i] = (short)(lane + i); fb[i] = (short)(lane * 3 + i); }
  u32x4_t v = {(unsigned)lane, 1u, 2u, 3u};
  for (int t = 0; t < tiles; ++t) {
    char* tile = dst + ((size_t)t * gridDim.x + blockIdx.x) * (128 << 10);
    if (do_store) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        size_t off;
        if (SMODE == 0) off = (size_t)((w & 1) * 128 + (k >> 1) * 16 + (lane & 15)) * 512 + (w >> 1) * 128 + (k & 1) * 64 + (lane >> 4) * 16;
        else if (SMODE == 1) off = (size_t)((w & 1) * 128 + k * 8 + (lane >> 3)) * 512 + (w >> 1) * 128 + (lane & 7) * 16;
        else if (SMODE == 2) off = (size_t)((w & 1) * 128 + (k >> 2) * 32 + (lane & 31)) * 512 + (w >> 1) * 128 + (k & 3) * 32 + (lane >> 5) * 16;
        else off = (size_t)(w * 16 + k) * 1024 + lane * 16;
        char* p = tile + off;
        if (NT) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
      }
      if (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (int m = 0; m < nm; ++m) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i], 0, 0, 0);
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += acc[i][0];
  if (r == 123456.f) sink[0] = r;
}

template <int SMODE, bool NT, bool DRAIN>
static void run_push(const char* name, char* dst, int nm, int do_store, float* sink) {
  const int tiles = 32;  // 32 x 128 KiB = 4 MiB per CU, 1 GiB over the chip
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  hipLaunchKernelGGL((push_kernel<SMODE, NT, DRAIN>), dim3(256), dim3(512), 0, 0, dst, tiles, nm, do_store, sink);
  HIPCHK(hipEventRecord(e0));
  hipLaunchKernelGGL((push_kernel<SMODE, NT, DRAIN>), dim3(256), dim3(512), 0, 0, dst, tiles, nm, do_store, sink);
  HIPCHK(hipEventRecord(e1)); HIPCHK(hipEventSynchronize(e1));
  float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = do_store ? 256.0 * tiles * (128 << 10) : 0.0;
  printf("%-44s nt %d drain %d mfma/tile/wave %5d stores %d: %8.3f ms  %6.2f us per tile  %6.2f TB/s (%5.1f GB/s per CU)\n", name, (int)NT,
         (int)DRAIN, nm * 8, do_store, ms, ms * 1e3 / tiles, bytes / ms * 1e-9, bytes / ms * 1e-6 / 256);
}

static void store_bench(char* dst, float* sink) {
  for (int nm : {0, 64, 288}) {  // 0: stores only; 512 / 2304 MFMA per wave and tile = the K loop of a 1x1 (K = 512) / 3x3 (K = 2304) layer
    if (nm) run_push<0, false, false>("(MFMA only)", dst, nm, 0, sink);
    run_push<0, false, false>("store 16 px x 64 B", dst, nm, 1, sink);
    run_push<0, true, false>("store 16 px x 64 B", dst, nm, 1, sink);
    run_push<0, true, true>("store 16 px x 64 B", dst, nm, 1, sink);
    run_push<1, false, false>("store 8 px x 128 B", dst, nm, 1, sink);
    run_push<1, true, false>("store 8 px x 128 B", dst, nm, 1, sink);
    run_push<1, true, true>("store 8 px x 128 B", dst, nm, 1, sink);
    run_push<2, false, false>("store 32 px x 32 B", dst, nm, 1, sink);
    run_push<2, true, false>("store 32 px x 32 B", dst, nm, 1, sink);
    run_push<3, false, false>("store 1 KiB contiguous", dst, nm, 1, sink);
    run_push<3, true, false>("store 1 KiB contiguous", dst, nm, 1, sink);
    run_push<3, true, true>("store 1 KiB contiguous", dst, nm, 1, sink);
  }
}

int main(int argc, char** argv) {
  const size_t total = (size_t)1 << 30;
  char* src; float* sink;
  HIPCHK(hipMalloc(&src, total + (1 << 20))); HIPCHK(hipMemset(src, 1, total)); HIPCHK(hipMalloc(&sink, 64));
  if (argc > 1 && argv[1][0] == 's') { store_bench(src, sink); return 0; }
  if (argc > 1 && argv[1][0] == 'h') {
    // 2 MiB per work-group: 512 MiB (one group per CU) / 1 GiB (two) of footprint cycled 16 / 8 times - beyond the 256 MiB MALL,
    // i.e. the HBM read rate itself (the 1 MiB windows below stay MALL-resident)
    const size_t window = (size_t)2 << 20;
    run<3, 8>("glds 1 KiB contiguous", src, window, 512, sink);
    run<2, 8>("glds 4 rows x 256 B (pitch 512)", src, window, 512, sink);
    run<1, 8>("glds 8 rows x 128 B (pitch 512)", src, window, 512, sink);
    run<0, 8>("glds 16 rows x 64 B (pitch 512)", src, window, 512, sink);
    run<5, 8>("global_load_dwordx4 -> VGPR, 1 KiB contiguous", src, window, 512, sink);
    return 0;
  }
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
