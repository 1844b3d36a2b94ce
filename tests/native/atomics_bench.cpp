// Micro-benchmark: the fp32-atomic epilogue of the weight-gradient kernels.  G work-groups each add a 64 KB block of partial sums
// into one of T result blocks, either with the default (agent-scope) atomics - coherent over the eight XCDs, so performed at the
// memory side - or with workgroup-scope atomics, which the hardware performs in the issuing XCD's own L2 (only correct when all
// adders of a block sit on one XCD: a two-level reduction would add per XCD first, then once across XCDs).
//   hipcc --offload-arch=gfx950 -O3 tests/native/atomics_bench.cpp -o tests/native/atomics_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

template <int SCOPE, bool PER_XCD>
__global__ __launch_bounds__(256) void add_kernel(float* __restrict__ dst, int T, int block_floats) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7;
  const int tile = (blockIdx.x / (PER_XCD ? 8 : 1)) % T;
  float* base = dst + ((size_t)(PER_XCD ? xcc * T : 0) + tile) * block_floats;
  for (int i = threadIdx.x; i < block_floats; i += 256) {
    const float v = (float)(i & 7) + 1.f;
    if (SCOPE == 0) __hip_atomic_fetch_add(base + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(base + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

template <int SCOPE, bool PER_XCD>
static void run(const char* name, float* dst, int G, int T, int block_floats) {
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  hipLaunchKernelGGL((add_kernel<SCOPE, PER_XCD>), dim3(G), dim3(256), 0, 0, dst, T, block_floats);
  HIPCHK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((add_kernel<SCOPE, PER_XCD>), dim3(G), dim3(256), 0, 0, dst, T, block_floats);
  HIPCHK(hipEventRecord(e1)); HIPCHK(hipEventSynchronize(e1));
  float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  const double bytes = (double)G * block_floats * 4;
  printf("%-44s G %4d T %3d block %3d KB: %7.1f us  %6.2f TB/s of partial sums  (%5.1f G atomics/s)\n", name, G, T, block_floats / 256,
         ms * 1e3, bytes / ms * 1e-9, G * (double)block_floats / ms * 1e-6);
}

int main() {
  float* dst; HIPCHK(hipMalloc(&dst, (size_t)64 << 20)); HIPCHK(hipMemset(dst, 0, (size_t)64 << 20));
  for (int T : {1, 4, 16}) {
    for (int G : {256, 1024}) {
      run<0, false>("agent scope, all XCDs into one block set", dst, G, T, 16384);
      run<1, true>("workgroup scope, one block set per XCD", dst, G, T, 16384);
      run<0, true>("agent scope, one block set per XCD", dst, G, T, 16384);
    }
  }
  return 0;
}
