// Micro-benchmark: does the order in which a second streaming pass walks a tensor matter to the 256 MiB Infinity Cache?
// Test infrastructure (informs the traversal order of the normalisation passes in norm.hip; numbers quoted in DESIGN.md).
//   hipcc --offload-arch=gfx950 -O3 tests/native/mall_bench.cpp -o tests/native/mall_bench && tests/native/mall_bench
// For a tensor of S bytes: pass 1 walks it front to back (read, write, or read A -> write B), pass 2 (timed) reads what pass 1
// touched either front to back again or back to front.  Between pairs a 1 GiB buffer is read to flush the cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

typedef __attribute__((ext_vector_type(4))) float f4;

// OP 0: read, 1: write, 2: read src -> write dst.  Block b owns the b-th (or, reversed, the b-th from the end) 64 KiB chunk.
template <int OP>
__global__ __launch_bounds__(256) void stream_kernel(const f4* __restrict__ src, f4* __restrict__ dst, int nchunks, int reverse,
                                                     float* sink) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const size_t chunk = reverse ? (size_t)(nchunks - 1 - c) : (size_t)c;
    const size_t base = chunk * 4096;  // 64 KiB = 4096 x 16 B
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const size_t idx = base + (size_t)i * 256 + threadIdx.x;
      if (OP == 0) { f4 v = src[idx]; acc += v; }
      if (OP == 1) { dst[idx] = acc; }
      if (OP == 2) { f4 v = src[idx]; dst[idx] = v * 1.5f; }
    }
  }
  if (OP == 0 && acc[0] + acc[1] + acc[2] + acc[3] == 123456.f) sink[0] = acc[0];
}

template <int OP>
static float launch(const char* src, char* dst, size_t bytes, int reverse, float* sink, bool timed) {
  const int nchunks = (int)(bytes >> 16);
  hipEvent_t e0, e1;
  if (timed) { HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventRecord(e0)); }
  hipLaunchKernelGGL((stream_kernel<OP>), dim3(nchunks), dim3(256), 0, 0, (const f4*)src, (f4*)dst, nchunks, reverse, sink);
  float ms = 0.f;
  if (timed) { HIPCHK(hipEventRecord(e1)); HIPCHK(hipEventSynchronize(e1)); HIPCHK(hipEventElapsedTime(&ms, e0, e1)); }
  return ms;
}

int main() {
  const size_t cap = (size_t)640 << 20;
  char *a, *b, *flush; float* sink;
  HIPCHK(hipMalloc(&a, cap)); HIPCHK(hipMalloc(&b, cap)); HIPCHK(hipMalloc(&flush, (size_t)1 << 30)); HIPCHK(hipMalloc(&sink, 64));
  HIPCHK(hipMemset(a, 0, cap)); HIPCHK(hipMemset(b, 0, cap)); HIPCHK(hipMemset(flush, 0, (size_t)1 << 30));
  printf("# pass 1 -> pass 2 (timed): TB/s of pass 2 by tensor size and direction of pass 2 (pass 1 is always front to back)\n");
  for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 384, 512, 640}) {
    const size_t S = mb << 20;
    for (int first = 0; first < 3; ++first) {      // what pass 1 does
      for (int second = 0; second < 2; ++second) { // what pass 2 does: 0 = read, 2 = read -> write elsewhere
        if (first == 2 && second == 0) continue;
        double rate[2] = {0, 0};
        for (int rev = 0; rev < 2; ++rev) {
          float best = 1e9f;
          for (int rep = 0; rep < 3; ++rep) {
            launch<0>(flush, nullptr, (size_t)1 << 30, 0, sink, false);
            // pass 1
            if (first == 0) launch<0>(a, nullptr, S, 0, sink, false);
            if (first == 1) launch<1>(nullptr, a, S, 0, sink, false);
            if (first == 2) launch<2>(b, a, S, 0, sink, false);       // reads b, writes a
            // pass 2 over a
            const float ms = second == 0 ? launch<0>(a, nullptr, S, rev, sink, true) : launch<2>(a, b, S, rev, sink, true);
            if (ms < best) best = ms;
          }
          const double moved = second == 0 ? (double)S : 2.0 * (double)S;
          rate[rev] = moved / best * 1e-9;
        }
        static const char* fn[3] = {"read a", "write a", "read b -> write a"};
        static const char* sn[2] = {"read a", "read a -> write b"};
        printf("%4zu MiB  pass 1 %-18s pass 2 %-18s same order %6.2f TB/s   reversed %6.2f TB/s\n", mb, fn[first], sn[second],
               rate[0], rate[1]);
      }
    }
  }
  return 0;
}
