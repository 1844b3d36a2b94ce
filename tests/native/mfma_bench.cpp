// Micro-benchmark: the MFMA + LDS-fragment-read loop of the conv kernels ALONE (no staging, no epilogue), by MFMA shape,
// waves per work-group, wave tile and LDS row length.  Test infrastructure: it answers what the K loop of conv_tile.hip /
// conv_halo.hip / conv_mfma32.hip can reach before operand delivery is added (numbers in profiles/r04_mfma_loop.txt).
//   hipcc --offload-arch=gfx950 -O3 tests/native/mfma_bench.cpp -o tests/native/mfma_bench && tests/native/mfma_bench
//
// A work-group owns a 256 (channels) x 256 (pixels) tile whose operands sit in an LDS ring of slots (rows of ROWB bytes =
// ROWB / 2 reduction channels, XOR-swizzled so that every fragment read is conflict-free) and multiplies slot after slot:
//   SHAPE 16: v_mfma_f32_16x16x32_bf16, fragment = 16 rows x 4 chunks of 16 B       (what rounds 1-3 use)
//   SHAPE 32: v_mfma_f32_32x32x16_bf16, fragment = 32 rows x 2 chunks of 16 B
//   WAVES 8:  wave tile 64 ch x 128 px (2 waves per SIMD, 128 accumulator registers)
//   WAVES 4:  wave tile 128 ch x 128 px (1 wave per SIMD, 256 accumulator registers)
// FLAGS bit 0: one s_barrier per 32-channel slab (the staging hand-over of the real kernels), bit 1: no LDS reads at all
// (fragments loaded once: the matrix pipe alone), bit 2: s_setprio(1) around MFMA runs.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE, int ROWB> __device__ __forceinline__ int frag_addr(int row0, int kstep, int lane) {
  constexpr int FR = SHAPE;                     // rows per fragment
  constexpr int CPK = SHAPE == 16 ? 4 : 2;      // 16-byte chunks per k step
  const int row = row0 + (lane & (FR - 1));
  const int chunk = kstep * CPK + (lane / FR);
  int sw;
  if constexpr (ROWB == 128) sw = (row >> 1) & 7;
  else if constexpr (SHAPE == 16) sw = (-(row >> 2)) & 3;
  else sw = (row >> 2) & 3;
  return row * ROWB + ((chunk ^ sw) << 4);
}

template <int SHAPE, int WAVES, int ROWB, int FLAGS>
__global__ __launch_bounds__(WAVES * 64) void loop_kernel(const uint4* __restrict__ init, int slabs, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SLOT = 512 * ROWB;                 // 256 weight rows + 256 pixel rows
  constexpr int NSLOT = 131072 / SLOT;             // 128 KB ring
  constexpr int WCH = WAVES == 8 ? 64 : 128;       // wave tile channels
  constexpr int WPX = 128;
  constexpr int FA = WCH / SHAPE, FB = WPX / SHAPE;  // fragments per operand
  constexpr int KS = ROWB / (SHAPE == 16 ? 64 : 32); // k steps per slot
  constexpr int SLABS_PER_SLOT = ROWB / 64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < 131072 / 16; i += WAVES * 64) reinterpret_cast<uint4*>(smem)[i] = init[i];
  __syncthreads();
  const int wr = WAVES == 8 ? (w >> 1) : (w >> 1), wc = w & 1;  // channel slice / pixel slice
  const int arow0 = wr * WCH, brow0 = 256 + wc * WPX;
  int aoff[FA], boff[FB];
#pragma unroll
  for (int i = 0; i < FA; ++i) aoff[i] = frag_addr<SHAPE, ROWB>(arow0 + i * SHAPE, 0, lane);
#pragma unroll
  for (int j = 0; j < FB; ++j) boff[j] = frag_addr<SHAPE, ROWB>(brow0 + j * SHAPE, 0, lane);
  // chunk index of k step s is XORed into bits 4.. of the address: (chunk0 + s * CPK) ^ sw == (chunk0 ^ sw) ^ (s * CPK) because
  // chunk0 < CPK and CPK is a power of two
  constexpr int CPK = SHAPE == 16 ? 4 : 2;

  typedef typename std::conditional<SHAPE == 16, f32x4, f32x16>::type acc_t;
  acc_t acc[FA][FB];
#pragma unroll
  for (int i = 0; i < FA; ++i)
#pragma unroll
    for (int j = 0; j < FB; ++j)
      for (int e = 0; e < (SHAPE == 16 ? 4 : 16); ++e) acc[i][j][e] = 0.f;

  bf16x8 a[2][FA], b[2][FB];
  auto load = [&](int buf, int slot, int s) {
#pragma unroll
    for (int i = 0; i < FA; ++i) a[buf][i] = *reinterpret_cast<const bf16x8*>(smem + slot * SLOT + (aoff[i] ^ ((s * CPK) << 4)));
#pragma unroll
    for (int j = 0; j < FB; ++j) b[buf][j] = *reinterpret_cast<const bf16x8*>(smem + slot * SLOT + (boff[j] ^ ((s * CPK) << 4)));
  };
  auto mma = [&](int buf) {
    if constexpr (FLAGS & 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = 0; j < FB; ++j)
#pragma unroll
      for (int i = 0; i < FA; ++i) {
        if constexpr (SHAPE == 16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[buf][i], b[buf][j], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[buf][i], b[buf][j], acc[i][j], 0, 0, 0);
      }
    if constexpr (FLAGS & 4) __builtin_amdgcn_s_setprio(0);
  };

  load(0, 0, 0);
  const int nslots = slabs / SLABS_PER_SLOT;
  int slot = 0;
  for (int it = 0; it < nslots; it += 2) {   // two slots per trip, so that the register double buffer index is a constant
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int nslot = slot + 1 == NSLOT ? 0 : slot + 1;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int cur = (u * KS + s) & 1;
        // fragments of the next k step are requested in front of this k step's MFMAs
        if constexpr (!(FLAGS & 2)) {
          if (s + 1 < KS) load(cur ^ 1, slot, s + 1);
          else load(cur ^ 1, nslot, 0);
        }
        mma((FLAGS & 2) ? 0 : cur);
        if constexpr (FLAGS & 1) {
          // one hand-over per 32-channel slab, as in the real kernels
          constexpr int per_slab = KS / SLABS_PER_SLOT;
          if ((s + 1) % per_slab == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
          }
        }
      }
      slot = nslot;
    }
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < FA; ++i)
#pragma unroll
    for (int j = 0; j < FB; ++j)
      for (int e = 0; e < (SHAPE == 16 ? 4 : 16); ++e) t += acc[i][j][e];
  if (t == 123456.789f) sink[blockIdx.x * WAVES * 64 + tid] = t;
}

static uint32_t rs = 777;
static uint16_t rnd_bf16() {
  rs = rs * 1664525u + 1013904223u;
  const float f = ((rs >> 8) & 0xffff) / 32768.0f - 1.0f;
  uint32_t u; memcpy(&u, &f, 4);
  return (uint16_t)(u >> 16);
}

struct Res { const char* name; std::vector<double> tf; };

template <int SHAPE, int WAVES, int ROWB, int FLAGS>
static double run_once(const uint4* init, float* sink, int slabs) {
  static bool set = false;
  if (!set) { HIPCHK(hipFuncSetAttribute((const void*)loop_kernel<SHAPE, WAVES, ROWB, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); set = true; }
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0));
  hipLaunchKernelGGL((loop_kernel<SHAPE, WAVES, ROWB, FLAGS>), dim3(256), dim3(WAVES * 64), 131072, 0, init, slabs, sink);
  HIPCHK(hipEventRecord(e1)); HIPCHK(hipEventSynchronize(e1));
  float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  HIPCHK(hipEventDestroy(e0)); HIPCHK(hipEventDestroy(e1));
  const double flop = 256.0 * 2.0 * 256 * 256 * 32 * slabs;
  return flop / (ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
  const int slabs = argc > 1 ? atoi(argv[1]) : 4096;   // 32-channel slabs per work-group (4096: ~0.55 ms at 2 PFLOP/s)
  const int zero = argc > 2 ? atoi(argv[2]) : 0;
  std::vector<uint16_t> h(65536);
  for (auto& v : h) v = zero ? 0 : rnd_bf16();
  uint4* init; float* sink;
  HIPCHK(hipMalloc(&init, 131072)); HIPCHK(hipMalloc(&sink, 256 * 512 * 4));
  HIPCHK(hipMemcpy(init, h.data(), 131072, hipMemcpyHostToDevice));
  struct V { const char* name; double (*fn)(const uint4*, float*, int); };
  const V vs[] = {
      {"16x16x32 8w 64B  mfma only      ", run_once<16, 8, 64, 2>},
      {"32x32x16 8w 64B  mfma only      ", run_once<32, 8, 64, 2>},
      {"32x32x16 4w 64B  mfma only      ", run_once<32, 4, 64, 2>},
      {"16x16x32 8w 64B  reads          ", run_once<16, 8, 64, 0>},
      {"16x16x32 8w 64B  reads+barrier  ", run_once<16, 8, 64, 1>},
      {"16x16x32 8w 64B  reads+bar+prio ", run_once<16, 8, 64, 5>},
      {"16x16x32 8w 128B reads+barrier  ", run_once<16, 8, 128, 1>},
      {"32x32x16 8w 64B  reads          ", run_once<32, 8, 64, 0>},
      {"32x32x16 8w 64B  reads+barrier  ", run_once<32, 8, 64, 1>},
      {"32x32x16 8w 64B  reads+bar+prio ", run_once<32, 8, 64, 5>},
      {"32x32x16 8w 128B reads+barrier  ", run_once<32, 8, 128, 1>},
      {"32x32x16 4w 64B  reads          ", run_once<32, 4, 64, 0>},
      {"32x32x16 4w 64B  reads+barrier  ", run_once<32, 4, 64, 1>},
      {"32x32x16 4w 128B reads          ", run_once<32, 4, 128, 0>},
      {"32x32x16 4w 128B reads+barrier  ", run_once<32, 4, 128, 1>},
      {"16x16x32 4w 64B  reads+barrier  ", run_once<16, 4, 64, 1>},
  };
  const int nv = sizeof(vs) / sizeof(vs[0]);
  const int passes = 5;
  std::vector<std::vector<double>> r(nv);
  for (int v = 0; v < nv; ++v) (void)vs[v].fn(init, sink, 64);  // code objects loaded, clocks up
  for (int p = 0; p < passes; ++p)
    for (int v = 0; v < nv; ++v) r[v].push_back(vs[v].fn(init, sink, slabs));
  printf("# tests/native/mfma_bench %d %d: K loop alone, 256 work-groups x (256 ch x 256 px) tile, %d slabs of 32 channels each, %s operands;\n"
         "# TFLOP/s, median of %d interleaved passes (min .. max)\n", slabs, zero, slabs, zero ? "ZERO" : "random", passes);
  for (int v = 0; v < nv; ++v) {
    std::sort(r[v].begin(), r[v].end());
    printf("%s %7.0f   (%5.0f .. %5.0f)\n", vs[v].name, r[v][passes / 2], r[v][0], r[v][passes - 1]);
  }
  return 0;
}
