// Shared device helpers for the u2seg_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <mutex>

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// hipFuncSetAttribute (dynamic LDS above 64 KB) applies to the CURRENT device's copy of a kernel: remember it per device, not per
// process - a process may drive several GPUs - and from several host threads (main + autograd):
// `static PerDeviceOnce once; if (auto g = once.first()) { ...set... }` - the guard holds the lock until the block is left, so a
// second thread cannot launch between "somebody is setting the attribute" and "it is set".
struct PerDeviceOnce {
  std::mutex mu;
  std::atomic<bool> done[64];
  PerDeviceOnce() { for (auto& d : done) d.store(false, std::memory_order_relaxed); }
  struct Guard {
    PerDeviceOnce* o; int d;
    Guard(PerDeviceOnce* o_, int d_) : o(o_), d(d_) {}
    Guard(const Guard&) = delete;
    Guard& operator=(const Guard&) = delete;
    explicit operator bool() const { return o != nullptr; }
    ~Guard() {
      if (!o) return;
      if (d >= 0) o->done[d].store(true, std::memory_order_release);
      o->mu.unlock();
    }
  };
  Guard first() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = -1;  // unknown device: set the attribute again (cheap)
    if (d >= 0 && done[d].load(std::memory_order_acquire)) return Guard(nullptr, d);
    mu.lock();
    if (d >= 0 && done[d].load(std::memory_order_relaxed)) { mu.unlock(); return Guard(nullptr, d); }
    return Guard(this, d);
  }
};

#define U2_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define U2_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even (the rule of torch's float -> bfloat16).  Round 6: the hardware conversion (v_cvt_pk_bf16_f32, one
// instruction; adjacent conversions pair up) instead of the seven-instruction integer form - the element-wise passes spend
// 20-26 VALU instructions per element, two thirds of them in two roundings, and at 16 bytes per lane that is as long as their
// memory time.  Same result for every finite input and infinity; a NaN stays a NaN (quiet, payload not preserved).
#ifndef U2_SOFT_F2BF
__device__ __forceinline__ bf16_t f2bf(float f) {
  const __bf16 h = (__bf16)f;
  return *reinterpret_cast<const bf16_t*>(&h);
}
#else
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x == 256; `red` is >= 4 floats of LDS. Result valid in all threads.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// Last flush of a work-group's BN column sums.  Lane `lane` of wave `w` holds the sums (st_s, st_ss) of channel st_n (< 0 or
// >= N: none); waves that walked the same channels add up through LDS and the first of them sends the two atomics.
// Why: the work-groups of a persistent kernel all finish together and their atomics meet on the same 2 N addresses, where they
// serialize - 2048 per address with per-wave flushes of a 256 x 8-wave grid on 64 channels cost 36 us at the end of a 154 us
// launch (res2 conv2, profiles/r04_stats_flush.txt).  `smem` is the kernel's LDS (12 bytes x threads used); every wave of the
// work-group must call, after its last LDS read and with no LDS-DMA in flight.
template <int NW>
__device__ __forceinline__ void wg_flush_column_sums(float* stats, int N, int st_n, float st_s, float st_ss, int w, int lane,
                                                     unsigned char* smem) {
  __syncthreads();
  int* rn = reinterpret_cast<int*>(smem);
  float* rs = reinterpret_cast<float*>(smem) + NW * 64;
  float* rq = rs + NW * 64;
  rn[w * 64 + lane] = st_n;
  rs[w * 64 + lane] = st_s;
  rq[w * 64 + lane] = st_ss;
  __syncthreads();
  bool lead = st_n >= 0 && st_n < N;
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    const bool same = rn[k * 64 + lane] == st_n;
    if (same && k < w) lead = false;
    s += same ? rs[k * 64 + lane] : 0.f;
    q += same ? rq[k * 64 + lane] : 0.f;
  }
  if (lead) {
    atomicAdd(stats + st_n, s);
    atomicAdd(stats + N + st_n, q);
  }
}


// XCD-aware bijective remap of a linear block id: blocks that land on the same XCD (bid % 8)
// get a contiguous range of logical tiles, so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

#define U2_CHECK_LAUNCH() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)

// Zero-fill of a few words as an ordinary kernel of the stream.  hipMemsetAsync is not used for scratch that the next kernels
// of the same stream update with atomics: a k-means run of the test-suite was seen (1 run in 8) with the list counter of the
// screening pass reset WHILE the pass was appending to it - the 8-byte memset queued in front of two kernels had not been
// executed in front of them.
__global__ static void u2_fill_words_kernel(unsigned* __restrict__ p, size_t n, unsigned v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
static inline void u2_fill_words(void* p, size_t nwords, unsigned v, hipStream_t s) {
  if (nwords == 0) return;
  size_t g = (nwords + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(u2_fill_words_kernel, dim3((unsigned)g), dim3(256), 0, s, (unsigned*)p, nwords, v);
}
static inline void u2_zero_words(void* p, size_t nwords, hipStream_t s) { u2_fill_words(p, nwords, 0u, s); }

