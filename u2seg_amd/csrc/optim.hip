// Fused per-parameter gradient clipping + SGD(momentum, weight decay) over one flat fp32 parameter arena.
//
// Replaces detectron2/solver/build.py:36-37,63-73 (per-parameter clip_grad_norm_(p, 1.0, 2.0)) followed by
// torch.optim.SGD(momentum, weight_decay) built at solver/build.py:119-139.
// The arena is cut into fixed chunks; chunk_tensor[c] names the parameter tensor a chunk belongs to,
// chunk_begin/chunk_len its element range (a chunk never straddles two tensors).
#include "common.h"
#include "u2seg_hip.h"

namespace {

// A chunk's element range [b, b + n) of the fp32 arena as an unaligned head (< 4 elements, up to the next 16-byte boundary of the
// arena, whose base torch aligns to 512 bytes), a body of float4 and a tail: the optimizer's passes are pure streams over 0.9 GB
// (parameters and momentum read + written, gradients read; 45 M elements), and with one 4-byte load per lane and 256 threads per
// 65 536-element chunk they ran at 2.7 TB/s (sgd) / 1.4 TB/s (sumsq) - round 6: 1024 threads per chunk, 16-byte accesses.
constexpr int OPT_THREADS = 1024;

__device__ __forceinline__ void chunk_split(long long b, int n, int& head, int& nvec) {
  head = min((int)((4 - (b & 3)) & 3), n);
  nvec = (n - head) >> 2;
}

// per-chunk sum of squares (no atomics: the per-tensor total is formed in a fixed order so that every data-parallel
// rank computes bit-identical clip coefficients from the same all-reduced gradients)
__global__ __launch_bounds__(OPT_THREADS) void sumsq_kernel(const float* __restrict__ g, const long long* __restrict__ chunk_begin,
                                                            const int* __restrict__ chunk_len, float* __restrict__ partial) {
  __shared__ float red[OPT_THREADS / 64];
  const int c = blockIdx.x;
  const long long b = chunk_begin[c];
  const float* p = g + b;
  const int n = chunk_len[c];
  int head, nvec;
  chunk_split(b, n, head, nvec);
  float s = 0.f;
  const float4* pv = reinterpret_cast<const float4*>(p + head);
  for (int i = threadIdx.x; i < nvec; i += OPT_THREADS) {
    const float4 v = pv[i];
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  {  // head and tail elements (at most 3 + 3)
    const int rest = n - head - 4 * nvec, t = threadIdx.x;
    if (t < head) { const float v = p[t]; s += v * v; }
    else if (t - head < rest) { const float v = p[head + 4 * nvec + (t - head)]; s += v * v; }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < OPT_THREADS / 64; ++k) tot += red[k];
    partial[c] = tot;
  }
}

__global__ __launch_bounds__(OPT_THREADS) void sgd_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                          const int* __restrict__ chunk_tensor,
                                                          const long long* __restrict__ chunk_begin,
                                                          const int* __restrict__ chunk_len, const float* __restrict__ partial,
                                                          const int* __restrict__ tensor_first_chunk,
                                                          const float* __restrict__ wd_per_tensor, float lr, float momentum,
                                                          float clip, float grad_scale) {
  const int c = blockIdx.x;
  const int t = chunk_tensor[c];
  const long long b = chunk_begin[c];
  const int n = chunk_len[c];
  float coef = 1.f;
  if (clip > 0.f) {
    float n2 = 0.f;
    for (int j = tensor_first_chunk[t]; j < tensor_first_chunk[t + 1]; ++j) n2 += partial[j];
    const float total = sqrtf(n2) * grad_scale;
    coef = fminf(clip / (total + 1e-6f), 1.f);
  }
  coef *= grad_scale;
  const float wd = wd_per_tensor[t];
  auto upd = [&](float p, float gg, float& mm) {
    const float d = gg * coef + wd * p;
    mm = momentum * mm + d;
    return p - lr * mm;
  };
  int head, nvec;
  chunk_split(b, n, head, nvec);
  float4* wv = reinterpret_cast<float4*>(w + b + head);
  float4* mv = reinterpret_cast<float4*>(m + b + head);
  const float4* gv = reinterpret_cast<const float4*>(g + b + head);
  for (int i = threadIdx.x; i < nvec; i += OPT_THREADS) {
    float4 p = wv[i], mm = mv[i];
    const float4 gg = gv[i];
    p.x = upd(p.x, gg.x, mm.x);
    p.y = upd(p.y, gg.y, mm.y);
    p.z = upd(p.z, gg.z, mm.z);
    p.w = upd(p.w, gg.w, mm.w);
    mv[i] = mm;
    wv[i] = p;
  }
  {
    const int rest = n - head - 4 * nvec, k = threadIdx.x;
    long long i = -1;
    if (k < head) i = b + k;
    else if (k - head < rest) i = b + head + 4 * nvec + (k - head);
    if (i >= 0) {
      float mm = m[i];
      w[i] = upd(w[i], g[i], mm);
      m[i] = mm;
    }
  }
}

}  // namespace

extern "C" int u2_sgd_clip_step(float* params, const float* grads, float* momentum_buf, const int* chunk_tensor,
                                const long long* chunk_begin, const int* chunk_len, int n_chunks, float* partial,
                                const int* tensor_first_chunk, const float* wd_per_tensor, float lr, float momentum,
                                float clip, float grad_scale, void* stream) {
  if (n_chunks <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (clip > 0.f) {
    hipLaunchKernelGGL(sumsq_kernel, dim3(n_chunks), dim3(OPT_THREADS), 0, s, grads, chunk_begin, chunk_len, partial);
    U2_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(sgd_kernel, dim3(n_chunks), dim3(OPT_THREADS), 0, s, params, grads, momentum_buf, chunk_tensor, chunk_begin,
                     chunk_len, partial, tensor_first_chunk, wd_per_tensor, lr, momentum, clip, grad_scale);
  U2_CHECK_LAUNCH();
  return 0;
}
