// Fused per-parameter gradient clipping + SGD(momentum, weight decay) over one flat fp32 parameter arena.
//
// Replaces detectron2/solver/build.py:36-37,63-73 (per-parameter clip_grad_norm_(p, 1.0, 2.0)) followed by
// torch.optim.SGD(momentum, weight_decay) built at solver/build.py:119-139.
// The arena is cut into fixed chunks; chunk_tensor[c] names the parameter tensor a chunk belongs to,
// chunk_begin/chunk_len its element range (a chunk never straddles two tensors).
#include "common.h"
#include "u2seg_hip.h"

namespace {

// per-chunk sum of squares (no atomics: the per-tensor total is formed in a fixed order so that every data-parallel
// rank computes bit-identical clip coefficients from the same all-reduced gradients)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, const long long* __restrict__ chunk_begin,
                                                    const int* __restrict__ chunk_len, float* __restrict__ partial) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  const float* p = g + chunk_begin[c];
  const int n = chunk_len[c];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { const float v = p[i]; s += v * v; }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[c] = s;
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
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
  for (int i = threadIdx.x; i < n; i += 256) {
    const float p = w[b + i];
    const float d = g[b + i] * coef + wd * p;
    const float mm = momentum * m[b + i] + d;
    m[b + i] = mm;
    w[b + i] = p - lr * mm;
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
    hipLaunchKernelGGL(sumsq_kernel, dim3(n_chunks), dim3(256), 0, s, grads, chunk_begin, chunk_len, partial);
    U2_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(sgd_kernel, dim3(n_chunks), dim3(256), 0, s, params, grads, momentum_buf, chunk_tensor, chunk_begin,
                     chunk_len, partial, tensor_first_chunk, wd_per_tensor, lr, momentum, clip, grad_scale);
  U2_CHECK_LAUNCH();
  return 0;
}
