// Pooling / resampling kernels over NHWC bf16 activations (HBM-bound, 16-byte accesses).
//
// Replaces (reference file:line):
//   F.max_pool2d 3x3 s2 p1 in the stem            detectron2/modeling/backbone/resnet.py:358
//   nearest x2 upsample + lateral add             detectron2/modeling/backbone/fpn.py:153-155
//   nn.Upsample(scale 2, bilinear, ac=False)      detectron2/modeling/meta_arch/semantic_seg.py:206-211
//   image normalisation + zero pad + im2col       detectron2/modeling/meta_arch/rcnn.py:223-234 feeding the
//                                                 7x7 s2 stem conv of backbone/resnet.py:355-357
#include "common.h"
#include "u2seg_hip.h"

namespace {

int ew_grid(size_t total) {
  size_t g = (total + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

// ---- max pool 3x3 stride 2 pad 1; records the winning window slot (0..8, first max wins) ----
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                          uint8_t* __restrict__ idx, int B, int H, int W, int C, int Ho,
                                                          int Wo) {
  const int cpr = C >> 3;
  for (int row_ = blockIdx.y; row_ < B * Ho; row_ += gridDim.y)  // grid.y = (image, row): no 64-bit divisions per item
  for (int t = blockIdx.x * 256 + threadIdx.x; t < Wo * cpr; t += gridDim.x * 256) {
    const int b = row_ / Ho, oy = row_ - b * Ho;
    const int ox = t / cpr, cc = t - ox * cpr;
    float best[8];
    uint8_t bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        bf16_t v[8];
        *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + iy) * W + ix) * C + cc * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = bf2f(v[e]);
          if (f > best[e] || f != f) { best[e] = f; bi[e] = (uint8_t)(ky * 3 + kx); }
        }
      }
    }
    bf16_t o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(best[e]);
    const size_t off = (((size_t)b * Ho + oy) * Wo + ox) * C + cc * 8;
    *reinterpret_cast<uint4*>(y + off) = *reinterpret_cast<const uint4*>(o);
    *reinterpret_cast<uint2*>(idx + off) = *reinterpret_cast<const uint2*>(bi);
  }
}

// gather form of the backward: every input pixel visits the <=4 windows that contain it
typedef unsigned int mp_u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const bf16_t* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          bf16_t* __restrict__ dx, int B, int H, int W, int C, int Ho,
                                                          int Wo) {
  // grid.y = (image, input row): the row decomposition is scalar, a thread's (column, chunk) two 32-bit operations (three 64-bit
  // divisions per 16-byte item were most of this kernel: 0.40 ms for a 550 MB map)
  const int cpr = C >> 3;
  for (int row_ = blockIdx.y; row_ < B * H; row_ += gridDim.y)  // grid.y = (image, row): no 64-bit divisions per item
  for (int t = blockIdx.x * 256 + threadIdx.x; t < W * cpr; t += gridDim.x * 256) {
    const int b = row_ / H, iy = row_ - b * H;
    const int ix = t / cpr, cc = t - ix * cpr;
    const size_t i = ((size_t)row_ * W + ix) * cpr + cc;
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = 0.f;
    // windows oy with oy*2-1 <= iy <= oy*2+1
    const int oy_lo = iy >> 1, oy_hi = (iy + 1) >> 1;
    const int ox_lo = ix >> 1, ox_hi = (ix + 1) >> 1;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      if (oy >= Ho) continue;
      const int ky = iy - (oy * 2 - 1);
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        if (ox >= Wo) continue;
        const int kx = ix - (ox * 2 - 1);
        const uint8_t slot = (uint8_t)(ky * 3 + kx);
        const size_t off = (((size_t)b * Ho + oy) * Wo + ox) * C + cc * 8;
        uint8_t bi[8];
        bf16_t dv[8];
        *reinterpret_cast<uint2*>(bi) = *reinterpret_cast<const uint2*>(idx + off);
        *reinterpret_cast<uint4*>(dv) = *reinterpret_cast<const uint4*>(dy + off);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (bi[e] == slot) g[e] += bf2f(dv[e]);
      }
    }
    bf16_t o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(g[e]);
    if (NT) __builtin_nontemporal_store(*reinterpret_cast<const mp_u32x4*>(o), reinterpret_cast<mp_u32x4*>(dx + i * 8));
    else *reinterpret_cast<uint4*>(dx + i * 8) = *reinterpret_cast<const uint4*>(o);
  }
}

// ---- Round 6: the stem's tail as one pass each way (backbone/resnet.py:355-359: conv1 -> norm -> relu_ -> max_pool2d) ----
// The ReLU output of the stem's normalisation (550 MB at batch 16 x 800 x 1344) has one reader, the max pool, and its gradient
// one writer, the pool's backward pass.  Forward: the pool evaluates relu(bf16(x * scale + shift)) on the nine taps itself
// (affine_act_fast_kernel's expression and rounding, then maxpool_fwd_kernel's comparison on the ROUNDED values: pooled values and
// winner slots are bit-identical to the two launches) - 1.1 GB less.  Backward: the column reduction and the apply pass of the
// normalisation rebuild the pool's gradient of an input pixel from the <= 4 windows that contain it (maxpool_bwd_kernel's sum,
// rounded to bf16 as the stored map was) instead of reading it: the 550 MB map is neither written nor read twice.
__device__ __forceinline__ void mp_load_coef(const float* __restrict__ a, int c, float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = a[c + e];
}
// two floats -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32: one instruction where f2bf is seven; these passes
// are VALU-bound - the first form of the forward pass spent 1 200 instructions per 16-byte item and ran at 1.8 TB/s)
typedef __attribute__((ext_vector_type(2))) __bf16 mp_bf16x2;
typedef __attribute__((ext_vector_type(2))) float mp_f32x2;
__device__ __forceinline__ uint32_t mp_pack(float lo, float hi) {
  const mp_f32x2 v = {lo, hi};
  const mp_bf16x2 r = __builtin_convertvector(v, mp_bf16x2);
  return *reinterpret_cast<const uint32_t*>(&r);
}

// Forward.  The activation of a tap is >= 0 (ReLU), so its bf16 pattern orders like an unsigned integer: the running maximum and
// its slot are ONE v_max_u32 on key = pattern << 4 | (8 - slot) - among equal values the lowest slot wins, max_pool2d's rule and
// maxpool_fwd_kernel's.  (A -0 the ReLU may leave compares as +0 and is stored as +0; NaN patterns sort above infinity.)
__global__ __launch_bounds__(256) void affine_relu_maxpool_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                                                      const float* __restrict__ shift, bf16_t* __restrict__ y,
                                                                      uint8_t* __restrict__ idx, int B, int H, int W, int C, int Ho,
                                                                      int Wo) {
  const int cpr = C >> 3;   // host: cpr divides 256, so a thread keeps its channel chunk for its whole life
  const int cc = threadIdx.x % cpr;
  float sc[8], sh[8];
  mp_load_coef(scale, cc * 8, sc);
  mp_load_coef(shift, cc * 8, sh);
  for (int row_ = blockIdx.y; row_ < B * Ho; row_ += gridDim.y)
  for (int t = blockIdx.x * 256 + threadIdx.x; t < Wo * cpr; t += gridDim.x * 256) {
    const int b = row_ / Ho, oy = row_ - b * Ho;
    const int ox = t / cpr;
    uint4 q[9];
    bool ok[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
        ok[ky * 3 + kx] = iy >= 0 && iy < H && ix >= 0 && ix < W;
        if (ok[ky * 3 + kx])
          q[ky * 3 + kx] = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + iy) * W + ix) * C + cc * 8);
      }
    uint32_t key[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) key[e] = 0u;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      if (!ok[k]) continue;
      const uint32_t* v = reinterpret_cast<const uint32_t*>(&q[k]);
      const uint32_t code = (uint32_t)(8 - k);
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const float f0 = fmaxf(__uint_as_float(v[e2] << 16) * sc[2 * e2] + sh[2 * e2], 0.f);
        const float f1 = fmaxf(__uint_as_float(v[e2] & 0xffff0000u) * sc[2 * e2 + 1] + sh[2 * e2 + 1], 0.f);
        const uint32_t pk = mp_pack(f0, f1) & 0x7fff7fffu;   // the activation as affine_act stores it
        key[2 * e2] = max(key[2 * e2], ((pk << 4) & 0xffff0u) | code);
        key[2 * e2 + 1] = max(key[2 * e2 + 1], ((pk >> 12) & 0xffff0u) | code);
      }
    }
    uint32_t o[4], bi[2] = {0u, 0u};
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) o[e2] = (key[2 * e2] >> 4) | ((key[2 * e2 + 1] >> 4) << 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) bi[e >> 2] |= (8u - (key[e] & 15u)) << (8 * (e & 3));
    const size_t off = (((size_t)b * Ho + oy) * Wo + ox) * C + cc * 8;
    *reinterpret_cast<uint4*>(y + off) = *reinterpret_cast<const uint4*>(o);
    *reinterpret_cast<uint2*>(idx + off) = *reinterpret_cast<const uint2*>(bi);
  }
}

// Backward.  A thread takes a 2 x 2 block of input pixels {2a, 2a + 1} x {2c, 2c + 1} and one 8-channel chunk: the block lies in the
// four windows (a .. a + 1) x (c .. c + 1), loaded once, and its pixels sit at FIXED slots of them - (2a, 2c): window (a, c) slot 4;
// (2a, 2c + 1): (a, c) 5, (a, c + 1) 3; (2a + 1, 2c): (a, c) 7, (a + 1, c) 1; (2a + 1, 2c + 1): (a, c) 8, (a, c + 1) 6, (a + 1, c) 2,
// (a + 1, c + 1) 0 - nine compare-and-add steps for four pixels, no divergence (the per-pixel form walked up to four windows per
// pixel with the lanes of a wave on both parities).  A pixel's gradient is the fp32 sum over its windows in maxpool_bwd_kernel's
// order, rounded to bf16 as the stored map was.
// APPLY = false: sums[0][c] += sum dz, sums[1][c] += sum dz * (x - mean) * invstd  (colreduce_kernel<1, 2, false>'s quantities);
// APPLY = true:  dx = k1 dz + k2 x + k3  (norm_bwd_apply_fast_kernel<2>'s expression);  dz = pool gradient where x * msc + msh > 0
template <bool APPLY>
__global__ __launch_bounds__(256) void affine_relu_maxpool_bwd_kernel(const bf16_t* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                                      const bf16_t* __restrict__ x, const float* __restrict__ p1,
                                                                      const float* __restrict__ p2, const float* __restrict__ p3,
                                                                      const float* __restrict__ msc, const float* __restrict__ msh,
                                                                      float* __restrict__ sums, bf16_t* __restrict__ dx, int B, int H,
                                                                      int W, int C, int Ho, int Wo) {
  __shared__ float part[2][2048];
  const int cpr = C >> 3;
  const int cc = threadIdx.x % cpr;
  const int Hb = (H + 1) >> 1, Wb = (W + 1) >> 1;
  float a1[8], a2[8], a3[8], ms[8], mh[8], s0[8], s1[8];
  mp_load_coef(p1, cc * 8, a1);   // APPLY: k1, k2, k3;  reduce: mean, invstd
  mp_load_coef(p2, cc * 8, a2);
  if (APPLY) mp_load_coef(p3, cc * 8, a3);
  mp_load_coef(msc, cc * 8, ms);
  mp_load_coef(msh, cc * 8, mh);
#pragma unroll
  for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
  for (int rp = blockIdx.y; rp < B * Hb; rp += gridDim.y)
  for (int t = blockIdx.x * 256 + threadIdx.x; t < Wb * cpr; t += gridDim.x * 256) {
    const int b = rp / Hb, a = rp - b * Hb;
    const int c = t / cpr;
    // the four windows: 0 = (a, c), 1 = (a, c + 1), 2 = (a + 1, c), 3 = (a + 1, c + 1)
    uint2 iq[4];
    uint4 dq[4], xq[4];
    bool wv[4], pv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int oy = a + (k >> 1), ox = c + (k & 1);
      wv[k] = oy < Ho && ox < Wo;
      iq[k] = uint2{0xffffffffu, 0xffffffffu};   // slot 255: matches nothing
      dq[k] = uint4{0u, 0u, 0u, 0u};
      if (wv[k]) {
        const size_t off = (((size_t)b * Ho + oy) * Wo + ox) * C + cc * 8;
        iq[k] = *reinterpret_cast<const uint2*>(idx + off);
        dq[k] = *reinterpret_cast<const uint4*>(dy + off);
      }
      const int iy = 2 * a + (k >> 1), ix = 2 * c + (k & 1);
      pv[k] = iy < H && ix < W;
      if (pv[k]) {
        const mp_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const mp_u32x4*>(x + (((size_t)b * H + iy) * W + ix) * C + cc * 8));
        xq[k] = uint4{v.x, v.y, v.z, v.w};
      }
    }
    float dv[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t* d = reinterpret_cast<const uint32_t*>(&dq[k]);
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        dv[k][2 * e2] = __uint_as_float(d[e2] << 16);
        dv[k][2 * e2 + 1] = __uint_as_float(d[e2] & 0xffff0000u);
      }
    }
    // (window, slot) pairs of pixel p, in maxpool_bwd_kernel's order
    constexpr int NPAIR[4] = {1, 2, 2, 4};
    constexpr int PW[4][4] = {{0, 0, 0, 0}, {0, 1, 0, 0}, {0, 2, 0, 0}, {0, 1, 2, 3}};
    constexpr int PS[4][4] = {{4, 0, 0, 0}, {5, 3, 0, 0}, {7, 1, 0, 0}, {8, 6, 2, 0}};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (!pv[p]) continue;
      float g[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j >= NPAIR[p]) continue;
        const int k = PW[p][j];
        const uint32_t pat = (uint32_t)PS[p][j] * 0x01010101u;
        const uint32_t m0 = iq[k].x ^ pat, m1 = iq[k].y ^ pat;   // a zero byte = this pixel is the window's winner
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t byte = ((e < 4 ? m0 : m1) >> (8 * (e & 3))) & 0xffu;
          g[e] += byte == 0u ? dv[k][e] : 0.f;
        }
      }
      const uint32_t* xv = reinterpret_cast<const uint32_t*>(&xq[p]);
      uint32_t o[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const uint32_t gp = mp_pack(g[2 * e2], g[2 * e2 + 1]);   // the gradient map as maxpool_bwd_kernel stores it
        float dz[2] = {__uint_as_float(gp << 16), __uint_as_float(gp & 0xffff0000u)};
        const float xf[2] = {__uint_as_float(xv[e2] << 16), __uint_as_float(xv[e2] & 0xffff0000u)};
        float r[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int e = 2 * e2 + h;
          if (!(xf[h] * ms[e] + mh[e] > 0.f)) dz[h] = 0.f;
          if (APPLY) {
            r[h] = a1[e] * dz[h] + a2[e] * xf[h] + a3[e];
          } else {
            s0[e] += dz[h];
            s1[e] += dz[h] * (xf[h] - a1[e]) * a2[e];
          }
        }
        if (APPLY) o[e2] = mp_pack(r[0], r[1]);
      }
      if (APPLY) {
        const int iy = 2 * a + (p >> 1), ix = 2 * c + (p & 1);
        __builtin_nontemporal_store(mp_u32x4{o[0], o[1], o[2], o[3]},
                                    reinterpret_cast<mp_u32x4*>(dx + (((size_t)b * H + iy) * W + ix) * C + cc * 8));
      }
    }
  }
  if (!APPLY) {
    const int rl = threadIdx.x / cpr, rows_par = 256 / cpr;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      part[0][rl * cpr * 8 + cc * 8 + e] = s0[e];
      part[1][rl * cpr * 8 + cc * 8 + e] = s1[e];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < cpr * 8; j += 256) {
      float t0 = 0.f, t1 = 0.f;
      for (int q = 0; q < rows_par; ++q) { t0 += part[0][q * cpr * 8 + j]; t1 += part[1][q * cpr * 8 + j]; }
      atomicAdd(sums + j, t0);
      atomicAdd(sums + C + j, t1);
    }
  }
}

// ---- FPN top-down: out = lateral + nearest_x2(top) ----
__global__ __launch_bounds__(256) void upadd_fwd_kernel(const bf16_t* __restrict__ lat, const bf16_t* __restrict__ top,
                                                        bf16_t* __restrict__ out, int B, int H, int W, int C) {
  const int cpr = C >> 3;
  const int Ht = H >> 1, Wt = W >> 1;
  for (int row_ = blockIdx.y; row_ < B * H; row_ += gridDim.y)  // grid.y = (image, row): no 64-bit divisions per item
  for (int t_ = blockIdx.x * 256 + threadIdx.x; t_ < W * cpr; t_ += gridDim.x * 256) {
    const int b = row_ / H, y = row_ - b * H;
    const int x = t_ / cpr, cc = t_ - x * cpr;
    const size_t i = ((size_t)row_ * W + x) * cpr + cc;
    bf16_t a[8], t[8], o[8];
    *reinterpret_cast<uint4*>(a) = *reinterpret_cast<const uint4*>(lat + i * 8);
    *reinterpret_cast<uint4*>(t) =
        *reinterpret_cast<const uint4*>(top + (((size_t)b * Ht + (y >> 1)) * Wt + (x >> 1)) * C + cc * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(a[e]) + bf2f(t[e]));
    *reinterpret_cast<uint4*>(out + i * 8) = *reinterpret_cast<const uint4*>(o);
  }
}

// dtop[y][x] = sum of the 2x2 block of dout
__global__ __launch_bounds__(256) void upadd_bwd_kernel(const bf16_t* __restrict__ dout, bf16_t* __restrict__ dtop, int B,
                                                        int H, int W, int C) {
  const int cpr = C >> 3;
  const int Ht = H >> 1, Wt = W >> 1;
  for (int row_ = blockIdx.y; row_ < B * Ht; row_ += gridDim.y)  // grid.y = (image, row): no 64-bit divisions per item
  for (int t_ = blockIdx.x * 256 + threadIdx.x; t_ < Wt * cpr; t_ += gridDim.x * 256) {
    const int b = row_ / Ht, y = row_ - b * Ht;
    const int x = t_ / cpr, cc = t_ - x * cpr;
    const size_t i = ((size_t)row_ * Wt + x) * cpr + cc;
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        bf16_t v[8];
        *reinterpret_cast<uint4*>(v) =
            *reinterpret_cast<const uint4*>(dout + (((size_t)b * H + 2 * y + dy) * W + 2 * x + dx) * C + cc * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] += bf2f(v[e]);
      }
    bf16_t o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(g[e]);
    *reinterpret_cast<uint4*>(dtop + i * 8) = *reinterpret_cast<const uint4*>(o);
  }
}

// ---- bilinear x2, align_corners=False: src = (dst + 0.5)/2 - 0.5 clamped at 0 ----
__device__ __forceinline__ void bil2_src(int d, int n_in, int& i0, int& i1, float& l1) {
  float s = (d + 0.5f) * 0.5f - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

__global__ __launch_bounds__(256) void bilinear2_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ addend,
                                                            bf16_t* __restrict__ out, int B, int H, int W, int C) {
  const int cpr = C >> 3;
  const int Ho = H * 2, Wo = W * 2;
  for (int row_ = blockIdx.y; row_ < B * Ho; row_ += gridDim.y)  // grid.y = (image, row): no 64-bit divisions per item
  for (int t_ = blockIdx.x * 256 + threadIdx.x; t_ < Wo * cpr; t_ += gridDim.x * 256) {
    const int b = row_ / Ho, oy = row_ - b * Ho;
    const int ox = t_ / cpr, cc = t_ - ox * cpr;
    const size_t i = ((size_t)row_ * Wo + ox) * cpr + cc;
    int y0, y1, x0, x1;
    float ly, lx;
    bil2_src(oy, H, y0, y1, ly);
    bil2_src(ox, W, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    bf16_t v00[8], v01[8], v10[8], v11[8], o[8];
    const bf16_t* base = x + (size_t)b * H * W * C + cc * 8;
    *reinterpret_cast<uint4*>(v00) = *reinterpret_cast<const uint4*>(base + ((size_t)y0 * W + x0) * C);
    *reinterpret_cast<uint4*>(v01) = *reinterpret_cast<const uint4*>(base + ((size_t)y0 * W + x1) * C);
    *reinterpret_cast<uint4*>(v10) = *reinterpret_cast<const uint4*>(base + ((size_t)y1 * W + x0) * C);
    *reinterpret_cast<uint4*>(v11) = *reinterpret_cast<const uint4*>(base + ((size_t)y1 * W + x1) * C);
    if (addend) *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(addend + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float f = hy * (hx * bf2f(v00[e]) + lx * bf2f(v01[e])) + ly * (hx * bf2f(v10[e]) + lx * bf2f(v11[e]));
      // meta_arch/semantic_seg.py:240-244 under autocast: nn.Upsample returns a bf16 tensor, the running sum is a bf16 add -
      // the interpolated value is rounded before the sum, as the two separate passes would
      if (addend) f = bf2f(f2bf(f)) + bf2f(o[e]);
      o[e] = f2bf(f);
    }
    *reinterpret_cast<uint4*>(out + i * 8) = *reinterpret_cast<const uint4*>(o);
  }
}

// Round 6: the same pass by blocks of 2 x 2 output pixels.  Output rows 2k + 1 and 2k + 2 (k = -1 .. H - 1) read the SAME two source
// rows (k, k + 1, clamped: bil2_src of either row returns them) with weights 0.25 / 0.75, and so do the columns: a thread loads
// four taps for up to four outputs (the per-pixel kernel: four taps per output, 16-byte loads at 17 TB/s out of L1 / L2 for a pass
// that moves 1 GB), and the horizontal step hx * v00 + lx * v01 is shared by the two rows.  Every output is formed by the per-pixel
// kernel's expression with bil2_src's own weights: bit-identical (tests: the semantic-head chain against the oracle, and
// test_bilinear_up2_blocked_equals_per_pixel against the kernel above, which stays as the reference form).
__global__ __launch_bounds__(256) void bilinear2_fwd_block_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ addend,
                                                                  bf16_t* __restrict__ out, int B, int H, int W, int C) {
  const int cpr = C >> 3;
  const int Ho = H * 2, Wo = W * 2;
  for (int row_ = blockIdx.y; row_ < B * (H + 1); row_ += gridDim.y)
  for (int t_ = blockIdx.x * 256 + threadIdx.x; t_ < (W + 1) * cpr; t_ += gridDim.x * 256) {
    const int b = row_ / (H + 1), k = row_ - b * (H + 1) - 1;
    const int jj = t_ / cpr, cc = t_ - jj * cpr;
    const int j = jj - 1;
    // output rows / columns of the block; the first valid one names the taps (the other one, where it exists, has the same)
    const int oya = 2 * k + 1, oyb = 2 * k + 2, oxa = 2 * j + 1, oxb = 2 * j + 2;
    const bool va = oya >= 0, vb = oyb < Ho, ua = oxa >= 0, ub = oxb < Wo;
    int y0, y1, x0, x1, t0, t1;
    float lya = 0.f, lyb = 0.f, lxa = 0.f, lxb = 0.f;
    if (va) bil2_src(oya, H, y0, y1, lya);
    if (vb) { if (va) bil2_src(oyb, H, t0, t1, lyb); else bil2_src(oyb, H, y0, y1, lyb); }
    if (ua) bil2_src(oxa, W, x0, x1, lxa);
    if (ub) { if (ua) bil2_src(oxb, W, t0, t1, lxb); else bil2_src(oxb, W, x0, x1, lxb); }
    const bf16_t* base = x + (size_t)b * H * W * C + cc * 8;
    const uint4 q00 = *reinterpret_cast<const uint4*>(base + ((size_t)y0 * W + x0) * C);
    const uint4 q01 = *reinterpret_cast<const uint4*>(base + ((size_t)y0 * W + x1) * C);
    const uint4 q10 = *reinterpret_cast<const uint4*>(base + ((size_t)y1 * W + x0) * C);
    const uint4 q11 = *reinterpret_cast<const uint4*>(base + ((size_t)y1 * W + x1) * C);
    uint4 aq[4];
    size_t at[4];
    bool ok[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int oy = (p >> 1) ? oyb : oya, ox = (p & 1) ? oxb : oxa;
      ok[p] = ((p >> 1) ? vb : va) && ((p & 1) ? ub : ua);
      at[p] = ((((size_t)b * Ho + oy) * Wo + ox) * cpr + cc) * 8;
      if (ok[p] && addend) aq[p] = *reinterpret_cast<const uint4*>(addend + at[p]);
    }
    const bf16_t* v00 = reinterpret_cast<const bf16_t*>(&q00);
    const bf16_t* v01 = reinterpret_cast<const bf16_t*>(&q01);
    const bf16_t* v10 = reinterpret_cast<const bf16_t*>(&q10);
    const bf16_t* v11 = reinterpret_cast<const bf16_t*>(&q11);
    bf16_t o[4][8];
#pragma unroll
    for (int cx = 0; cx < 2; ++cx) {
      const float lx = cx ? lxb : lxa, hx = 1.f - lx;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float top = hx * bf2f(v00[e]) + lx * bf2f(v01[e]);
        const float bot = hx * bf2f(v10[e]) + lx * bf2f(v11[e]);
#pragma unroll
        for (int ry = 0; ry < 2; ++ry) {
          const float ly = ry ? lyb : lya, hy = 1.f - ly;
          float f = hy * top + ly * bot;
          const int p = ry * 2 + cx;
          if (addend) f = bf2f(f2bf(f)) + bf2f(reinterpret_cast<const bf16_t*>(&aq[p])[e]);
          o[p][e] = f2bf(f);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (ok[p]) *reinterpret_cast<uint4*>(out + at[p]) = *reinterpret_cast<const uint4*>(o[p]);
  }
}

// gather form of the transpose: input pixel (y,x) collects from output rows {2y-1..2y+2} that reference it
__global__ __launch_bounds__(256) void bilinear2_bwd_kernel(const bf16_t* __restrict__ dout, bf16_t* __restrict__ dx, int B,
                                                            int H, int W, int C) {
  const int cpr = C >> 3;
  const int Ho = H * 2, Wo = W * 2;
  for (int row_ = blockIdx.y; row_ < B * H; row_ += gridDim.y)  // grid.y = (image, row): no 64-bit divisions per item
  for (int t_ = blockIdx.x * 256 + threadIdx.x; t_ < W * cpr; t_ += gridDim.x * 256) {
    const int b = row_ / H, y = row_ - b * H;
    const int x = t_ / cpr, cc = t_ - x * cpr;
    const size_t i = ((size_t)row_ * W + x) * cpr + cc;
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = 0.f;
    for (int oy = max(0, 2 * y - 2); oy <= min(Ho - 1, 2 * y + 2); ++oy) {
      int y0, y1; float ly;
      bil2_src(oy, H, y0, y1, ly);
      float wy = 0.f;
      if (y0 == y) wy += 1.f - ly;
      if (y1 == y) wy += ly;
      if (wy == 0.f) continue;
      for (int ox = max(0, 2 * x - 2); ox <= min(Wo - 1, 2 * x + 2); ++ox) {
        int x0, x1; float lx;
        bil2_src(ox, W, x0, x1, lx);
        float wx = 0.f;
        if (x0 == x) wx += 1.f - lx;
        if (x1 == x) wx += lx;
        if (wx == 0.f) continue;
        bf16_t v[8];
        *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(dout + (((size_t)b * Ho + oy) * Wo + ox) * C + cc * 8);
        const float wgt = wy * wx;
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] += wgt * bf2f(v[e]);
      }
    }
    bf16_t o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(g[e]);
    *reinterpret_cast<uint4*>(dx + i * 8) = *reinterpret_cast<const uint4*>(o);
  }
}

// ---- stem im2col: uint8/float CHW image -> (x-mean)/std -> zero pad -> [B*Ho*Wo][KP] bf16 rows with
//      K ordered (kh, kw, c) to match weights permuted to [64][7][7][3] and zero padded to KP.
//      One work-group = 8 x 32 output pixels of one image: the 21 x 69 x 3 input window is normalised once into LDS
//      (zero outside the image), then every 16-byte chunk of the 256 K-rows is assembled from LDS through a k -> offset
//      table and stored coalesced (320 B per pixel row). ----
constexpr int ST_TH = 8, ST_TW = 32, ST_IH = 2 * ST_TH + 5, ST_IW = 2 * ST_TW + 5, ST_IWP = ST_IW + 1;
struct StemImages { const void* img[32]; int h[32]; int w[32]; };

template <typename T>
__global__ __launch_bounds__(256) void stem_im2col_kernel(const StemImages imgs, const float* __restrict__ mean,
                                                          const float* __restrict__ stdv, bf16_t* __restrict__ col, int b0,
                                                          int Ho, int Wo, int KP) {
  __shared__ bf16_t tile[3 * ST_IH * ST_IWP];
  __shared__ __attribute__((aligned(16))) short koff[512];  // K index -> offset inside the tile, -1 = zero padding
  const int tid = threadIdx.x;
  const int bi = blockIdx.z;
  const T* __restrict__ img = reinterpret_cast<const T*>(imgs.img[bi]);
  const int h = imgs.h[bi], w = imgs.w[bi];
  const int oy0 = blockIdx.y * ST_TH, ox0 = blockIdx.x * ST_TW;
  const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
  const size_t plane = (size_t)h * w;
  for (int i = tid; i < 3 * ST_IH * ST_IW; i += 256) {
    const int x = i % ST_IW;
    const int r = (i / ST_IW) % ST_IH;
    const int c = i / (ST_IW * ST_IH);
    const int iy = iy0 + r, ix = ix0 + x;
    float v = 0.f;
    if (iy >= 0 && iy < h && ix >= 0 && ix < w) v = ((float)img[c * plane + (size_t)iy * w + ix] - mean[c]) / stdv[c];
    tile[(c * ST_IH + r) * ST_IWP + x] = f2bf(v);
  }
  for (int k = tid; k < KP; k += 256) {
    short o = -1;
    if (k < 147) {
      const int c = k % 3, t = k / 3;
      o = (short)((c * ST_IH + t / 7) * ST_IWP + t % 7);
    }
    koff[k] = o;
  }
  __syncthreads();
  const int cpr = KP >> 3;
  for (int i = tid; i < ST_TH * ST_TW * cpr; i += 256) {
    const int cc = i % cpr;
    const int p = i / cpr;
    const int lx = p % ST_TW, ly = p / ST_TW;
    const int oy = oy0 + ly, ox = ox0 + lx;
    if (oy >= Ho || ox >= Wo) continue;
    const int base = (2 * ly) * ST_IWP + 2 * lx;
    short ko[8];
    *reinterpret_cast<uint4*>(ko) = *reinterpret_cast<const uint4*>(koff + cc * 8);
    bf16_t o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = ko[e] >= 0 ? tile[base + ko[e]] : (bf16_t)0;
    *reinterpret_cast<uint4*>(col + ((size_t)(b0 + bi) * Ho * Wo + (size_t)oy * Wo + ox) * KP + cc * 8) =
        *reinterpret_cast<const uint4*>(o);
  }
}

}  // namespace

static unsigned rows_grid(long long rows) { return (unsigned)(rows < 65535 ? (rows > 0 ? rows : 1) : 65535); }

extern "C" int u2_maxpool3x3s2_fwd(const void* x, void* y, void* idx, int B, int H, int W, int C, void* stream) {
  if (C & 7) return -1;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const size_t total = (size_t)B * Ho * Wo * (C >> 3);
  if (!total) return 0;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((Wo * (C >> 3) + 255) / 256, rows_grid(B * Ho)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)y, (uint8_t*)idx, B, H, W, C, Ho, Wo);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_maxpool3x3s2_bwd(const void* dy, const void* idx, void* dx, int B, int H, int W, int C, void* stream) {
  if (C & 7) return -1;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const size_t total = (size_t)B * H * W * (C >> 3);
  if (!total) return 0;
  // the stem's 550 MB gradient map is beyond the MALL: non-temporal stores, 401 -> 326 us
  if (total * 16 > ((size_t)256 << 20))
    hipLaunchKernelGGL(maxpool_bwd_kernel<true>, dim3((W * (C >> 3) + 255) / 256, rows_grid(B * H)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dy, (const uint8_t*)idx, (bf16_t*)dx, B, H, W, C, Ho, Wo);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel<false>, dim3((W * (C >> 3) + 255) / 256, rows_grid(B * H)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dy, (const uint8_t*)idx, (bf16_t*)dx, B, H, W, C, Ho, Wo);
  U2_CHECK_LAUNCH();
  return 0;
}

static bool stem_tail_ok(int C) { const int cpr = C >> 3; return (C & 7) == 0 && cpr >= 1 && cpr <= 256 && 256 % cpr == 0; }

extern "C" int u2_affine_relu_maxpool_fwd(const void* x, const float* scale, const float* shift, void* y, void* idx, int B, int H,
                                          int W, int C, void* stream) {
  if (!stem_tail_ok(C)) return -1;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  if (!((size_t)B * Ho * Wo)) return 0;
  hipLaunchKernelGGL(affine_relu_maxpool_fwd_kernel, dim3((Wo * (C >> 3) + 255) / 256, rows_grid(B * Ho)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)x, scale, shift, (bf16_t*)y, (uint8_t*)idx, B, H, W, C, Ho, Wo);
  U2_CHECK_LAUNCH();
  return 0;
}

// rows of the grid: the reduction keeps its sums per thread and flushes once per work-group (2 C atomics), so its work-groups walk
// many rows; the apply pass takes the same shape
static dim3 stem_tail_bwd_grid(int B, int H, int W, int C) {
  const int Hb = (H + 1) / 2, Wb = (W + 1) / 2;   // a thread takes a 2 x 2 block of input pixels
  const int gx = (Wb * (C >> 3) + 255) / 256;
  int gy = 4096 / gx;
  if (gy < 1) gy = 1;
  if (gy > B * Hb) gy = B * Hb;
  return dim3(gx, gy);
}

extern "C" int u2_affine_relu_maxpool_bwd_reduce(const void* dy, const void* idx, const void* x, const float* mean,
                                                 const float* invstd, const float* mask_scale, const float* mask_shift, float* sums,
                                                 int B, int H, int W, int C, void* stream) {
  if (!stem_tail_ok(C)) return -1;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  if (!((size_t)B * H * W)) return 0;
  hipLaunchKernelGGL(affine_relu_maxpool_bwd_kernel<false>, stem_tail_bwd_grid(B, H, W, C), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, (const uint8_t*)idx, (const bf16_t*)x, mean, invstd, (const float*)nullptr, mask_scale,
                     mask_shift, sums, (bf16_t*)nullptr, B, H, W, C, Ho, Wo);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_affine_relu_maxpool_bwd_apply(const void* dy, const void* idx, const void* x, const float* k1, const float* k2,
                                                const float* k3, const float* mask_scale, const float* mask_shift, void* dx, int B,
                                                int H, int W, int C, void* stream) {
  if (!stem_tail_ok(C)) return -1;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  if (!((size_t)B * H * W)) return 0;
  hipLaunchKernelGGL(affine_relu_maxpool_bwd_kernel<true>, stem_tail_bwd_grid(B, H, W, C), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, (const uint8_t*)idx, (const bf16_t*)x, k1, k2, k3, mask_scale, mask_shift, (float*)nullptr,
                     (bf16_t*)dx, B, H, W, C, Ho, Wo);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_fpn_upsample_add_fwd(const void* lateral, const void* top, void* out, int B, int H, int W, int C,
                                       void* stream) {
  if ((C & 7) || (H & 1) || (W & 1)) return -1;
  const size_t total = (size_t)B * H * W * (C >> 3);
  if (!total) return 0;
  hipLaunchKernelGGL(upadd_fwd_kernel, dim3((W * (C >> 3) + 255) / 256, rows_grid(B * H)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)lateral,
                     (const bf16_t*)top, (bf16_t*)out, B, H, W, C);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_fpn_upsample_add_bwd(const void* dout, void* dtop, int B, int H, int W, int C, void* stream) {
  if ((C & 7) || (H & 1) || (W & 1)) return -1;
  const size_t total = (size_t)B * (H / 2) * (W / 2) * (C >> 3);
  if (!total) return 0;
  hipLaunchKernelGGL(upadd_bwd_kernel, dim3(((W / 2) * (C >> 3) + 255) / 256, rows_grid(B * (H / 2))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout,
                     (bf16_t*)dtop, B, H, W, C);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_bilinear_up2_fwd(const void* x, const void* addend, void* out, int B, int H, int W, int C, void* stream) {
  if (C & 7) return -1;
  const size_t total = (size_t)B * H * 2 * W * 2 * (C >> 3);
  if (!total) return 0;
  // U2_BILINEAR_PER_PIXEL=1: the round-1 kernel, one output pixel per thread (A/B runs and the blocked kernel's test; read per call)
  const char* pp = getenv("U2_BILINEAR_PER_PIXEL");
  if (pp && atoi(pp) == 1)
    hipLaunchKernelGGL(bilinear2_fwd_kernel, dim3((W * 2 * (C >> 3) + 255) / 256, rows_grid(B * H * 2)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (const bf16_t*)addend, (bf16_t*)out, B, H, W, C);
  else
    hipLaunchKernelGGL(bilinear2_fwd_block_kernel, dim3(((W + 1) * (C >> 3) + 255) / 256, rows_grid((long long)B * (H + 1))), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)addend, (bf16_t*)out, B, H, W, C);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_bilinear_up2_bwd(const void* dout, void* dx, int B, int H, int W, int C, void* stream) {
  if (C & 7) return -1;
  const size_t total = (size_t)B * H * W * (C >> 3);
  if (!total) return 0;
  hipLaunchKernelGGL(bilinear2_bwd_kernel, dim3((W * (C >> 3) + 255) / 256, rows_grid(B * H)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout,
                     (bf16_t*)dx, B, H, W, C);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_stem_im2col_batch(const void* const* imgs, const int* hs, const int* ws, int n_imgs, int is_uint8,
                                    const float* mean, const float* stdv, void* col, int Hpad, int Wpad, int KP,
                                    void* stream) {
  if (KP < 147 || (KP & 31) || KP > 512) return -1;
  const int Ho = (Hpad + 6 - 7) / 2 + 1, Wo = (Wpad + 6 - 7) / 2 + 1;
  for (int b0 = 0; b0 < n_imgs; b0 += 32) {
    const int nb = n_imgs - b0 < 32 ? n_imgs - b0 : 32;
    StemImages si;
    for (int i = 0; i < nb; ++i) { si.img[i] = imgs[b0 + i]; si.h[i] = hs[b0 + i]; si.w[i] = ws[b0 + i]; }
    const dim3 grid((Wo + ST_TW - 1) / ST_TW, (Ho + ST_TH - 1) / ST_TH, nb);
    if (is_uint8)
      hipLaunchKernelGGL(stem_im2col_kernel<uint8_t>, grid, dim3(256), 0, (hipStream_t)stream, si, mean, stdv, (bf16_t*)col,
                         b0, Ho, Wo, KP);
    else
      hipLaunchKernelGGL(stem_im2col_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, si, mean, stdv, (bf16_t*)col, b0,
                         Ho, Wo, KP);
    U2_CHECK_LAUNCH();
  }
  return 0;
}

// Gradient of p6 = p5[:, ::2, ::2, :] (LastLevelMaxPool = max_pool2d(kernel 1, stride 2), backbone/fpn.py:188-200): dx[b][y][x][:] =
// g[b][y/2][x/2][:] at even (y, x), zero elsewhere - one pass (autograd's chain for the two slices: two zero fills and two strided
// copies, 0.12 ms for an 8.6 MB map).
__global__ __launch_bounds__(256) void subsample2_bwd_kernel(const bf16_t* __restrict__ g, bf16_t* __restrict__ dx, int B, int H,
                                                             int W, int C) {
  const int cpr = C >> 3, Hs = (H + 1) >> 1, Ws = (W + 1) >> 1;
  const size_t total = (size_t)B * H * W * cpr;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % cpr);
    size_t t = i / cpr;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H), b = (int)(t / H);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (!((x | y) & 1)) v = *reinterpret_cast<const uint4*>(g + (((size_t)b * Hs + (y >> 1)) * Ws + (x >> 1)) * C + ch * 8);
    *reinterpret_cast<uint4*>(dx + i * 8) = v;
  }
}

extern "C" int u2_subsample2_bwd(const void* g, void* dx, int B, int H, int W, int C, void* stream) {
  if ((C & 7) || B < 0 || H < 1 || W < 1) return -1;
  const size_t total = (size_t)B * H * W * (C >> 3);
  if (!total) return 0;
  size_t grid = (total + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(subsample2_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g, (bf16_t*)dx, B, H,
                     W, C);
  U2_CHECK_LAUNCH();
  return 0;
}

// ImageList.from_tensors(gt_sem_seg, size_divisibility, ignore_value) (structures/image_list.py:70-122 as called by
// meta_arch/panoptic_fpn.py:118-126) for the label maps of a batch in one launch: out[b][y][x] (uint8) = labels_b[y][x] inside the
// image, `pad` outside.  Labels arrive as int64 (the dataset mapper's dtype) or uint8; per image it was a conversion and a strided
// copy, 32 launches + the fill per 16-image step.  One thread = 16 output bytes (Wpad % 16 == 0).
struct LabelImages { const void* img[32]; int h[32]; int w[32]; };
template <typename T>
__global__ __launch_bounds__(256) void label_pad_kernel(const LabelImages imgs, uint8_t* __restrict__ out, int b0, int Hpad, int Wpad,
                                                        int pad) {
  const int bi = blockIdx.y;
  const T* __restrict__ src = reinterpret_cast<const T*>(imgs.img[bi]);
  const int h = imgs.h[bi], w = imgs.w[bi];
  const int wq = Wpad >> 4;
  const long long total = (long long)Hpad * wq;
  uint8_t* o = out + (size_t)(b0 + bi) * Hpad * Wpad;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int y = (int)(i / wq), x0 = (int)(i - (long long)y * wq) * 16;
    __attribute__((aligned(16))) uint8_t v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = (y < h && x0 + e < w) ? (uint8_t)src[(size_t)y * w + x0 + e] : (uint8_t)pad;
    *reinterpret_cast<uint4*>(o + (size_t)y * Wpad + x0) = *reinterpret_cast<const uint4*>(v);
  }
}

extern "C" int u2_label_pad_batch(const void* const* imgs, const int* hs, const int* ws, int n_imgs, int is_int64, void* out,
                                  int Hpad, int Wpad, int pad, void* stream) {
  if (Hpad < 1 || Wpad < 16 || (Wpad & 15) || pad < 0 || pad > 255 || ((uintptr_t)out & 15)) return -1;
  for (int b0 = 0; b0 < n_imgs; b0 += 32) {
    const int nb = n_imgs - b0 < 32 ? n_imgs - b0 : 32;
    LabelImages li;
    for (int i = 0; i < nb; ++i) {
      li.img[i] = imgs[b0 + i]; li.h[i] = hs[b0 + i]; li.w[i] = ws[b0 + i];
      if (li.h[i] < 0 || li.w[i] < 0 || li.h[i] > Hpad || li.w[i] > Wpad) return -1;
    }
    long long gx = ((long long)Hpad * (Wpad >> 4) + 255) / 256;
    if (gx > 128) gx = 128;
    const dim3 grid((unsigned)gx, nb);
    if (is_int64)
      hipLaunchKernelGGL(label_pad_kernel<long long>, grid, dim3(256), 0, (hipStream_t)stream, li, (uint8_t*)out, b0, Hpad, Wpad, pad);
    else
      hipLaunchKernelGGL(label_pad_kernel<uint8_t>, grid, dim3(256), 0, (hipStream_t)stream, li, (uint8_t*)out, b0, Hpad, Wpad, pad);
    U2_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int u2_stem_im2col(const void* img, int is_uint8, const float* mean, const float* stdv, void* col, int b, int h,
                              int w, int Hpad, int Wpad, int KP, void* stream) {
  // single image written to batch slot b: shift the column base instead of the batch index
  const int Ho = (Hpad + 6 - 7) / 2 + 1, Wo = (Wpad + 6 - 7) / 2 + 1;
  const void* imgs[1] = {img};
  return u2_stem_im2col_batch(imgs, &h, &w, 1, is_uint8, mean, stdv, (unsigned short*)col + (size_t)b * Ho * Wo * KP, Hpad, Wpad,
                              KP, stream);
}
