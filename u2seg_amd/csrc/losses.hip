// Loss kernels (fp32 arithmetic on bf16 logits; forward and the logit gradient produced in one pass).
//
// Replaces (reference file:line):
//   SemSegFPNHead.losses: bilinear x4 + F.cross_entropy(mean, ignore 255)   meta_arch/semantic_seg.py:255-267
//   FastRCNNOutputLayers.losses: cross_entropy(mean) + box_reg_loss (L1)     roi_heads/fast_rcnn.py:307-347,424-463
//   mask_rcnn_loss: gt-class channel gather + BCE-with-logits(mean)          roi_heads/mask_head.py:33-112
//     (fused with the 1x1 predictor conv of mask_head.py:258 so only the gt-class channel is computed)
//   RPN.losses: BCE-with-logits(sum) + L1 over positives, / (batch_per_image * N)   proposal_generator/rpn.py:366-429
//   Box2BoxTransform.get_deltas                                               modeling/box_regression.py:43-76
#include "common.h"
#include "u2seg_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Semantic head loss: logits [B][h][w][LP] bf16 at stride 4, targets uint8 [B][4h][4w].
//
// "Owner computes": a work-group owns a 15 x 7 block of low-resolution logits and evaluates every full-resolution
// pixel that touches them (64 x 32 pixels: 60 x 28 of its own plus a halo its neighbours evaluate too, 1.22x the
// minimum work).  The logit gradient of the owned block is accumulated in LDS and written once with plain stores:
// no global atomics, no pre-zeroed accumulator.  A pixel's loss is counted by the group owning its upper-left tap.
//
// Work split: with scale 4 and align_corners=False the 4 x 4 pixels x = 4g+2..4g+5, y = 4k+2..4k+5 share one 2 x 2
// set of taps (g, g+1) x (k, k+1).  Four adjacent lanes take such a cell, 8 classes each (softmax max / sum cross the
// four lanes with two DPP shuffles): the 4 taps x 8 classes are read from LDS once, the 16 pixels run out of
// registers, and the cell's gradient is merged in registers into 4 taps x 8 values before it touches LDS.
// ------------------------------------------------------------------------------------------------
constexpr int SS_OW = 15, SS_OH = 7;                 // owned taps per work-group
constexpr int SS_LW = SS_OW + 2, SS_LH = SS_OH + 2;  // staged taps (one halo ring)
constexpr int SS_MAXC = 32, SS_PITCH = 36;           // 36 floats: 16-byte aligned rows, tap columns on distinct banks
constexpr int SS_ZPITCH = 40;                        // the staged logits stay bf16 (round 5): 80-byte rows, 12 KB instead of 22 KB -
                                                     // four work-groups per CU instead of three

// reductions over the 4 lanes of a quad: DPP quad_perm [1,0,3,2] (0xB1) and [2,3,0,1] (0x4E)
template <int CTRL>
__device__ __forceinline__ float quad_swap(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, quad_swap<0xB1>(v));
  return fmaxf(v, quad_swap<0x4E>(v));
}
__device__ __forceinline__ float quad_sum(float v) {
  v += quad_swap<0xB1>(v);
  return v + quad_swap<0x4E>(v);
}

__global__ __launch_bounds__(256, 4) void semseg_ce_kernel(const bf16_t* __restrict__ logits, const uint8_t* __restrict__ target,
                                                        float* __restrict__ grad_acc, float* __restrict__ loss_sum,
                                                        float* __restrict__ valid_cnt, int B, int h, int w, int LP, int NC,
                                                        int ignore) {
  __shared__ __attribute__((aligned(16))) bf16_t zt[SS_LH * SS_LW * SS_ZPITCH];
  __shared__ __attribute__((aligned(16))) float gt[SS_LH * SS_LW * SS_PITCH];
  __shared__ __attribute__((aligned(16))) uint8_t lab[32 * 64];  // the tile's labels; `ignore` outside the image
  __shared__ float red[4];
  const int H = 4 * h, W = 4 * w;
  const int b = blockIdx.z;
  const int ox0 = (int)blockIdx.x * SS_OW, oy0 = (int)blockIdx.y * SS_OH;  // first owned tap
  const int lx0 = ox0 - 1, ly0 = oy0 - 1;                                  // first staged tap
  const int tid = threadIdx.x;
  {  // labels: 32 rows x 64 pixels starting at (4*oy0 - 2, 4*ox0 - 2); rows are 2-byte aligned in memory
    const int px0 = 4 * ox0 - 2, py0 = 4 * oy0 - 2;
    for (int i = tid; i < 32 * 32; i += 256) {
      const int r = i >> 5, cpair = (i & 31) * 2;
      const int y = py0 + r, x = px0 + cpair;
      uint8_t l0 = (uint8_t)ignore, l1 = (uint8_t)ignore;
      if (y >= 0 && y < H && x >= 0 && x + 1 < W) {
        const unsigned short two = *reinterpret_cast<const unsigned short*>(target + ((size_t)b * H + y) * W + x);
        l0 = (uint8_t)(two & 0xFF);
        l1 = (uint8_t)(two >> 8);
      }
      lab[r * 64 + cpair] = l0;
      lab[r * 64 + cpair + 1] = l1;
    }
  }
  for (int i = tid; i < SS_LH * SS_LW * (SS_MAXC / 8); i += 256) {  // logits: 16-byte chunks -> fp32, zero past NC
    const int ch = i % (SS_MAXC / 8);
    const int t = i / (SS_MAXC / 8);
    const int tx = t % SS_LW, ty = t / SS_LW;
    const int lx = min(max(lx0 + tx, 0), w - 1), ly = min(max(ly0 + ty, 0), h - 1);
    bf16_t v[8];
    if (ch * 8 < LP) {
      *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(logits + (((size_t)b * h + ly) * w + lx) * LP + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (ch * 8 + e < NC) ? v[e] : (bf16_t)0;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0;
    }
    float* gd = gt + t * SS_PITCH + ch * 8;
    *reinterpret_cast<uint4*>(zt + t * SS_ZPITCH + ch * 8) = *reinterpret_cast<const uint4*>(v);
    *reinterpret_cast<float4*>(gd) = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(gd + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();

  const int cj = tid & 3;          // class group: classes 8cj .. 8cj+7
  const int qx = (tid >> 2) & 15;  // cell column inside the tile
  const int wv = tid >> 6;
  const int gx = lx0 + qx;         // cell column: pixels 4gx+2 .. 4gx+5, taps gx and gx+1
  bool cvalid[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) cvalid[e] = cj * 8 + e < NC;
  float my_loss = 0.f, my_cnt = 0.f;

#pragma unroll 1
  for (int cy = wv; cy < SS_OH + 1; cy += 4) {  // cell row: pixels 4gy+2 .. 4gy+5, taps gy and gy+1
    const int gy = ly0 + cy;
    if (gx >= w || gy >= h) continue;
    // tap coordinates of the cell (clamped exactly as the per-pixel formulas below would produce them)
    const int x0 = max(gx, 0), y0 = max(gy, 0);
    const int x1 = x0 + (x0 < w - 1 ? 1 : 0), y1 = y0 + (y0 < h - 1 ? 1 : 0);
    const int t00 = (y0 - ly0) * SS_LW + (x0 - lx0), t01 = (y0 - ly0) * SS_LW + (x1 - lx0);
    const int t10 = (y1 - ly0) * SS_LW + (x0 - lx0), t11 = (y1 - ly0) * SS_LW + (x1 - lx0);
    const int i00 = t00 * SS_PITCH + cj * 8, i01 = t01 * SS_PITCH + cj * 8;   // the taps' rows of the gradient tile
    const int i10 = t10 * SS_PITCH + cj * 8, i11 = t11 * SS_PITCH + cj * 8;
    // the four taps stay packed (16 registers instead of 32: the kernel is compiled for four waves per SIMD) and are widened
    // where a pixel column uses them
    bf16_t h00[8], h01[8], h10[8], h11[8];
    *reinterpret_cast<uint4*>(h00) = *reinterpret_cast<const uint4*>(zt + t00 * SS_ZPITCH + cj * 8);
    *reinterpret_cast<uint4*>(h01) = *reinterpret_cast<const uint4*>(zt + t01 * SS_ZPITCH + cj * 8);
    *reinterpret_cast<uint4*>(h10) = *reinterpret_cast<const uint4*>(zt + t10 * SS_ZPITCH + cj * 8);
    *reinterpret_cast<uint4*>(h11) = *reinterpret_cast<const uint4*>(zt + t11 * SS_ZPITCH + cj * 8);
    const bool own_cell = y0 >= oy0 && y0 < oy0 + SS_OH && x0 >= ox0 && x0 < ox0 + SS_OW;
    unsigned labrow[4];  // the cell's 4 x 4 labels: row r = bytes of labrow[r]
#pragma unroll
    for (int r = 0; r < 4; ++r) labrow[r] = *reinterpret_cast<const unsigned*>(lab + (cy * 4 + r) * 64 + qx * 4);
    float a00[8], a01[8], a10[8], a11[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a00[e] = 0.f; a01[e] = 0.f; a10[e] = 0.f; a11[e] = 0.f; }
    bool any = false;
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
      const int x = 4 * gx + 2 + i;
      const float sx = fmaxf((x + 0.5f) * 0.25f - 0.5f, 0.f);
      const float lx = sx - (float)(int)sx, hx = 1.f - lx;
      float p[8], q[8], u0[8], u1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // same association as ATen's upsample_bilinear2d: hy*(hx*v00 + lx*v01) + ly*(hx*v10 + lx*v11)
        p[e] = hx * bf2f(h00[e]) + lx * bf2f(h01[e]);
        q[e] = hx * bf2f(h10[e]) + lx * bf2f(h11[e]);
        u0[e] = 0.f;
        u1[e] = 0.f;
      }
#pragma unroll 1
      for (int r = 0; r < 4; ++r) {
        const int y = 4 * gy + 2 + r;
        const int t = (labrow[r] >> (8 * i)) & 0xFF;
        if (t == ignore) continue;  // also pixels outside the image; uniform over the four class lanes of the pixel
        any = true;
        const float sy = fmaxf((y + 0.5f) * 0.25f - 0.5f, 0.f);
        const float ly = sy - (float)(int)sy, hy = 1.f - ly;
        float z[8];
        float mx = -INFINITY, z_t = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          z[e] = cvalid[e] ? hy * p[e] + ly * q[e] : -INFINITY;
          mx = fmaxf(mx, z[e]);
          if (cj * 8 + e == t) z_t = z[e];
        }
        mx = quad_max(mx);
        float se = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { z[e] = __expf(z[e] - mx); se += z[e]; }  // exp(-inf) = 0 for the padding classes
        se = quad_sum(se);
        if (own_cell) {  // the lane holding class t subtracts z_t, lane 0 of the pixel adds the log-sum-exp
          my_loss -= z_t;
          if (cj == 0) { my_loss += mx + __logf(se); my_cnt += 1.f; }
        }
        const float inv = 1.f / se;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float g = z[e] * inv - (cj * 8 + e == t ? 1.f : 0.f);
          u0[e] += hy * g;
          u1[e] += ly * g;
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        a00[e] += hx * u0[e];
        a01[e] += lx * u0[e];
        a10[e] += hx * u1[e];
        a11[e] += lx * u1[e];
      }
    }
    if (any) {
      // contributions to taps outside the owned block land in the (discarded) halo ring of gt
      const int o00 = i00, o01 = i01, o10 = i10, o11 = i11;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (cvalid[e]) {
          atomicAdd(&gt[o00 + e], a00[e]);
          atomicAdd(&gt[o01 + e], a01[e]);
          atomicAdd(&gt[o10 + e], a10[e]);
          atomicAdd(&gt[o11 + e], a11[e]);
        }
      }
    }
  }
  const float bl = block_sum_256(my_loss, red);
  const float bc = block_sum_256(my_cnt, red);
  if (tid == 0 && bc > 0.f) { atomicAdd(loss_sum, bl); atomicAdd(valid_cnt, bc); }
  __syncthreads();
  // every logit of the owned block is written exactly once (channels NC..LP-1 as zeros)
  for (int i = tid; i < SS_OH * SS_OW * LP; i += 256) {
    const int c = i % LP;
    const int t = i / LP;
    const int ox = t % SS_OW, oy = t / SS_OW;
    const int lx = ox0 + ox, ly = oy0 + oy;
    if (lx >= w || ly >= h) continue;
    const float g = c < NC ? gt[((oy + 1) * SS_LW + (ox + 1)) * SS_PITCH + c] : 0.f;
    grad_acc[(((size_t)b * h + ly) * w + lx) * LP + c] = g;
  }
}

// dlogits(bf16) = grad_acc * (*gscale) / (*valid_cnt)
__global__ __launch_bounds__(256) void scale_to_bf16_kernel(const float* __restrict__ acc, const float* __restrict__ num,
                                                            const float* __restrict__ den, float mult,
                                                            bf16_t* __restrict__ out, size_t n) {
  const float s = mult * (num ? *num : 1.f) / (den ? *den : 1.f);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = f2bf(acc[i] * s);
}

// ------------------------------------------------------------------------------------------------
// Row softmax cross-entropy: logits [R][LP] bf16 (NC valid columns), labels int64; one wave per row.
// loss_sum += sum_r (lse - z[label]);  dlogits[r][c] = (softmax - onehot) * gscale  (pad columns = 0)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_ce_kernel(const bf16_t* __restrict__ logits, const long long* __restrict__ labels,
                                                         bf16_t* __restrict__ dlogits, float* __restrict__ loss_sum, int R,
                                                         int NC, int LP, float gscale) {
  const int lane = threadIdx.x & 63;
  if (LP <= 1024 && (LP & 7) == 0) {
    // a wave walks rows with the grid's stride and sends ONE atomic at the end: 8192 same-address atomics (one per row) were
    // the whole 0.11 ms of this kernel
    float loss_acc = 0.f;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < R; row += gridDim.x * 4) {
    const bf16_t* zr = logits + (size_t)row * LP;
    // the whole row in registers (<= 2 x 16 bytes per lane): one read of the logits instead of three passes of 2-byte loads
    const int chunks = LP >> 3;
    float v[2][8];
    float mx = -INFINITY;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int ch = it * 64 + lane;
      bf16_t raw[8];
      if (ch < chunks) *reinterpret_cast<uint4*>(raw) = *reinterpret_cast<const uint4*>(zr + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[it][e] = (ch < chunks && ch * 8 + e < NC) ? bf2f(raw[e]) : -INFINITY;
        mx = fmaxf(mx, v[it][e]);
      }
    }
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[it][e] = __expf(v[it][e] - mx);  // exp(-inf) = 0 for the pad columns
        se += v[it][e];
      }
    se = wave_sum(se);
    const int t = (int)labels[row];
    const float inv = 1.f / se;
    bf16_t* dr = dlogits + (size_t)row * LP;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int ch = it * 64 + lane;
      if (ch < chunks) {
        bf16_t ov[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = ch * 8 + e;
          ov[e] = f2bf(c < NC ? (v[it][e] * inv - (c == t ? 1.f : 0.f)) * gscale : 0.f);
        }
        *reinterpret_cast<uint4*>(dr + ch * 8) = *reinterpret_cast<const uint4*>(ov);
      }
    }
    loss_acc += mx + __logf(se) - bf2f(zr[t]);
    }
    if (lane == 0 && loss_acc != 0.f) atomicAdd(loss_sum, loss_acc);
    return;
  }
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const bf16_t* zr = logits + (size_t)row * LP;
  float mx = -INFINITY;
  for (int c = lane; c < NC; c += 64) mx = fmaxf(mx, bf2f(zr[c]));
  mx = wave_max(mx);
  float se = 0.f;
  for (int c = lane; c < NC; c += 64) se += __expf(bf2f(zr[c]) - mx);
  se = wave_sum(se);
  const int t = (int)labels[row];
  const float inv = 1.f / se;
  bf16_t* dr = dlogits + (size_t)row * LP;
  for (int c = lane; c < LP; c += 64) {
    float g = 0.f;
    if (c < NC) g = (__expf(bf2f(zr[c]) - mx) * inv - (c == t ? 1.f : 0.f)) * gscale;
    dr[c] = f2bf(g);
  }
  if (lane == 0) atomicAdd(loss_sum, mx + __logf(se) - bf2f(zr[t]));
}

// ------------------------------------------------------------------------------------------------
// Mask head: logit[n][p] = x[n][p][:] . Wp[cls_n][:] + bp[cls_n] (rounded to bf16 like the autocast conv),
// loss_sum += BCEwithLogits(logit, target);  dx = dlogit * Wp[cls];  dWp[cls] += sum_p dlogit x;  dbp[cls] += sum dlogit
// x: [N][P][C] bf16 (C = 256), target uint8 [N][P].  One workgroup per ROI, one wave per position.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_predict_bce_kernel(const bf16_t* __restrict__ x, const float* __restrict__ Wp,
                                                               const float* __restrict__ bp, const long long* __restrict__ cls,
                                                               const uint8_t* __restrict__ target, bf16_t* __restrict__ dx,
                                                               float* __restrict__ dWp, float* __restrict__ dbp,
                                                               float* __restrict__ loss_sum, bf16_t* __restrict__ logit_out,
                                                               int P, int C, float gscale, int phased_side,
                                                               const float* __restrict__ gmul) {
  __shared__ float red[4][260];
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int k = (int)cls[n];
  // C == 256: 4 channels per lane; weights are rounded to bf16 like the autocast conv operand
  float wq[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) wq[e] = bf2f(f2bf(Wp[(size_t)k * C + lane * 4 + e]));
  const float bias = bf2f(f2bf(bp[k]));
  if (gmul) gscale *= *gmul;   // the upstream gradient of the loss (a device scalar): the backward launch
  float dw[4] = {0.f, 0.f, 0.f, 0.f};
  float db = 0.f, ls = 0.f;
  // round 6: blockIdx.y = a slice of the ROI's positions.  With one work-group per ROI a launch was 260 work-groups whose waves
  // walked 196 positions each, one dependent load -> wave sum -> store chain at a time: 0.18 ms for 104 MB.
  for (int p = wv + 4 * blockIdx.y; p < P; p += 4 * gridDim.y) {
    const size_t off = ((size_t)n * P + p) * C + lane * 4;
    bf16_t xv[4];
    *reinterpret_cast<uint2*>(xv) = *reinterpret_cast<const uint2*>(x + off);
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) d += bf2f(xv[e]) * wq[e];
    d = wave_sum(d);
    const float z = bf2f(f2bf(d + bias));
    // phased: position p of x / dx is (h, w, dy, dx) of the deconvolution's unshuffled output = pixel (2 h + dy, 2 w + dx) of the
    // target and of logit_out
    int o = p;
    if (phased_side) {
      const int S = phased_side >> 1, hw = p >> 2;
      const int h = hw / S, w = hw - h * S;
      o = (2 * h + ((p >> 1) & 1)) * phased_side + 2 * w + (p & 1);
    }
    const float t = (float)target[(size_t)n * P + o];
    // max(z,0) - z*t + log1p(exp(-|z|))
    ls += fmaxf(z, 0.f) - z * t + log1pf(__expf(-fabsf(z)));
    const float sg = 1.f / (1.f + __expf(-z));
    const float g = (sg - t) * gscale;
    bf16_t ov[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ov[e] = f2bf(g * wq[e]);
      dw[e] += g * bf2f(xv[e]);
    }
    if (dx) *reinterpret_cast<uint2*>(dx + off) = *reinterpret_cast<const uint2*>(ov);
    db += g;
    if (logit_out && lane == 0) logit_out[(size_t)n * P + o] = f2bf(z);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wv][lane * 4 + e] = dw[e];
  if (lane == 0) { red[wv][256] = db; red[wv][257] = ls; }
  __syncthreads();
  const int t = threadIdx.x;
  const float s = red[0][t] + red[1][t] + red[2][t] + red[3][t];
  if (dWp) atomicAdd(dWp + (size_t)k * C + t, s);
  if (t == 0) {
    if (dbp) atomicAdd(dbp + k, red[0][256] + red[1][256] + red[2][256] + red[3][256]);
    if (loss_sum) atomicAdd(loss_sum, red[0][257] + red[1][257] + red[2][257] + red[3][257]);
  }
}

// ------------------------------------------------------------------------------------------------
// Box coding helpers (fp32, same operation order as box_regression.py:43-116)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void get_deltas(const float* s, const float* t, float wx, float wy, float ww, float wh, float* d) {
  const float sw = s[2] - s[0], sh = s[3] - s[1];
  const float scx = s[0] + 0.5f * sw, scy = s[1] + 0.5f * sh;
  const float tw = t[2] - t[0], th = t[3] - t[1];
  const float tcx = t[0] + 0.5f * tw, tcy = t[1] + 0.5f * th;
  d[0] = wx * (tcx - scx) / sw;
  d[1] = wy * (tcy - scy) / sh;
  d[2] = ww * logf(tw / sw);
  d[3] = wh * logf(th / sh);
}

// RPN losses for one FPN level.  obj: [B][HW][LPo] (A valid cols), dlt: [B][HW][LPd] (4A valid cols),
// labels int8 [B][Atot] (-1 ignore / 0 / 1, already subsampled), match int32 [B][Atot], gt fp32 [B][G][4],
// anchors fp32 [HW*A][4] of this level.  Writes dobj/ddlt (bf16, same layouts, pads zero) and adds to loss[0:2].
// FUSED3: one 32-wide map holds both predictions of A = 3 anchors (columns 0-2 objectness, 3-14 deltas) and takes one gradient row
template <bool FUSED3>
__global__ __launch_bounds__(256) void rpn_loss_level_kernel(const bf16_t* __restrict__ obj, const bf16_t* __restrict__ dlt,
                                                             const int8_t* __restrict__ labels, const int* __restrict__ match,
                                                             const float* __restrict__ gt, const float* __restrict__ anchors,
                                                             bf16_t* __restrict__ dobj, bf16_t* __restrict__ ddlt,
                                                             float* __restrict__ loss, int B, int HW, int A, int LPo, int LPd,
                                                             int Atot, int lvl_off, int G, float gscale) {
  __shared__ float red[4];
  float lc = 0.f, ll = 0.f;
  const size_t total = (size_t)B * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int b = (int)(i / HW);
    const int pix = (int)(i - (size_t)b * HW);
    bf16_t go_l[8], gd_l[16];
#pragma unroll
    for (int c = 0; c < 8; ++c) go_l[c] = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) gd_l[c] = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (a >= A) break;
      const int aidx = pix * A + a;
      const int lab = labels[(size_t)b * Atot + lvl_off + aidx];
      if (lab < 0) continue;
      const float z = bf2f(obj[i * LPo + a]);
      const float t = (float)lab;
      lc += fmaxf(z, 0.f) - z * t + log1pf(__expf(-fabsf(z)));
      go_l[a] = f2bf((1.f / (1.f + __expf(-z)) - t) * gscale);
      if (lab == 1) {
        const int g = match[(size_t)b * Atot + lvl_off + aidx];
        float d[4];
        get_deltas(anchors + (size_t)aidx * 4, gt + ((size_t)b * G + g) * 4, 1.f, 1.f, 1.f, 1.f, d);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float pr = bf2f(dlt[i * LPd + a * 4 + q]);
          const float df = pr - d[q];
          ll += fabsf(df);
          gd_l[a * 4 + q] = f2bf((df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * gscale);
        }
      }
    }
    uint4* go = reinterpret_cast<uint4*>(dobj + i * LPo);
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    if (FUSED3) {
      bf16_t row[16];
#pragma unroll
      for (int c = 0; c < 3; ++c) row[c] = go_l[c];
#pragma unroll
      for (int c = 0; c < 12; ++c) row[3 + c] = gd_l[c];
      row[15] = 0;
      go[0] = *reinterpret_cast<const uint4*>(row);
      go[1] = *reinterpret_cast<const uint4*>(row + 8);
      for (int c = 2; c < LPo / 8; ++c) go[c] = zero4;
      continue;
    }
    uint4* gd = reinterpret_cast<uint4*>(ddlt + i * LPd);
    go[0] = *reinterpret_cast<const uint4*>(go_l);
    for (int c = 1; c < LPo / 8; ++c) go[c] = zero4;
    gd[0] = *reinterpret_cast<const uint4*>(gd_l);
    gd[1] = *reinterpret_cast<const uint4*>(gd_l + 8);
    for (int c = 2; c < LPd / 8; ++c) gd[c] = zero4;
  }
  const float s0 = block_sum_256(lc, red);
  const float s1 = block_sum_256(ll, red);
  if (threadIdx.x == 0) { atomicAdd(loss + 0, s0); atomicAdd(loss + 1, s1); }
}

// Box-head regression loss (class agnostic): pred [R][LP] bf16 (4 valid), proposals/gt fp32 [R][4],
// labels int64 [R]; fg = 0 <= label < bg_label.  loss += sum_fg |pred - target|; dpred = sign * gscale.
__global__ __launch_bounds__(256) void box_reg_l1_kernel(const bf16_t* __restrict__ pred, const float* __restrict__ prop,
                                                         const float* __restrict__ gtb, const long long* __restrict__ labels,
                                                         bf16_t* __restrict__ dpred, float* __restrict__ loss, int R, int LP,
                                                         int bg_label, float wx, float wy, float ww, float wh, float gscale) {
  __shared__ float red[4];
  float l = 0.f;
  for (int r = blockIdx.x * 256 + threadIdx.x; r < R; r += gridDim.x * 256) {
    bf16_t* gp = dpred + (size_t)r * LP;
    for (int c = 0; c < LP; ++c) gp[c] = 0;
    const long long lab = labels[r];
    if (lab < 0 || lab >= bg_label) continue;
    float d[4];
    get_deltas(prop + (size_t)r * 4, gtb + (size_t)r * 4, wx, wy, ww, wh, d);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float df = bf2f(pred[(size_t)r * LP + q]) - d[q];
      l += fabsf(df);
      gp[q] = f2bf((df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * gscale);
    }
  }
  const float s = block_sum_256(l, red);
  if (threadIdx.x == 0) atomicAdd(loss, s);
}

}  // namespace

extern "C" int u2_semseg_upsample_ce(const void* logits, const void* target, float* grad_acc, float* loss_sum,
                                     float* valid_cnt, int B, int h, int w, int LP, int NC, int ignore, void* stream) {
  if (NC > SS_MAXC || LP < NC || (LP & 7) || ignore < 0 || ignore > 255) return -1;
  if (B <= 0) return 0;
  const dim3 grid((w + SS_OW - 1) / SS_OW, (h + SS_OH - 1) / SS_OH, B);
  hipLaunchKernelGGL(semseg_ce_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits,
                     (const uint8_t*)target, grad_acc, loss_sum, valid_cnt, B, h, w, LP, NC, ignore);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_scale_to_bf16(const float* acc, const float* num, const float* den, float mult, void* out,
                                long long n, void* stream) {
  if (n <= 0) return 0;
  size_t g = ((size_t)n + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(scale_to_bf16_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, acc, num, den, mult,
                     (bf16_t*)out, (size_t)n);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_softmax_ce(const void* logits, const void* labels, void* dlogits, float* loss_sum, int R, int NC,
                             int LP, float gscale, void* stream) {
  if (R <= 0) return 0;
  int grid = (R + 3) / 4;
  if (LP <= 1024 && (LP & 7) == 0 && grid > 512) grid = 512;  // the register-resident form loops over rows
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits,
                     (const long long*)labels, (bf16_t*)dlogits, loss_sum, R, NC, LP, gscale);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_mask_predict_bce(const void* x, const float* Wp, const float* bp, const void* cls, const void* target,
                                   void* dx, float* dWp, float* dbp, float* loss_sum, void* logit_out, int N, int P,
                                   int C, float gscale, int phased_side, const float* gmul, void* stream) {
  if (C != 256 || (phased_side && ((phased_side & 1) || phased_side * phased_side != P))) return -1;
  if (N <= 0) return 0;
  // 8 slices: 0.184 -> 0.074 ms per launch at N = 260, P = 784; 24 slices measure 0.081 (256 more atomics per work-group)
  const int slices = P >= 512 ? 8 : (P >= 128 ? 4 : 1);
  hipLaunchKernelGGL(mask_predict_bce_kernel, dim3(N, slices), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, Wp, bp,
                     (const long long*)cls, (const uint8_t*)target, (bf16_t*)dx, dWp, dbp, loss_sum, (bf16_t*)logit_out,
                     P, C, gscale, phased_side, gmul);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_rpn_loss_level(const void* obj, const void* dlt, const void* labels, const int* match, const float* gt,
                                 const float* anchors, void* dobj, void* ddlt, float* loss, int B, int HW, int A, int LPo,
                                 int LPd, int Atot, int lvl_off, int G, float gscale, void* stream) {
  if (A > 4 || (LPo & 7)) return -1;
  const bool fused = dlt == nullptr;  // fused map: the deltas are columns 3 .. 14 of `obj`, the gradients one row of `dobj`
  if (fused) {
    if (ddlt || A != 3 || LPo < 16) return -1;
    dlt = (const bf16_t*)obj + A;
    LPd = LPo;
  } else if (!ddlt || LPd < 16 || (LPd & 7)) {
    return -1;
  }
  if (B <= 0 || HW <= 0) return 0;
  size_t g = ((size_t)B * HW + 255) / 256;
  if (g > 2048) g = 2048;
  if (fused)
    hipLaunchKernelGGL(rpn_loss_level_kernel<true>, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)obj,
                       (const bf16_t*)dlt, (const int8_t*)labels, match, gt, anchors, (bf16_t*)dobj, (bf16_t*)ddlt, loss, B,
                       HW, A, LPo, LPd, Atot, lvl_off, G, gscale);
  else
    hipLaunchKernelGGL(rpn_loss_level_kernel<false>, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)obj,
                       (const bf16_t*)dlt, (const int8_t*)labels, match, gt, anchors, (bf16_t*)dobj, (bf16_t*)ddlt, loss, B,
                       HW, A, LPo, LPd, Atot, lvl_off, G, gscale);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_box_reg_l1(const void* pred, const float* prop, const float* gtb, const void* labels, void* dpred,
                             float* loss, int R, int LP, int bg_label, float wx, float wy, float ww, float wh, float gscale,
                             void* stream) {
  if (R <= 0) return 0;
  int g = (R + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(box_reg_l1_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)pred, prop, gtb,
                     (const long long*)labels, (bf16_t*)dpred, loss, R, LP, bg_label, wx, wy, ww, wh, gscale);
  U2_CHECK_LAUNCH();
  return 0;
}
