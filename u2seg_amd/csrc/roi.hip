// ROI bookkeeping kernels: ROIAlign (fwd/bwd), IoU matching, box decoding, NMS, FPN level assignment.
// Integer outputs (match indices, labels, NMS keep lists, levels) are meant to be bit-identical to the
// CPU oracle on identical fp32 inputs, so this file must be compiled with -ffp-contract=off.
//
// Replaces (reference file:line):
//   ROIAlign -> torchvision.ops.roi_align(aligned=True, sampling_ratio=0)  detectron2/layers/roi_align.py:49-65
//      (arithmetic restated from layers/csrc/ROIAlignRotated/ROIAlignRotated_cpu.cpp:27-129,201-416 at angle 0)
//   assign_boxes_to_levels                                                  detectron2/modeling/poolers.py:23-59
//   pairwise_iou + Matcher                                                  structures/boxes.py:312-358, modeling/matcher.py:62-127
//   Box2BoxTransform.apply_deltas + Boxes.clip                              modeling/box_regression.py:78-116, structures/boxes.py:172-181
//   batched_nms (per-group NMS, suppress iff IoU > thr)                      detectron2/layers/nms.py:9-20
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>
#include "common.h"
#include "u2seg_hip.h"

namespace {

struct RoiLevels {
  const bf16_t* feat[4];
  float* gfeat[4];
  int H[4], W[4];
  float scale[4];
  int flags = 0;   // bit 0 (A/B runs: U2_ROI_FWD=bins): the forward pass takes the one-item-per-bin form of rounds 1-4
};

__device__ __forceinline__ bool bil_prep(float y, float x, int H, int W, int& yl, int& xl, int& yh, int& xh, float& w1,
                                         float& w2, float& w3, float& w4) {
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return false;
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  yl = (int)y;
  xl = (int)x;
  if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else { yh = yl + 1; }
  if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else { xh = xl + 1; }
  const float ly = y - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
  w1 = hy * hx; w2 = hy * lx; w3 = ly * hx; w4 = ly * lx;
  return true;
}

// rois: [R][5] fp32 (batch, x0, y0, x1, y1); level[R] int32 in [0,4); out [R][PH][PW][C] bf16
template <bool BWD>
__global__ __launch_bounds__(256) void roi_align_kernel(const RoiLevels lv, const float* __restrict__ rois,
                                                        const int* __restrict__ level, bf16_t* __restrict__ out,
                                                        const bf16_t* __restrict__ dout, int R, int C, int PH, int PW,
                                                        float gscale) {
  const int cpr = C >> 3;
  const size_t total = (size_t)R * PH * PW * cpr;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cc = (int)(i % cpr);
    size_t p = i / cpr;
    const int pw = (int)(p % PW); p /= PW;
    const int ph = (int)(p % PH);
    const int r = (int)(p / PH);
    const int l = level[r];
    const int H = lv.H[l], W = lv.W[l];
    const float sc = lv.scale[l];
    const float* roi = rois + (size_t)r * 5;
    const int b = (int)roi[0];
    const float sw = roi[1] * sc - 0.5f, sh = roi[2] * sc - 0.5f;
    const float ew = roi[3] * sc - 0.5f, eh = roi[4] * sc - 0.5f;
    const float rw = ew - sw, rh = eh - sh;
    const float bh = rh / (float)PH, bw = rw / (float)PW;
    const int gh = (int)ceilf(rh / (float)PH), gw = (int)ceilf(rw / (float)PW);
    const float cnt = (float)max(gh * gw, 1);
    float acc[8];
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (BWD) {
      bf16_t dv[8];
      *reinterpret_cast<uint4*>(dv) = *reinterpret_cast<const uint4*>(dout + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = bf2f(dv[e]) * gscale / cnt;
    }
    const size_t plane = (size_t)b * H * W;
    for (int iy = 0; iy < gh; ++iy) {
      const float y = sh + ph * bh + (iy + 0.5f) * bh / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        const float x = sw + pw * bw + (ix + 0.5f) * bw / (float)gw;
        int yl, xl, yh, xh;
        float w1, w2, w3, w4;
        if (!bil_prep(y, x, H, W, yl, xl, yh, xh, w1, w2, w3, w4)) continue;
        const size_t o1 = (plane + (size_t)yl * W + xl) * C + cc * 8;
        const size_t o2 = (plane + (size_t)yl * W + xh) * C + cc * 8;
        const size_t o3 = (plane + (size_t)yh * W + xl) * C + cc * 8;
        const size_t o4 = (plane + (size_t)yh * W + xh) * C + cc * 8;
        if (!BWD) {
          bf16_t v1[8], v2[8], v3[8], v4[8];
          const bf16_t* f = lv.feat[l];
          *reinterpret_cast<uint4*>(v1) = *reinterpret_cast<const uint4*>(f + o1);
          *reinterpret_cast<uint4*>(v2) = *reinterpret_cast<const uint4*>(f + o2);
          *reinterpret_cast<uint4*>(v3) = *reinterpret_cast<const uint4*>(f + o3);
          *reinterpret_cast<uint4*>(v4) = *reinterpret_cast<const uint4*>(f + o4);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            acc[e] += w1 * bf2f(v1[e]) + w2 * bf2f(v2[e]) + w3 * bf2f(v3[e]) + w4 * bf2f(v4[e]);
        } else {
          float* gf = lv.gfeat[l];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            atomicAdd(gf + o1 + e, w1 * g[e]);
            atomicAdd(gf + o2 + e, w2 * g[e]);
            atomicAdd(gf + o3 + e, w3 * g[e]);
            atomicAdd(gf + o4 + e, w4 * g[e]);
          }
        }
      }
    }
    if (!BWD) {
      bf16_t o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e] / cnt);
      *reinterpret_cast<uint4*>(out + i * 8) = *reinterpret_cast<const uint4*>(o);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ROIAlign backward as a per-pixel gather (no atomics, deterministic): every feature pixel of level l of
// image b visits the ROIs assigned to (b, l) (list `order`, segment offsets `seg`), and for those whose
// sampling footprint touches the pixel accumulates  sum_{samples} w_y * w_x * dout[r][ph][pw][:] / count.
// One workgroup = an 8x8 pixel tile; ROIs overlapping the tile are compacted into LDS first.
// ---------------------------------------------------------------------------------------------
struct RoiGeom { float sh, sw, bh, bw; int gh, gw, r; float inv_cnt; int py0, py1, px0, px1; };

// weight with which sample position `pos` (one axis, extent n) feeds pixel `p`; 0 if the sample is skipped
__device__ __forceinline__ float axis_weight(float pos, int n, int p) {
  if (pos < -1.0f || pos > (float)n) return 0.f;
  if (pos <= 0.f) pos = 0.f;
  int lo = (int)pos, hi;
  if (lo >= n - 1) { hi = lo = n - 1; pos = (float)lo; } else { hi = lo + 1; }
  const float l = pos - (float)lo, h = 1.f - l;
  float w = 0.f;
  if (lo == p) w += h;
  if (hi == p) w += l;
  return w;
}

// ---------------------------------------------------------------------------------------------
// ROIAlign forward in the same separable form: out[ph][pw] = (1/count) * sum_y sum_x Wy[ph][y] * Wx[pw][x] * feat[y][x]
// with Wy[ph][y] = sum over the bin row's samples of their bilinear weight on pixel row y.  One work-group per ROI:
// the per-axis tables (<= 16 pixels per bin) are built once in LDS, then every (bin, 8-channel chunk) item reads each
// feature pixel its bin touches exactly once (~25-36 reads) instead of 4 per sample (64 at 4 x 4 samples per bin).
// ---------------------------------------------------------------------------------------------
constexpr int FS_MAXP = 14, FS_MAXR = 16;
// Waves per SIMD the forward kernel is compiled for.  The pass is bound by the latency of one dependent 16-byte load per thread
// times the number of threads resident (71-75 % of the wave cycles parked, 3.4 TB/s through the L1s): at 92 registers five
// waves fit a SIMD, at <= 64 eight - measured 1.48 -> 1.23 ms for 32 000 ROIs of 7 x 7, 0.44 -> 0.37 for 3 200 of 14 x 14
// (profiles/r05_pmc_roi.txt; the handful of spilled registers sit in the sample-by-sample fall-back)
#ifndef FS_OCC
#define FS_OCC 8
#endif
#ifndef FS_PAIR
#define FS_PAIR 1
#endif
#ifndef FS_QUAD
#define FS_QUAD 1
#endif

// One work item = (bin row ph, 8-channel chunk) sweeping ALL bin columns of the row (round 5).  Neighbouring bins overlap by one or
// two pixel columns (bin pw ends at floor(last sample) + 1, bin pw + 1 starts at floor(its first sample) >= floor(that last sample)),
// and with one item per bin those columns were fetched once per bin: (roi width + 2 P) loads per pixel row where (roi width + 2) do,
// 1.4-1.75x the 16-byte L1 / L2 requests for ROIs of 14-28 pixels - and the pass is bound by exactly those requests (19 TB/s of
// them at inference, SQ counters: 71 % of the wave cycles parked on memory, profiles/r05_pmc_roi.txt).  The item keeps the last
// two pixels it loaded; every bin still adds its terms in (ky, kx) order with the same products, so the result is bit-identical
// to the per-bin form.
template <int PP>
__device__ __forceinline__ void roi_fwd_bin_rows(const bf16_t* __restrict__ f, size_t plane, int W, int C, int cpr,
                                                 const float (*wtab)[FS_MAXP][FS_MAXR], const int (*lo)[FS_MAXP],
                                                 const int (*cnt)[FS_MAXP], float inv_cnt, bf16_t* __restrict__ out, int r, int tid) {
  for (int it = tid; it < PP * cpr; it += 256) {
    const int cc = it % cpr, ph = it / cpr;
    float acc[PP][8];
#pragma unroll
    for (int pw = 0; pw < PP; ++pw)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[pw][e] = 0.f;
    const int ny = cnt[0][ph], y0 = lo[0][ph];
    for (int ky = 0; ky < ny; ++ky) {
      const float a = wtab[0][ph][ky];
      if (a == 0.f) continue;
      const bf16_t* rowp = f + (plane + (size_t)(y0 + ky) * W) * C + cc * 8;
      int cx0 = -1, cx1 = -1;
      uint4 cv0 = make_uint4(0, 0, 0, 0), cv1 = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int pw = 0; pw < PP; ++pw) {
        const int nx = cnt[1][pw], x0 = lo[1][pw];
        for (int kx = 0; kx < nx; ++kx) {
          const float wgt = a * wtab[1][pw][kx];
          if (wgt == 0.f) continue;
          const int x = x0 + kx;
          uint4 u;
          if (x == cx1) {
            u = cv1;
          } else if (x == cx0) {
            u = cv0;
          } else {
            u = *reinterpret_cast<const uint4*>(rowp + (size_t)x * C);
            cv0 = cv1; cx0 = cx1;
            cv1 = u; cx1 = x;
          }
          const unsigned wd[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[pw][2 * e] += wgt * __uint_as_float(wd[e] << 16);
            acc[pw][2 * e + 1] += wgt * __uint_as_float(wd[e] & 0xffff0000u);
          }
        }
      }
    }
#pragma unroll
    for (int pw = 0; pw < PP; ++pw) {
      bf16_t o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[pw][e] * inv_cnt);
      *reinterpret_cast<uint4*>(out + (((size_t)r * PP + ph) * PP + pw) * C + cc * 8) = *reinterpret_cast<const uint4*>(o);
    }
  }
}

// PT: 7 / 14 = the bin-row form for that pooled size (its own instantiation: 56 / 112 accumulator registers), 0 = one item per bin
template <int PT>
__global__ __launch_bounds__(256, FS_OCC) void roi_align_fwd_sep_kernel(const RoiLevels lv, const float* __restrict__ rois,
                                                                const int* __restrict__ level, const int* __restrict__ order,
                                                                bf16_t* __restrict__ out, int C, int P) {
  __shared__ float wtab[2][FS_MAXP][FS_MAXR];
  __shared__ int lo[2][FS_MAXP], cnt[2][FS_MAXP];
  __shared__ int s_fallback;
  // processing order: XCD x (work-groups x, x + 8, ...) takes a contiguous part of the (sorted) list
  const int r = order ? order[xcd_remap((int)blockIdx.x, (int)gridDim.x)] : (int)blockIdx.x;
  const int tid = threadIdx.x;
  const int l = level[r];
  const int H = lv.H[l], W = lv.W[l];
  const float sc = lv.scale[l];
  const float* roi = rois + (size_t)r * 5;
  const int b = (int)roi[0];
  const float sw = roi[1] * sc - 0.5f, sh = roi[2] * sc - 0.5f;
  const float ew = roi[3] * sc - 0.5f, eh = roi[4] * sc - 0.5f;
  const float rw = ew - sw, rh = eh - sh;
  const float bh = rh / (float)P, bw = rw / (float)P;
  const int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
  const float inv_cnt = 1.f / (float)max(gh * gw, 1);
  if (tid == 0) s_fallback = 0;
  __syncthreads();
  if (tid < 2 * P) {
    const int axis = tid / P, bin = tid - axis * P;
    const int n = axis ? W : H, g = axis ? gw : gh;
    const float s0 = axis ? sw : sh, bs = axis ? bw : bh;
    int a = 0, c = 0;
    if (g > 0) {
      const float first = s0 + bin * bs + 0.5f * bs / (float)g;
      const float last = s0 + bin * bs + ((float)g - 0.5f) * bs / (float)g;
      a = min(max((int)floorf(first), 0), n - 1);
      const int z = min(max((int)floorf(last) + 1, 0), n - 1);
      c = z - a + 1;
      if (c > FS_MAXR) atomicOr(&s_fallback, 1);
    }
    lo[axis][bin] = a;
    cnt[axis][bin] = c;
  }
  __syncthreads();
  const int cpr = C >> 3;
  const size_t plane = (size_t)b * H * W;
  const bf16_t* f = lv.feat[l];
  if (s_fallback) {
    // a bin wider than the table (never with FPN level assignment, kept for generality): sample by sample
    for (int it = tid; it < P * P * cpr; it += 256) {
      const int cc = it % cpr;
      const int pw = (it / cpr) % P, ph = it / (cpr * P);
      float acc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = 0.f;
      for (int iy = 0; iy < gh; ++iy) {
        const float y = sh + ph * bh + (iy + 0.5f) * bh / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
          const float x = sw + pw * bw + (ix + 0.5f) * bw / (float)gw;
          int yl, xl, yh, xh;
          float w1, w2, w3, w4;
          if (!bil_prep(y, x, H, W, yl, xl, yh, xh, w1, w2, w3, w4)) continue;
          const int ys[4] = {yl, yl, yh, yh}, xs[4] = {xl, xh, xl, xh};
          const float ws[4] = {w1, w2, w3, w4};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            bf16_t v[8];
            *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(f + (plane + (size_t)ys[t] * W + xs[t]) * C + cc * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += ws[t] * bf2f(v[e]);
          }
        }
      }
      bf16_t o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e] * inv_cnt);
      *reinterpret_cast<uint4*>(out + (((size_t)r * P + ph) * P + pw) * C + cc * 8) = *reinterpret_cast<const uint4*>(o);
    }
    return;
  }
  for (int t = tid; t < 2 * P * FS_MAXR; t += 256) {
    const int k = t % FS_MAXR;
    const int bin = (t / FS_MAXR) % P, axis = t / (FS_MAXR * P);
    float sum = 0.f;
    if (k < cnt[axis][bin]) {
      const int n = axis ? W : H, g = axis ? gw : gh;
      const float s0 = axis ? sw : sh, bs = axis ? bw : bh;
      const float step = bs / (float)g;
      const int pix = lo[axis][bin] + k;
      for (int i = 0; i < g; ++i) sum += axis_weight(s0 + bin * bs + (i + 0.5f) * step, n, pix);
    }
    wtab[axis][bin][k] = sum;
  }
  __syncthreads();
  if constexpr (PT > 0) {
    roi_fwd_bin_rows<PT>(f, plane, W, C, cpr, wtab, lo, cnt, inv_cnt, out, r, tid);
    return;
  }
  // (A rows-then-columns form - T[x][c] = sum_y Wy[ph][y] feat[y][x][c] per bin row staged in LDS as fp32, then the column sums -
  // reads every pixel of a bin row once, 22 k instead of 39 k 16-byte loads per 7 x 7 ROI, but its 40 KB stage leaves three
  // work-groups per CU and two barriers per bin row: measured equal to this one-pass form in training (1.36 ms per step) and at
  // inference (4.9 vs 5.0 ms per 32-image batch), so the simpler form stays.)
  for (int it = tid; it < P * P * cpr; it += 256) {
    const int cc = it % cpr;
    const int pw = (it / cpr) % P, ph = it / (cpr * P);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const int ny = cnt[0][ph], nx = cnt[1][pw], y0 = lo[0][ph], x0 = lo[1][pw];
#if FS_QUAD
    // round 6: two pixel rows x two pixel columns of the bin in flight per thread (four independent 16-byte loads per round trip
    // instead of two: the pass is bound by the latency of its dependent loads times the threads resident).  A missing second row /
    // column re-reads the first one with weight 0; the sum's terms are the same, their order within a 2 x 2 block is not.
    for (int ky = 0; ky < ny; ky += 2) {
      const bool r1 = ky + 1 < ny;
      const float a0 = wtab[0][ph][ky], a1 = r1 ? wtab[0][ph][ky + 1] : 0.f;
      if (a0 == 0.f && a1 == 0.f) continue;
      const bf16_t* row0 = f + (plane + (size_t)(y0 + ky) * W + x0) * C + cc * 8;
      const bf16_t* row1 = r1 ? row0 + (size_t)W * C : row0;
      for (int kx = 0; kx < nx; kx += 2) {
        const bool c1 = kx + 1 < nx;
        const float b0 = wtab[1][pw][kx], b1 = c1 ? wtab[1][pw][kx + 1] : 0.f;
        const size_t o0 = (size_t)kx * C, o1 = c1 ? o0 + C : o0;
        bf16_t v00[8], v01[8], v10[8], v11[8];
        *reinterpret_cast<uint4*>(v00) = *reinterpret_cast<const uint4*>(row0 + o0);
        *reinterpret_cast<uint4*>(v01) = *reinterpret_cast<const uint4*>(row0 + o1);
        *reinterpret_cast<uint4*>(v10) = *reinterpret_cast<const uint4*>(row1 + o0);
        *reinterpret_cast<uint4*>(v11) = *reinterpret_cast<const uint4*>(row1 + o1);
        const float w00 = a0 * b0, w01 = a0 * b1, w10 = a1 * b0, w11 = a1 * b1;
        if (w00 != 0.f) {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += w00 * bf2f(v00[e]);
        }
        if (w01 != 0.f) {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += w01 * bf2f(v01[e]);
        }
        if (w10 != 0.f) {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += w10 * bf2f(v10[e]);
        }
        if (w11 != 0.f) {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += w11 * bf2f(v11[e]);
        }
      }
    }
#else
    for (int ky = 0; ky < ny; ++ky) {
      const float a = wtab[0][ph][ky];
      if (a == 0.f) continue;
      const bf16_t* rowp = f + (plane + (size_t)(y0 + ky) * W + x0) * C + cc * 8;
      int kx = 0;
#if FS_PAIR
      // two pixels of the row in flight per thread (the pass is latency-bound: profiles/r05_pmc_roi.txt); both loads are issued
      // whatever their weights (every tabulated pixel lies inside the map), the terms are still added in kx order and a zero
      // weight still adds nothing, so the result keeps its bits
      for (; kx + 1 < nx; kx += 2) {
        const float w0 = a * wtab[1][pw][kx], w1 = a * wtab[1][pw][kx + 1];
        bf16_t v0[8], v1[8];
        *reinterpret_cast<uint4*>(v0) = *reinterpret_cast<const uint4*>(rowp + (size_t)kx * C);
        *reinterpret_cast<uint4*>(v1) = *reinterpret_cast<const uint4*>(rowp + (size_t)(kx + 1) * C);
        if (w0 != 0.f) {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += w0 * bf2f(v0[e]);
        }
        if (w1 != 0.f) {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += w1 * bf2f(v1[e]);
        }
      }
#endif
      for (; kx < nx; ++kx) {
        const float wgt = a * wtab[1][pw][kx];
        if (wgt == 0.f) continue;
        bf16_t v[8];
        *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(rowp + (size_t)kx * C);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += wgt * bf2f(v[e]);
      }
    }
#endif
    bf16_t o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e] * inv_cnt);
    *reinterpret_cast<uint4*>(out + (((size_t)r * P + ph) * P + pw) * C + cc * 8) = *reinterpret_cast<const uint4*>(o);
  }
}

// Separable form.  The samples of a bin form a product grid and a bilinear weight is w_y * w_x, so
//   sum_{iy, ix} w_y(iy) w_x(ix) = (sum_iy w_y(iy)) * (sum_ix w_x(ix)):
// per (ROI, pixel row, bin row) and (ROI, pixel column, bin column) the 1-D sums are tabulated in LDS by a few threads,
// and a pixel then needs one dout read per (bin row, bin column) pair it touches (typically 2 x 2) instead of one per
// sample (gh * gw per bin).  Up to four ROI sets (the three cascade box stages and the mask pooler; each with its own
// pooled size and gradient scale) are folded into one pass, so the level's gradient map is written exactly once.
struct RoiSetDev { const float* rois; const int* order; const int* seg; const bf16_t* dout; int P; float gscale; };
struct RoiSetsDev { RoiSetDev s[4]; int n; };

#ifndef GS_MAXL_V
#define GS_MAXL_V 256
#endif
constexpr int GS_TS = 8, GS_MAXL = GS_MAXL_V, GS_KB = 4, GS_MAXP = 14;
constexpr int GS_STAGE_BYTES = 24576;  // 48 bins of 256 channels
typedef float gs_f2 __attribute__((ext_vector_type(2)));

// The accumulation reads, per (pixel quad, ROI), the 2 x 2 ... 3 x 3 bins of dout that touch the quad: 16 quads of a tile
// re-read the same 16-25 bin rows of a ROI, and with the reads coming from L2 that re-reading was the kernel's time (1.17 of
// the 1.47 ms of the stride-4 level; 17 TB/s of 16-byte loads).  The bin rows a batch of ROIs has in common with the tile are
// therefore staged in LDS once (coalesced 512-byte rows) and the quads read them from there; a ROI whose bins do not fit
// the stage (a 14 x 14 mask ROI on a coarse level) keeps reading from L2.
#ifndef GS_OCC
#define GS_OCC 4
#endif
#ifdef GS_TRACE
// debug build (-DGS_TRACE, tools/exp/roi_gather_trace.sh): shader-clock ticks of thread 0 per phase, summed over the work-groups
// (one slot per work-group, plain adds: stamps through same-address atomics slowed the kernel six-fold)
constexpr int GS_TRACE_WGS = 20000;
__device__ unsigned long long g_gs_trace[GS_TRACE_WGS][8];
#define GS_T(k) do { if (tid == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); g_gs_trace[gs_wg][k] += now_ - t_last; t_last = now_; } } while (0)
#else
#define GS_T(k) do { } while (0)
#endif
__global__ __launch_bounds__(256, GS_OCC) void roi_align_bwd_gather_kernel(const RoiSetsDev sets, bf16_t* __restrict__ gfeat, int level,
                                                                   int nlevels, int H, int W, int C, float scale,
                                                                   const bf16_t* __restrict__ add0, const bf16_t* __restrict__ add1) {
  constexpr int TS = GS_TS;
  constexpr int NQ = 2;    // items per thread: an item = (tile row, 4-pixel quad of that row, 8-channel chunk)
  __shared__ RoiGeom list[GS_MAXL];
  __shared__ int nlist;
  __shared__ float tabY[GS_KB][TS][GS_MAXP];                                  // [roi][tile row][bin row]
  __shared__ __attribute__((aligned(16))) float tabX[GS_KB][GS_MAXP][TS];     // [roi][bin column][tile column]
  __shared__ int rmask[GS_KB], cmask[GS_KB];                                  // bins with a weight on some tile row / column
  // round 6: per tile row / per 4-pixel quad of columns, the bins that carry weight on it (double-buffered by batch parity: the
  // buffer of the next batch is cleared while this one is multiplied).  The accumulation walks the set bits instead of testing
  // every bin row and column of the ROI's rectangle for a zero weight - an LDS round trip per test, ~30 per (item, ROI), which a
  // phase trace (tools/exp/roi_gather_trace.sh) put at half of a work-group's life on the stride-4 level.
  __shared__ unsigned rowbits[2][GS_KB][GS_TS], colbits[2][GS_KB][2];
  __shared__ __attribute__((aligned(16))) unsigned char stage[GS_STAGE_BYTES];
  const int b = blockIdx.z;
  const int ty0 = blockIdx.y * TS, tx0 = blockIdx.x * TS;
  const int tid = threadIdx.x;
  const int cpr = C >> 3;             // 16-byte chunks per pixel
  const int items = TS * 2 * cpr;     // C <= 256: at most 512
  const int stage_slots = GS_STAGE_BYTES / (C * 2);
  gs_f2 acc[NQ][4][4];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[q][j][e] = gs_f2{0.f, 0.f};
  if (tid < GS_KB) { rmask[tid] = 0; cmask[tid] = 0; }
  if (tid < 2 * GS_KB * TS) (&rowbits[0][0][0])[tid] = 0u;
  if (tid < 2 * GS_KB * 2) (&colbits[0][0][0])[tid] = 0u;
  int par = 0;   // batch parity (uniform)
#ifdef GS_TRACE
  const int gs_wg = (int)(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % GS_TRACE_WGS;
  unsigned long long t_last = __builtin_readcyclecounter();
  if (tid == 0) g_gs_trace[gs_wg][7] += 1ull;
#endif

  for (int si = 0; si < sets.n; ++si) {
    const RoiSetDev st = sets.s[si];
    const int P = st.P;
    const int beg = st.seg[b * nlevels + level], end = st.seg[b * nlevels + level + 1];
    for (int base = beg; base < end; base += GS_MAXL) {
      __syncthreads();
      if (tid == 0) nlist = 0;
      __syncthreads();
      const int idx = base + tid;
      if (tid < GS_MAXL && idx < end) {
        const int r = st.order[idx];
        const float* roi = st.rois + (size_t)r * 5;
        RoiGeom g;
        g.sw = roi[1] * scale - 0.5f; g.sh = roi[2] * scale - 0.5f;
        const float ew = roi[3] * scale - 0.5f, eh = roi[4] * scale - 0.5f;
        const float rw = ew - g.sw, rh = eh - g.sh;
        g.bh = rh / (float)P; g.bw = rw / (float)P;
        g.gh = (int)ceilf(rh / (float)P); g.gw = (int)ceilf(rw / (float)P);
        g.r = r;
        g.inv_cnt = st.gscale / (float)max(g.gh * g.gw, 1);
        // pixel footprint of the ROI's taps (clamped samples pile up on the border rows / columns)
        g.py0 = max(0, (int)floorf(g.sh) - 1); g.py1 = min(H - 1, (int)ceilf(eh) + 1);
        g.px0 = max(0, (int)floorf(g.sw) - 1); g.px1 = min(W - 1, (int)ceilf(ew) + 1);
        if (g.gh > 0 && g.gw > 0 && g.py1 >= ty0 && g.py0 < ty0 + TS && g.px1 >= tx0 && g.px0 < tx0 + TS) {
          const int slot = atomicAdd(&nlist, 1);
          list[slot] = g;
        }
      }
      __syncthreads();
      const int n = nlist;
      GS_T(0);   // scan
      for (int k0 = 0; k0 < n; k0 += GS_KB) {
        const int nb = min(GS_KB, n - k0);
#ifdef GS_TRACE
        if (tid == 0) g_gs_trace[gs_wg][6] += 1ull;
#endif
        // 1-D weight sums of this batch of ROIs for the tile's 8 pixel rows and 8 pixel columns
        for (int t = tid; t < nb * 2 * TS * P; t += 256) {
          const int bin = t % P;
          int rest = t / P;
          const int rc = rest % TS; rest /= TS;
          const int axis = rest & 1, kk = rest >> 1;
          const RoiGeom& g = list[k0 + kk];
          float sum = 0.f;
          if (axis == 0) {
            const float inv_g = g.bh / (float)g.gh;
            for (int iy = 0; iy < g.gh; ++iy) sum += axis_weight(g.sh + bin * g.bh + (iy + 0.5f) * inv_g, H, ty0 + rc);
            tabY[kk][rc][bin] = sum;
            if (sum != 0.f) { atomicOr(&rmask[kk], 1 << bin); atomicOr(&rowbits[par][kk][rc], 1u << bin); }
          } else {
            const float inv_g = g.bw / (float)g.gw;
            for (int ix = 0; ix < g.gw; ++ix) sum += axis_weight(g.sw + bin * g.bw + (ix + 0.5f) * inv_g, W, tx0 + rc);
            tabX[kk][bin][rc] = sum;
            if (sum != 0.f) { atomicOr(&cmask[kk], 1 << bin); atomicOr(&colbits[par][kk][rc >> 2], 1u << bin); }
          }
        }
        __syncthreads();
        if (tid < GS_KB * TS) (&rowbits[par ^ 1][0][0])[tid] = 0u;
        if (tid < GS_KB * 2) (&colbits[par ^ 1][0][0])[tid] = 0u;
        GS_T(1);   // tables
        // the bin rectangle of every ROI of the batch that carries weight on the tile, and its place in the stage
        int ph_lo[GS_KB], ph_n[GS_KB], pw_lo[GS_KB], pw_n[GS_KB], sbase[GS_KB];
        int used = 0;
#pragma unroll
        for (int kk = 0; kk < GS_KB; ++kk) {
          const int rm = kk < nb ? rmask[kk] : 0, cm = kk < nb ? cmask[kk] : 0;
          ph_lo[kk] = rm ? __ffs(rm) - 1 : 0;
          ph_n[kk] = rm ? 32 - __clz(rm) - ph_lo[kk] : 0;
          pw_lo[kk] = cm ? __ffs(cm) - 1 : 0;
          pw_n[kk] = cm ? 32 - __clz(cm) - pw_lo[kk] : 0;
          if (!cm) ph_n[kk] = 0;
          const int bins = ph_n[kk] * pw_n[kk];
          sbase[kk] = -1;
          if (bins > 0 && used + bins <= stage_slots) { sbase[kk] = used; used += bins; }
        }
#pragma unroll
        for (int kk = 0; kk < GS_KB; ++kk) {
          if (sbase[kk] < 0) continue;
          const RoiGeom& g = list[k0 + kk];
          const bf16_t* src = st.dout + (size_t)g.r * P * P * C;
          const int chunks = ph_n[kk] * pw_n[kk] * cpr;
          for (int t = tid; t < chunks; t += 256) {
            const int ch = t % cpr, bin = t / cpr;
            const int ph = ph_lo[kk] + bin / pw_n[kk], pw = pw_lo[kk] + bin % pw_n[kk];
            *reinterpret_cast<uint4*>(stage + ((size_t)(sbase[kk] + bin) * cpr + ch) * 16) =
                *reinterpret_cast<const uint4*>(src + (size_t)(ph * P + pw) * C + ch * 8);
          }
        }
        __syncthreads();
        GS_T(2);   // staging
        if (tid < GS_KB) { rmask[tid] = 0; cmask[tid] = 0; }  // for the next batch (every thread holds its copy by now)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int item = q * 256 + tid;
          const int ch = item % cpr;
          const int rq = item / cpr;
          const int row = rq >> 1, quad = rq & 1;
          const int py = ty0 + row, px = tx0 + quad * 4;
          if (item >= items || py >= H || px >= W) continue;
#pragma unroll
          for (int kk = 0; kk < GS_KB; ++kk) {
            if (kk >= nb || ph_n[kk] == 0) continue;
            unsigned rb = rowbits[par][kk][row];
            const unsigned cb = colbits[par][kk][quad];
            if (rb == 0u || cb == 0u) continue;   // no weight on this row / these four columns (also: outside the ROI's footprint)
            const RoiGeom& g = list[k0 + kk];
            const float ic = g.inv_cnt;
            const bool staged = sbase[kk] >= 0;
            const bf16_t* rbase = st.dout + (size_t)g.r * P * P * C + ch * 8;
            const unsigned char* sb = stage + ((size_t)sbase[kk] * cpr + ch) * 16;
            while (rb) {
              const int ph = __ffs(rb) - 1;
              rb &= rb - 1u;
              const float ayc = tabY[kk][row][ph] * ic;
              const int i = ph - ph_lo[kk];
              unsigned m = cb;
              while (m) {   // two bin columns per turn: their four LDS reads are independent and go out together
                const int pw0 = __ffs(m) - 1;
                m &= m - 1u;
                const bool two = m != 0u;
                const int pw1 = two ? __ffs(m) - 1 : pw0;
                m &= m - 1u;
                const float4 ax0 = *reinterpret_cast<const float4*>(&tabX[kk][pw0][quad * 4]);
                float4 ax1 = *reinterpret_cast<const float4*>(&tabX[kk][pw1][quad * 4]);
                uint4 dq0, dq1;
                if (staged) {
                  dq0 = *reinterpret_cast<const uint4*>(sb + (size_t)(i * pw_n[kk] + pw0 - pw_lo[kk]) * cpr * 16);
                  dq1 = *reinterpret_cast<const uint4*>(sb + (size_t)(i * pw_n[kk] + pw1 - pw_lo[kk]) * cpr * 16);
                } else {
                  dq0 = *reinterpret_cast<const uint4*>(rbase + (size_t)(ph * P + pw0) * C);
                  dq1 = *reinterpret_cast<const uint4*>(rbase + (size_t)(ph * P + pw1) * C);
                }
                if (!two) ax1 = make_float4(0.f, 0.f, 0.f, 0.f);
                const uint32_t dw0[4] = {dq0.x, dq0.y, dq0.z, dq0.w}, dw1[4] = {dq1.x, dq1.y, dq1.z, dq1.w};
                const float u0 = ayc * ax0.x, u1 = ayc * ax0.y, u2 = ayc * ax0.z, u3 = ayc * ax0.w;
                const float v0 = ayc * ax1.x, v1 = ayc * ax1.y, v2 = ayc * ax1.z, v3 = ayc * ax1.w;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const gs_f2 d0 = gs_f2{__uint_as_float(dw0[e] << 16), __uint_as_float(dw0[e] & 0xffff0000u)};
                  const gs_f2 d1 = gs_f2{__uint_as_float(dw1[e] << 16), __uint_as_float(dw1[e] & 0xffff0000u)};
                  acc[q][0][e] = __builtin_elementwise_fma(gs_f2{u0, u0}, d0, acc[q][0][e]);
                  acc[q][1][e] = __builtin_elementwise_fma(gs_f2{u1, u1}, d0, acc[q][1][e]);
                  acc[q][2][e] = __builtin_elementwise_fma(gs_f2{u2, u2}, d0, acc[q][2][e]);
                  acc[q][3][e] = __builtin_elementwise_fma(gs_f2{u3, u3}, d0, acc[q][3][e]);
                  acc[q][0][e] = __builtin_elementwise_fma(gs_f2{v0, v0}, d1, acc[q][0][e]);
                  acc[q][1][e] = __builtin_elementwise_fma(gs_f2{v1, v1}, d1, acc[q][1][e]);
                  acc[q][2][e] = __builtin_elementwise_fma(gs_f2{v2, v2}, d1, acc[q][2][e]);
                  acc[q][3][e] = __builtin_elementwise_fma(gs_f2{v3, v3}, d1, acc[q][3][e]);
                }
              }
            }
          }
        }
        __syncthreads();
        par ^= 1;
        GS_T(3);   // accumulation
      }
    }
  }
  GS_T(4);     // what is left of the loops
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int item = q * 256 + tid;
    const int ch = item % cpr;
    const int rq = item / cpr;
    const int row = rq >> 1, quad = rq & 1;
    const int py = ty0 + row;
    if (item >= items || py >= H) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int px = tx0 + quad * 4 + j;
      if (px >= W) continue;
      const size_t at = (((size_t)b * H + py) * W + px) * C + ch * 8;
      // round 6: the map's other gradients (the RPN's and the semantic head's: autograd's accumulation over the map's readers)
      // are added here, in fp32, before the one rounding - the separate u2_add_n pass read this map back and the others again
      if (add0) {
        bf16_t u[8];
        *reinterpret_cast<uint4*>(u) = *reinterpret_cast<const uint4*>(add0 + at);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[q][j][e].x += bf2f(u[2 * e]); acc[q][j][e].y += bf2f(u[2 * e + 1]); }
      }
      if (add1) {
        bf16_t u[8];
        *reinterpret_cast<uint4*>(u) = *reinterpret_cast<const uint4*>(add1 + at);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[q][j][e].x += bf2f(u[2 * e]); acc[q][j][e].y += bf2f(u[2 * e + 1]); }
      }
      bf16_t o[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[2 * e] = f2bf(acc[q][j][e].x); o[2 * e + 1] = f2bf(acc[q][j][e].y); }
      *reinterpret_cast<uint4*>(gfeat + at) = *reinterpret_cast<const uint4*>(o);
    }
  }
  GS_T(5);     // epilogue (issue only: the stores retire behind the work-group)
}

// single-channel fp32 ROIAlign used for ground-truth mask crops (BitMasks.crop_and_resize,
// structures/masks.py:191-218): mask uint8 [Nm][H][W], rois [R][5] (mask index, box), out uint8 = (val >= 0.5)
__global__ __launch_bounds__(256) void mask_crop_kernel(const uint8_t* __restrict__ masks, const float* __restrict__ rois,
                                                        uint8_t* __restrict__ out, int R, int H, int W, int P) {
  const size_t total = (size_t)R * P * P;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    size_t p = i;
    const int pw = (int)(p % P); p /= P;
    const int ph = (int)(p % P);
    const int r = (int)(p / P);
    const float* roi = rois + (size_t)r * 5;
    const int b = (int)roi[0];
    const float sw = roi[1] - 0.5f, sh = roi[2] - 0.5f, ew = roi[3] - 0.5f, eh = roi[4] - 0.5f;
    const float rw = ew - sw, rh = eh - sh;
    const float bh = rh / (float)P, bw = rw / (float)P;
    const int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
    const float cnt = (float)max(gh * gw, 1);
    float acc = 0.f;
    const uint8_t* m = masks + (size_t)b * H * W;
    for (int iy = 0; iy < gh; ++iy) {
      const float y = sh + ph * bh + (iy + 0.5f) * bh / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        const float x = sw + pw * bw + (ix + 0.5f) * bw / (float)gw;
        int yl, xl, yh, xh;
        float w1, w2, w3, w4;
        if (!bil_prep(y, x, H, W, yl, xl, yh, xh, w1, w2, w3, w4)) continue;
        acc += w1 * (float)m[(size_t)yl * W + xl] + w2 * (float)m[(size_t)yl * W + xh] + w3 * (float)m[(size_t)yh * W + xl] +
               w4 * (float)m[(size_t)yh * W + xh];
      }
    }
    out[i] = (acc / cnt >= 0.5f) ? 1 : 0;
  }
}

// the same crop for a batch of images whose bitmaps live in separate tensors (possibly of different sizes): one launch,
// roi_image[r] selects the image of ROI r, rois[r][0] the bitmap row inside that image's tensor
struct MaskImages { const uint8_t* base[32]; int H[32]; int W[32]; };

__global__ __launch_bounds__(256) void mask_crop_batch_kernel(const MaskImages imgs, const float* __restrict__ rois,
                                                              const int* __restrict__ roi_image, uint8_t* __restrict__ out,
                                                              int R, int P, int img0) {
  const size_t total = (size_t)R * P * P;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    size_t p = i;
    const int pw = (int)(p % P); p /= P;
    const int ph = (int)(p % P);
    const int r = (int)(p / P);
    const int im = roi_image[r] - img0;
    if (im < 0 || im >= 32) continue;  // belongs to another chunk of images
    const int H = imgs.H[im], W = imgs.W[im];
    const float* roi = rois + (size_t)r * 5;
    const int b = (int)roi[0];
    const float sw = roi[1] - 0.5f, sh = roi[2] - 0.5f, ew = roi[3] - 0.5f, eh = roi[4] - 0.5f;
    const float rw = ew - sw, rh = eh - sh;
    const float bh = rh / (float)P, bw = rw / (float)P;
    const int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
    const float cnt = (float)max(gh * gw, 1);
    float acc = 0.f;
    const uint8_t* m = imgs.base[im] + (size_t)b * H * W;
    for (int iy = 0; iy < gh; ++iy) {
      const float y = sh + ph * bh + (iy + 0.5f) * bh / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        const float x = sw + pw * bw + (ix + 0.5f) * bw / (float)gw;
        int yl, xl, yh, xh;
        float w1, w2, w3, w4;
        if (!bil_prep(y, x, H, W, yl, xl, yh, xh, w1, w2, w3, w4)) continue;
        acc += w1 * (float)m[(size_t)yl * W + xl] + w2 * (float)m[(size_t)yl * W + xh] + w3 * (float)m[(size_t)yh * W + xl] +
               w4 * (float)m[(size_t)yh * W + xh];
      }
    }
    out[i] = (acc / cnt >= 0.5f) ? 1 : 0;
  }
}

__global__ void assign_levels_kernel(const float* __restrict__ boxes, int* __restrict__ level, int n, int min_level,
                                     int max_level, float canonical_size, int canonical_level) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* b = boxes + (size_t)i * 4;
  const float area = (b[2] - b[0]) * (b[3] - b[1]);
  const float s = sqrtf(area);
  float l = floorf((float)canonical_level + log2f(s / canonical_size + 1e-8f));
  l = fminf(fmaxf(l, (float)min_level), (float)max_level);
  level[i] = (int)l - min_level;
}

__device__ __forceinline__ float box_iou(const float* a, const float* b) {
  const float aa = (a[2] - a[0]) * (a[3] - a[1]);
  const float ab = (b[2] - b[0]) * (b[3] - b[1]);
  const float w = fminf(a[2], b[2]) - fmaxf(a[0], b[0]);
  const float h = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
  const float inter = fmaxf(w, 0.f) * fmaxf(h, 0.f);
  return inter > 0.f ? inter / (aa + ab - inter) : 0.f;
}

// pass 1: per candidate box the best gt (first maximum) and its IoU; per gt the max IoU over candidates.
// boxes [B][n][4] (per_image_boxes) or [n][4] shared by all images (anchors); gt [B][G][4]; ngt [B].
__global__ __launch_bounds__(256) void iou_match_kernel(const float* __restrict__ boxes, int per_image_boxes,
                                                        const float* __restrict__ gt, const int* __restrict__ ngt,
                                                        int* __restrict__ match, float* __restrict__ mval,
                                                        unsigned int* __restrict__ gt_max, int n, int G) {
  const int b = blockIdx.y;
  const int ng = ngt[b];
  extern __shared__ float sgt[];  // [G][4]
  for (int i = threadIdx.x; i < ng * 4; i += 256) sgt[i] = gt[(size_t)b * G * 4 + i];
  __syncthreads();
  // per-gt maxima are first reduced inside the work-group (LDS atomics), then one global atomic per gt
  unsigned int* smax = reinterpret_cast<unsigned int*>(sgt + G * 4);
  if (gt_max)
    for (int i = threadIdx.x; i < ng; i += 256) smax[i] = 0u;
  __syncthreads();
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float* bx = boxes + ((size_t)(per_image_boxes ? b : 0) * n + i) * 4;
    const float a[4] = {bx[0], bx[1], bx[2], bx[3]};
    float best = 0.f;
    int bi = 0;
    for (int g = 0; g < ng; ++g) {
      const float v = box_iou(sgt + g * 4, a);
      if (g == 0 || v > best) { best = v; bi = g; }
      if (gt_max && v > 0.f) atomicMax(smax + g, __float_as_uint(v));
    }
    match[(size_t)b * n + i] = bi;
    mval[(size_t)b * n + i] = best;
  }
  if (gt_max) {
    __syncthreads();
    for (int i = threadIdx.x; i < ng; i += 256)
      if (smax[i]) atomicMax(gt_max + (size_t)b * G + i, smax[i]);
  }
}

// pass 2: labels from thresholds (val < lo -> 0, lo <= val < hi -> -1, val >= hi -> 1) and, when
// allow_low_quality, label 1 for every candidate whose IoU with some gt equals that gt's maximum.
__global__ __launch_bounds__(256) void match_label_kernel(const float* __restrict__ boxes, int per_image_boxes,
                                                          const float* __restrict__ gt, const int* __restrict__ ngt,
                                                          const float* __restrict__ mval,
                                                          const unsigned int* __restrict__ gt_max,
                                                          int8_t* __restrict__ labels, int n, int G, float lo, float hi,
                                                          int allow_low_quality) {
  const int b = blockIdx.y;
  const int ng = ngt[b];
  extern __shared__ float sgt[];  // [G][4] boxes then [G] maxima
  float* smax = sgt + G * 4;
  for (int i = threadIdx.x; i < ng * 4; i += 256) sgt[i] = gt[(size_t)b * G * 4 + i];
  if (allow_low_quality)
    for (int i = threadIdx.x; i < ng; i += 256) smax[i] = __uint_as_float(gt_max[(size_t)b * G + i]);
  __syncthreads();
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    int8_t lab;
    if (ng == 0) {
      lab = 0;
    } else {
      const float v = mval[(size_t)b * n + i];
      lab = v < lo ? 0 : (v < hi ? -1 : 1);
      if (allow_low_quality) {
        const float* bx = boxes + ((size_t)(per_image_boxes ? b : 0) * n + i) * 4;
        const float a[4] = {bx[0], bx[1], bx[2], bx[3]};
        for (int g = 0; g < ng; ++g)
          if (box_iou(sgt + g * 4, a) == smax[g]) lab = 1;
      }
    }
    labels[(size_t)b * n + i] = lab;
  }
}

// boxes = apply_deltas(deltas, src) then clip to (h, w) of the image `img[i]`
__global__ void apply_deltas_kernel(const float* __restrict__ src, const float* __restrict__ deltas,
                                    const int* __restrict__ img, const float* __restrict__ sizes, float* __restrict__ out,
                                    int n, float wx, float wy, float ww, float wh, float clamp, int do_clip) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* s = src + (size_t)i * 4;
  const float* d = deltas + (size_t)i * 4;
  const float w = s[2] - s[0], h = s[3] - s[1];
  const float cx = s[0] + 0.5f * w, cy = s[1] + 0.5f * h;
  const float dx = d[0] / wx, dy = d[1] / wy;
  float dw = d[2] / ww, dh = d[3] / wh;
  dw = fminf(dw, clamp);
  dh = fminf(dh, clamp);
  const float pcx = dx * w + cx, pcy = dy * h + cy;
  const float pw = expf(dw) * w, ph = expf(dh) * h;
  float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
  if (do_clip) {
    const int b = img ? img[i] : 0;
    const float H = sizes[b * 2 + 0], W = sizes[b * 2 + 1];
    x1 = fminf(fmaxf(x1, 0.f), W); x2 = fminf(fmaxf(x2, 0.f), W);
    y1 = fminf(fmaxf(y1, 0.f), H); y2 = fminf(fmaxf(y2, 0.f), H);
  }
  float* o = out + (size_t)i * 4;
  o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2;
}

// ---- RPN proposal decoding, all levels in one launch (round 5) ---------------------------------------------------------------
// rpn.py:482-533 + proposal_utils.py:56-91 for the candidates the per-level top-k selected: gather the 4 deltas of anchor idx from
// the NHWC bf16 map (pixel idx / A, channels (idx % A) * 4 .. + 3 of the deltas view), apply them to the anchor, clip to the image,
// and emit the row layout the per-(image, level) NMS works on: boxes / scores [B * L][kmax] (row = image * L + level, a short
// level padded with zero boxes and score -3e38) and keep = finite & wider and taller than min_size.  Non-finite candidates are
// counted (the caller raises on the host when it first reads counts).  Same arithmetic as apply_deltas_kernel above.
struct RpnDecodeArgs {
  U2RpnLevel lv[8];
  int L, A, B, kmax;
  const float* sizes;
  float wx, wy, ww, wh, clamp, min_size;
  float* boxes; float* scores; signed char* keep; int* nonfinite;
};
__global__ __launch_bounds__(256) void rpn_decode_kernel(const RpnDecodeArgs a) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = blockIdx.y;             // image * L + level
  if (j >= a.kmax) return;
  const int b = row / a.L, l = row - b * a.L;
  const U2RpnLevel lv = a.lv[l];
  float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, sc = -3.0e38f;
  bool keep = false;
  if (j < lv.k) {
    const int i = lv.idx[(size_t)b * lv.k + j];
    const int pix = i / a.A, anc = i - pix * a.A;
    const bf16_t* dp = reinterpret_cast<const bf16_t*>(lv.deltas) + ((size_t)b * (lv.hwa / a.A) + pix) * lv.pitch + anc * 4;
    const float* s = lv.anchors + (size_t)i * 4;
    const float d0 = bf2f(dp[0]), d1 = bf2f(dp[1]), d2 = bf2f(dp[2]), d3 = bf2f(dp[3]);
    const float w = s[2] - s[0], h = s[3] - s[1];
    const float cx = s[0] + 0.5f * w, cy = s[1] + 0.5f * h;
    const float dx = d0 / a.wx, dy = d1 / a.wy;
    float dw = d2 / a.ww, dh = d3 / a.wh;
    dw = fminf(dw, a.clamp);
    dh = fminf(dh, a.clamp);
    const float pcx = dx * w + cx, pcy = dy * h + cy;
    const float pw = expf(dw) * w, ph = expf(dh) * h;
    x1 = pcx - 0.5f * pw; y1 = pcy - 0.5f * ph; x2 = pcx + 0.5f * pw; y2 = pcy + 0.5f * ph;
    sc = lv.scores[(size_t)b * lv.k + j];
    // the reference tests the UNCLIPPED boxes (proposal_utils.py:93-99, clip comes after): fminf / fmaxf swallow NaN and
    // clamp Inf, so a test behind the clamp could never fire and diverged deltas would turn into silently dropped empty boxes
    const bool finite = isfinite(x1) && isfinite(y1) && isfinite(x2) && isfinite(y2) && isfinite(sc);
    const float H = a.sizes[b * 2 + 0], W = a.sizes[b * 2 + 1];
    x1 = fminf(fmaxf(x1, 0.f), W); x2 = fminf(fmaxf(x2, 0.f), W);
    y1 = fminf(fmaxf(y1, 0.f), H); y2 = fminf(fmaxf(y2, 0.f), H);
    if (!finite) atomicAdd(a.nonfinite, 1);
    keep = finite && (x2 - x1) > a.min_size && (y2 - y1) > a.min_size;
  }
  const size_t o = (size_t)row * a.kmax + j;
  float* bo = a.boxes + o * 4;
  bo[0] = x1; bo[1] = y1; bo[2] = x2; bo[3] = y2;
  a.scores[o] = sc;
  a.keep[o] = keep ? 1 : 0;
}

// ---- NMS: suppression bit matrix, then an in-order scan (64 rows at a time) ----
// boxes [B][n][4] sorted by descending score, group [B][n] int32, cnt [B]
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, const int* __restrict__ group,
                                                      const int* __restrict__ cnt, unsigned long long* __restrict__ mask,
                                                      int n, int NB, float thr) {
  const int b = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;
  const int nb = cnt[b];
  if (rb * 64 >= nb || cb * 64 >= nb) return;
  __shared__ float cbx[64][4];
  __shared__ int cgr[64];
  const int t = threadIdx.x;
  const int j = cb * 64 + t;
  if (j < nb) {
    const float* q = boxes + ((size_t)b * n + j) * 4;
    cbx[t][0] = q[0]; cbx[t][1] = q[1]; cbx[t][2] = q[2]; cbx[t][3] = q[3];
    cgr[t] = group[(size_t)b * n + j];
  }
  __syncthreads();
  const int i = rb * 64 + t;
  if (i >= nb) return;
  const float* q = boxes + ((size_t)b * n + i) * 4;
  const float a[4] = {q[0], q[1], q[2], q[3]};
  const float aa = (a[2] - a[0]) * (a[3] - a[1]);
  const int gi = group[(size_t)b * n + i];
  unsigned long long bits = 0;
  const int jn = min(64, nb - cb * 64);
  for (int k = (rb == cb ? t + 1 : 0); k < jn; ++k) {
    if (cgr[k] != gi) continue;
    const float w = fmaxf(fminf(a[2], cbx[k][2]) - fmaxf(a[0], cbx[k][0]), 0.f);
    const float h = fmaxf(fminf(a[3], cbx[k][3]) - fmaxf(a[1], cbx[k][1]), 0.f);
    const float inter = w * h;
    const float ab = (cbx[k][2] - cbx[k][0]) * (cbx[k][3] - cbx[k][1]);
    if (inter / (aa + ab - inter) > thr) bits |= 1ull << k;
  }
  mask[((size_t)b * n + i) * NB + cb] = bits;
}

// In-order scan.  Greedy NMS is serial over the candidates, so the kernel is organised around the length of the
// dependent chain per 64-candidate block: wave 0 resolves the block in registers (64 shuffle steps); meanwhile waves 1-3
// already OR together, for the NEXT block's column word, the suppression rows of every box kept before this block (a
// lazy column-wise update: nothing that is not about to be consumed is touched); after one barrier all threads add the
// rows kept in this block (<= 64 loads, one latency) and the next block's "removed" word is final.  The diagonal rows
// of the next block are fetched in the same shadow.  The previous version updated all later column words eagerly after
// every block (64 dependent loads per thread, ~22 us per block); this one needs ~4 us per block.
__global__ __launch_bounds__(256) void nms_scan_kernel(const unsigned long long* __restrict__ mask, const int* __restrict__ cnt,
                                                       int* __restrict__ keep, int* __restrict__ nkeep, int n, int NB,
                                                       int max_keep) {
  extern __shared__ int kept_rows[];  // [max_keep] candidate indices kept so far
  __shared__ unsigned long long s_removed;   // final "removed" word of the block about to be scanned
  __shared__ unsigned long long s_partial;   // OR of older kept rows for the next block's word
  __shared__ unsigned long long s_diag[64];  // diagonal rows of the block about to be scanned
  __shared__ int s_count, s_prev;
  const int b = blockIdx.x;
  const int nb = cnt[b];
  const int nblk = (nb + 63) / 64;
  const int tid = threadIdx.x;
  const size_t rowbase = (size_t)b * n;
  if (tid == 0) { s_count = 0; s_prev = 0; s_removed = 0ull; s_partial = 0ull; }
  if (tid < 64) s_diag[tid] = (tid < nb) ? mask[(rowbase + tid) * NB + 0] : 0ull;
  __syncthreads();
  for (int cb = 0; cb < nblk; ++cb) {
    const int kprev = s_count;  // boxes kept before this block
    if (tid < 64) {
      // ---- wave 0: resolve block cb ----
      const unsigned long long dm = s_diag[tid];
      unsigned long long rem = s_removed;
      if (nb - cb * 64 < 64) rem |= ~0ull << (nb - cb * 64);
      unsigned long long kept = 0;
      for (int t = 0; t < 64; ++t) {
        const unsigned long long row = __shfl(dm, t, 64);
        if (!((rem >> t) & 1ull)) { kept |= 1ull << t; rem |= row; }
      }
      if ((kept >> tid) & 1ull) {
        const int pos = kprev + __popcll(kept & ((1ull << tid) - 1ull));
        if (pos < max_keep) { keep[(size_t)b * max_keep + pos] = cb * 64 + tid; kept_rows[pos] = cb * 64 + tid; }
      }
      if (tid == 0) { s_prev = kprev; s_count = min(kprev + __popcll(kept), max_keep + 64); }
    } else if (cb + 1 < nblk) {
      // ---- waves 1-3: older kept rows, next block's column word ----
      unsigned long long acc = 0ull;
      for (int k = tid - 64; k < min(kprev, max_keep); k += 192) acc |= mask[(rowbase + kept_rows[k]) * NB + cb + 1];
      for (int off = 32; off > 0; off >>= 1) acc |= __shfl_xor(acc, off, 64);
      if ((tid & 63) == 0 && acc) atomicOr(&s_partial, acc);
    }
    __syncthreads();
    const int knew = min(s_count, max_keep);
    if (s_count >= max_keep || cb + 1 >= nblk) break;
    // ---- everybody: rows kept in this block, next block's word; fetch the next diagonal block ----
    {
      unsigned long long acc = 0ull;
      for (int k = s_prev + tid; k < knew; k += 256) acc |= mask[(rowbase + kept_rows[k]) * NB + cb + 1];
      if (acc) atomicOr(&s_partial, acc);
      if (tid >= 192) {
        const int i = (cb + 1) * 64 + (tid - 192);
        s_diag[tid - 192] = (i < nb) ? mask[(rowbase + i) * NB + cb + 1] : 0ull;
      }
    }
    __syncthreads();
    if (tid == 0) { s_removed = s_partial; s_partial = 0ull; }
    __syncthreads();
  }
  if (tid == 0) nkeep[b] = min(s_count, max_keep);
}

}  // namespace

static int ew_blocks(size_t total) {
  size_t g = (total + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int u2_roi_align_fwd(const void* const* feats, const int* Hs, const int* Ws, const float* scales, int nlevels,
                                const float* rois, const int* level, const int* order, void* out, int R, int C, int PH, int PW,
                                void* stream) {
  if (nlevels < 1 || nlevels > 4 || (C & 7)) return -1;
  if (R <= 0) return 0;
  RoiLevels lv;
  for (int l = 0; l < 4; ++l) {
    const int s = l < nlevels ? l : 0;
    lv.feat[l] = (const bf16_t*)feats[s]; lv.gfeat[l] = nullptr; lv.H[l] = Hs[s]; lv.W[l] = Ws[s]; lv.scale[l] = scales[s];
  }
  if (PH == PW && PH <= FS_MAXP) {  // separable per-ROI kernel
    // measured (scratch-style A/B, 32 000 ROIs of 7 x 7 / 3 200 of 14 x 14, round 5): the bin-row form needs 28 % fewer L1 accesses
    // and is 14 % / 44 % SLOWER (1.70 vs 1.49 ms, 0.62 vs 0.43) - the pass is bound by L2 misses (42 % hit rate, 6 GB of fabric
    // reads per launch for 1.5 GB of maps, profiles/r05_pmc_roi.txt), not by hits; it stays selectable (U2_ROI_FWD=rows)
    static const int per_bin = [] { const char* e = getenv("U2_ROI_FWD"); return e && !strcmp(e, "rows") ? 0 : 1; }();
    lv.flags = per_bin;
    if (PH == 7 && !per_bin)
      hipLaunchKernelGGL(roi_align_fwd_sep_kernel<7>, dim3(R), dim3(256), 0, (hipStream_t)stream, lv, rois, level, order, (bf16_t*)out, C, PH);
    else if (PH == 14 && !per_bin)
      hipLaunchKernelGGL(roi_align_fwd_sep_kernel<14>, dim3(R), dim3(256), 0, (hipStream_t)stream, lv, rois, level, order, (bf16_t*)out, C, PH);
    else
      hipLaunchKernelGGL(roi_align_fwd_sep_kernel<0>, dim3(R), dim3(256), 0, (hipStream_t)stream, lv, rois, level, order, (bf16_t*)out, C, PH);
    U2_CHECK_LAUNCH();
    return 0;
  }
  const size_t total = (size_t)R * PH * PW * (C >> 3);
  hipLaunchKernelGGL(roi_align_kernel<false>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, lv, rois, level,
                     (bf16_t*)out, nullptr, R, C, PH, PW, 1.f);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_roi_align_bwd(float* const* gfeats, const int* Hs, const int* Ws, const float* scales, int nlevels,
                                const float* rois, const int* level, const void* dout, int R, int C, int PH, int PW,
                                float gscale, void* stream) {
  if (nlevels < 1 || nlevels > 4 || (C & 7)) return -1;
  if (R <= 0) return 0;
  RoiLevels lv;
  for (int l = 0; l < 4; ++l) {
    const int s = l < nlevels ? l : 0;
    lv.feat[l] = nullptr; lv.gfeat[l] = gfeats[s]; lv.H[l] = Hs[s]; lv.W[l] = Ws[s]; lv.scale[l] = scales[s];
  }
  const size_t total = (size_t)R * PH * PW * (C >> 3);
  hipLaunchKernelGGL(roi_align_kernel<true>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, lv, rois, level,
                     nullptr, (const bf16_t*)dout, R, C, PH, PW, gscale);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_mask_crop(const void* masks, const float* rois, void* out, int R, int H, int W, int P, void* stream) {
  if (R <= 0) return 0;
  hipLaunchKernelGGL(mask_crop_kernel, dim3(ew_blocks((size_t)R * P * P)), dim3(256), 0, (hipStream_t)stream,
                     (const uint8_t*)masks, rois, (uint8_t*)out, R, H, W, P);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_mask_crop_batch(const void* const* mask_bases, const int* Hs, const int* Ws, int n_images, const float* rois,
                                  const int* roi_image, void* out, int R, int P, void* stream) {
  if (R <= 0 || n_images <= 0) return 0;
  for (int i0 = 0; i0 < n_images; i0 += 32) {
    const int nb = n_images - i0 < 32 ? n_images - i0 : 32;
    MaskImages mi;
    for (int i = 0; i < nb; ++i) { mi.base[i] = (const uint8_t*)mask_bases[i0 + i]; mi.H[i] = Hs[i0 + i]; mi.W[i] = Ws[i0 + i]; }
    hipLaunchKernelGGL(mask_crop_batch_kernel, dim3(ew_blocks((size_t)R * P * P)), dim3(256), 0, (hipStream_t)stream, mi, rois,
                       roi_image, (uint8_t*)out, R, P, i0);
    U2_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int u2_assign_levels(const float* boxes, int* level, int n, int min_level, int max_level, float canonical_size,
                                int canonical_level, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(assign_levels_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, boxes, level, n,
                     min_level, max_level, canonical_size, canonical_level);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_iou_match(const float* boxes, int per_image_boxes, const float* gt, const int* ngt, int* match,
                            float* mval, unsigned int* gt_max, signed char* labels, int B, int n, int G, float lo, float hi,
                            int allow_low_quality, void* stream) {
  if (B <= 0 || n <= 0) return 0;
  if (G < 1) return -1;
  hipStream_t s = (hipStream_t)stream;
  int gx = (n + 255) / 256;
  if (gx > 1024) gx = 1024;
  const size_t lds = (size_t)G * 5 * sizeof(float);
  if (allow_low_quality) {
    if (!gt_max) return -1;
    u2_zero_words(gt_max, (size_t)B * G, s);
    U2_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(iou_match_kernel, dim3(gx, B), dim3(256), lds, s, boxes, per_image_boxes, gt, ngt, match, mval,
                     allow_low_quality ? gt_max : nullptr, n, G);
  U2_CHECK_LAUNCH();
  hipLaunchKernelGGL(match_label_kernel, dim3(gx, B), dim3(256), lds, s, boxes, per_image_boxes, gt, ngt, mval, gt_max,
                     (int8_t*)labels, n, G, lo, hi, allow_low_quality);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_apply_deltas(const float* src, const float* deltas, const int* img, const float* sizes, float* out, int n,
                               float wx, float wy, float ww, float wh, float clamp, int do_clip, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(apply_deltas_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, deltas, img, sizes,
                     out, n, wx, wy, ww, wh, clamp, do_clip);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_rpn_decode(const U2RpnLevel* levels, int L, int A, int B, int kmax, const float* sizes, float wx, float wy, float ww,
                             float wh, float clamp, float min_size, float* boxes, float* scores, signed char* keep, int* nonfinite,
                             void* stream) {
  if (L < 1 || L > 8 || A < 1 || kmax < 1) return -1;
  if (B <= 0) return 0;
  RpnDecodeArgs a;
  for (int l = 0; l < L; ++l) {
    a.lv[l] = levels[l];
    if (levels[l].k > kmax || levels[l].k < 0 || levels[l].hwa % A || (levels[l].pitch & 3)) return -1;
  }
  a.L = L; a.A = A; a.B = B; a.kmax = kmax; a.sizes = sizes;
  a.wx = wx; a.wy = wy; a.ww = ww; a.wh = wh; a.clamp = clamp; a.min_size = min_size;
  a.boxes = boxes; a.scores = scores; a.keep = keep; a.nonfinite = nonfinite;
  hipLaunchKernelGGL(rpn_decode_kernel, dim3((kmax + 255) / 256, B * L), dim3(256), 0, (hipStream_t)stream, a);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" long long u2_nms_workspace_bytes(int B, int n) {
  const long long NB = (n + 63) / 64;
  return (long long)B * n * NB * 8;
}

extern "C" int u2_batched_nms(const float* boxes, const int* group, const int* cnt, void* workspace, int* keep, int* nkeep,
                              int B, int n, float thr, int max_keep, void* stream) {
  if (B <= 0 || n <= 0) return 0;
  if (max_keep > 16384) return -1;  // the scan keeps the kept list (4 B per box) in LDS
  const int NB = (n + 63) / 64;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(NB, NB, B), dim3(64), 0, s, boxes, group, cnt, (unsigned long long*)workspace, n,
                     NB, thr);
  U2_CHECK_LAUNCH();
  hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(256), (size_t)(max_keep > 0 ? max_keep : 1) * 4, s,
                     (const unsigned long long*)workspace, cnt, keep, nkeep, n, NB, max_keep);
  U2_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// ROIs grouped by (image, level) for the gather above: order[R] = the ROI indices sorted by key = image * nlevels + level,
// equal keys in index order (a stable sort: the gather adds a pixel's ROIs in this order), seg[k] = first position of key k,
// seg[nkeys] = R.  One work-group, a counting sort in LDS: thread t owns the t-th contiguous slice of the indices; it counts its
// slice per key (cnt[t][k]), thread k turns column k into exclusive prefixes over the slices, the key bases are a scan of the
// column totals, and thread t places its slice in index order - stable by construction, no host synchronisation.
// (torch: argsort(stable) + bincount + cumsum + casts, 0.18 ms of device time and one host synchronisation per call, four calls
//  per step; a first version with one thread per key walking all R keys took as long, 164 us: 8192 dependent LDS reads.)
// ---------------------------------------------------------------------------------------------
constexpr int RG_MAXKEYS = 256, RG_MAXR = 32768, RG_LDS_LIMIT = 160 * 1024 - 4096;
__global__ __launch_bounds__(256) void roi_group_kernel(const float* __restrict__ rois, const int* __restrict__ level,
                                                        int* __restrict__ order, int* __restrict__ seg, int R, int nl,
                                                        int nkeys) {
  extern __shared__ unsigned short rg_lds[];
  __shared__ int total[RG_MAXKEYS], base[RG_MAXKEYS];
  const int pitch = nkeys + 2;                          // 16-bit counters; (nkeys + 2) / 2 words per row is odd: no bank conflicts
  unsigned short* cnt = rg_lds;                         // [256][pitch]
  unsigned short* keys = rg_lds + 256 * pitch;          // [R]
  const int tid = threadIdx.x;
  for (int i = tid; i < 256 * pitch; i += 256) cnt[i] = 0;
  for (int i = tid; i < R; i += 256) {
    const int k = (int)rois[(size_t)i * 5] * nl + level[i];
    keys[i] = (unsigned short)min(max(k, 0), nkeys - 1);
  }
  const int slice = (R + 255) / 256;
  const int i0 = min(R, tid * slice), i1 = min(R, i0 + slice);
  __syncthreads();
  unsigned short* mine = cnt + tid * pitch;
  for (int i = i0; i < i1; ++i) ++mine[keys[i]];
  __syncthreads();
  if (tid < nkeys) {                                    // column tid: counts of the slices -> exclusive prefixes, and the total
    int run = 0;
    for (int t = 0; t < 256; ++t) {
      const int c = cnt[t * pitch + tid];
      cnt[t * pitch + tid] = (unsigned short)run;
      run += c;
    }
    total[tid] = run;
  }
  __syncthreads();
  if (tid == 0) {
    int pos = 0;
    for (int k = 0; k < nkeys; ++k) { base[k] = pos; seg[k] = pos; pos += total[k]; }
    seg[nkeys] = pos;
  }
  __syncthreads();
  for (int i = i0; i < i1; ++i) {
    const int k = keys[i];
    order[base[k] + mine[k]++] = i;
  }
}

extern "C" int u2_roi_group(const float* rois, const int* level, int* order, int* seg, int R, int num_images, int nlevels,
                            void* stream) {
  const int nkeys = num_images * nlevels;
  if (R < 0 || nkeys < 1 || nkeys > RG_MAXKEYS || R > RG_MAXR) return -1;
  const size_t lds = ((size_t)256 * (nkeys + 2) + (size_t)(R > 0 ? R : 1)) * 2;
  if (lds > (size_t)RG_LDS_LIMIT) return -1;
  static PerDeviceOnce attr_set;
  if (auto once_guard = attr_set.first()) {
    (void)hipFuncSetAttribute((const void*)roi_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RG_LDS_LIMIT);
  }
  hipLaunchKernelGGL(roi_group_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, rois, level, order, seg, R, nlevels, nkeys);
  U2_CHECK_LAUNCH();
  return 0;
}

#ifdef GS_TRACE
extern "C" int u2_debug_gs_trace(unsigned long long* out8, int reset) {   // sums over the work-group slots
  static std::vector<unsigned long long> h((size_t)GS_TRACE_WGS * 8);
  if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_gs_trace), h.size() * 8) != hipSuccess) return -1;
  for (int k = 0; k < 8; ++k) out8[k] = 0;
  for (int w = 0; w < GS_TRACE_WGS; ++w)
    for (int k = 0; k < 8; ++k) out8[k] += h[(size_t)w * 8 + k];
  if (reset) {
    std::fill(h.begin(), h.end(), 0ull);
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_gs_trace), h.data(), h.size() * 8) != hipSuccess) return -1;
  }
  return 0;
}
#endif

extern "C" int u2_roi_align_bwd_gather_sum(void* const* gfeats, const int* Hs, const int* Ws, const float* scales, int nlevels,
                                           int level_mask, int nsets, const void* const* rois, const void* const* order,
                                           const void* const* seg, const void* const* dout, const int* P, const float* gscale,
                                           const void* const* add0, const void* const* add1, int B, int C, void* stream) {
  if (nlevels < 1 || nlevels > 4 || (C & 7) || C > 256 || nsets < 1 || nsets > 4) return -1;
  if (B <= 0) return 0;
  RoiSetsDev sets;
  sets.n = nsets;
  for (int i = 0; i < nsets; ++i) {
    if (P[i] < 1 || P[i] > GS_MAXP) return -1;
    sets.s[i].rois = (const float*)rois[i]; sets.s[i].order = (const int*)order[i]; sets.s[i].seg = (const int*)seg[i];
    sets.s[i].dout = (const bf16_t*)dout[i]; sets.s[i].P = P[i]; sets.s[i].gscale = gscale[i];
  }
  const char* e_lv = getenv("U2_ROI_LEVEL");  // experiments: launch one level only
  const int only = e_lv ? atoi(e_lv) : -1;
  for (int l = 0; l < nlevels; ++l) {
    if ((only >= 0 && l != only) || !((level_mask >> l) & 1) || !gfeats[l]) continue;
    const dim3 grid((Ws[l] + 7) / 8, (Hs[l] + 7) / 8, B);
    hipLaunchKernelGGL(roi_align_bwd_gather_kernel, grid, dim3(256), 0, (hipStream_t)stream, sets, (bf16_t*)gfeats[l], l,
                       nlevels, Hs[l], Ws[l], C, scales[l], add0 ? (const bf16_t*)add0[l] : nullptr,
                       add1 ? (const bf16_t*)add1[l] : nullptr);
    U2_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int u2_roi_align_bwd_gather_multi(void* const* gfeats, const int* Hs, const int* Ws, const float* scales,
                                             int nlevels, int nsets, const void* const* rois, const void* const* order,
                                             const void* const* seg, const void* const* dout, const int* P,
                                             const float* gscale, int B, int C, void* stream) {
  return u2_roi_align_bwd_gather_sum(gfeats, Hs, Ws, scales, nlevels, 0xf, nsets, rois, order, seg, dout, P, gscale, nullptr, nullptr,
                                     B, C, stream);
}

extern "C" int u2_roi_align_bwd_gather(void* const* gfeats, const int* Hs, const int* Ws, const float* scales, int nlevels,
                                       const float* rois, const int* order, const int* seg, const void* dout, int B, int C,
                                       int PH, int PW, float gscale, void* stream) {
  if (PH != PW) return -1;
  const void* r[1] = {rois}; const void* o[1] = {order}; const void* sg[1] = {seg}; const void* d[1] = {dout};
  return u2_roi_align_bwd_gather_multi(gfeats, Hs, Ws, scales, nlevels, 1, r, o, sg, d, &PH, &gscale, B, C, stream);
}
