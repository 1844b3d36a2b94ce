// Inference tails of the panoptic step (HBM-bound byte / integer work).
//
// Replaces (reference file:line):
//   paste_masks_in_image / _do_paste_mask (GPU branch)          detectron2/layers/mask_ops.py:17-147
//   combine_semantic_and_instance_outputs                       detectron2/modeling/meta_arch/panoptic_fpn.py:184-269
#include "common.h"
#include "u2seg_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Mask paste: out[k][y][x] = bilinear(prob[k], grid) >= thr with the sampling grid of F.grid_sample(align_corners=False)
// over the box, zero outside the P x P map.  The output is walked as a flat byte array, 8 bytes per thread (one aligned
// 64-bit store); pixels farther than one source texel from the box are written as 0 without sampling.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float paste_axis_coord(float pix, float lo, float hi, int P) {
  const float g = (pix + 0.5f - lo) / (hi - lo) * 2.f - 1.f;   // the normalised grid value the reference builds
  return ((g + 1.f) * (float)P - 1.f) / 2.f;                   // grid_sample's unnormalisation, align_corners=False
}

__device__ __forceinline__ void paste_masks_block(const float* __restrict__ probs, const float* __restrict__ boxes,
                                                  uint8_t* __restrict__ out, long long total, int P, int H, int W, float thr,
                                                  long long block) {
  const long long i8 = (block * 256 + threadIdx.x) * 8;
  if (i8 >= total) return;
  const long long hw = (long long)H * W;
  // (mask, row, column) of the block's first byte by scalar divisions (uniform), of the thread's first byte by at most a few
  // subtractions: two 64-bit divisions per thread were a third of the kernel
  const long long b8 = block * 2048;
  int k = (int)(b8 / hw);
  const long long rem = b8 - (long long)k * hw;
  int y = (int)(rem / W);
  int x = (int)(rem - (long long)y * W) + (int)threadIdx.x * 8;
  while (x >= W) { x -= W; ++y; }
  while (y >= H) { y -= H; ++k; }
  // 8 bytes inside one row that the mask's box does not reach (most of an 800 x 1333 canvas): zeros, no arithmetic
  if (x + 8 <= W && i8 + 8 <= total) {
    const float by0 = boxes[(size_t)k * 4 + 1], by1 = boxes[(size_t)k * 4 + 3];
    const float bx0 = boxes[(size_t)k * 4 + 0], bx1 = boxes[(size_t)k * 4 + 2];
    const float iy_ = paste_axis_coord((float)y, by0, by1, P);
    bool empty = !(iy_ > -1.f && iy_ < (float)P);
    if (!empty) {
      const float ixa = paste_axis_coord((float)x, bx0, bx1, P), ixb = paste_axis_coord((float)(x + 7), bx0, bx1, P);
      empty = !(fmaxf(ixa, ixb) > -1.f && fminf(ixa, ixb) < (float)P);
    }
    if (empty) {
      *reinterpret_cast<unsigned long long*>(out + i8) = 0ull;
      return;
    }
  }
  unsigned long long bits = 0ull;
  int cur_k = -1, cur_y = -1;
  float x0 = 0.f, y0 = 0.f, x1 = 0.f, y1 = 0.f, iy = 0.f, wy_n = 0.f, wy_s = 0.f;
  int yn = 0;
  bool row_in = false;
  const float* pk = probs;
  const int nvalid = (int)((total - i8) < 8 ? (total - i8) : 8);
  for (int e = 0; e < nvalid; ++e) {
    if (k != cur_k) {
      cur_k = k; cur_y = -1;
      x0 = boxes[(size_t)k * 4 + 0]; y0 = boxes[(size_t)k * 4 + 1]; x1 = boxes[(size_t)k * 4 + 2]; y1 = boxes[(size_t)k * 4 + 3];
      pk = probs + (size_t)k * P * P;
    }
    if (y != cur_y) {
      cur_y = y;
      iy = paste_axis_coord((float)y, y0, y1, P);
      const float fy = floorf(iy);
      yn = (int)fy;
      wy_s = iy - fy;          // weight of row yn + 1
      wy_n = (fy + 1.f) - iy;  // weight of row yn
      row_in = iy > -1.f && iy < (float)P;
    }
    bool on = false;
    if (row_in) {
      const float ix = paste_axis_coord((float)x, x0, x1, P);
      if (ix > -1.f && ix < (float)P) {
        const float fx = floorf(ix);
        const int xw = (int)fx;
        const float wx_e = ix - fx, wx_w = (fx + 1.f) - ix;
        const bool yn_ok = yn >= 0 && yn < P, ys_ok = yn + 1 >= 0 && yn + 1 < P;
        const bool xw_ok = xw >= 0 && xw < P, xe_ok = xw + 1 >= 0 && xw + 1 < P;
        float v = 0.f;
        if (yn_ok && xw_ok) v += pk[yn * P + xw] * (wx_w * wy_n);
        if (yn_ok && xe_ok) v += pk[yn * P + xw + 1] * (wx_e * wy_n);
        if (ys_ok && xw_ok) v += pk[(yn + 1) * P + xw] * (wx_w * wy_s);
        if (ys_ok && xe_ok) v += pk[(yn + 1) * P + xw + 1] * (wx_e * wy_s);
        on = v >= thr;
      }
    }
    if (on) bits |= 1ull << (8 * e);
    if (++x == W) { x = 0; if (++y == H) { y = 0; ++k; } }
  }
  if (nvalid == 8) {
    *reinterpret_cast<unsigned long long*>(out + i8) = bits;
  } else {
    for (int e = 0; e < nvalid; ++e) out[i8 + e] = (uint8_t)((bits >> (8 * e)) & 1ull);
  }
}

__global__ __launch_bounds__(256) void paste_masks_kernel(const float* __restrict__ probs, const float* __restrict__ boxes,
                                                          uint8_t* __restrict__ out, long long total, int P, int H, int W,
                                                          float thr) {
  paste_masks_block(probs, boxes, out, total, P, H, W, thr, (long long)blockIdx.x);
}

// The masks of a whole batch of images in ONE launch (detector_postprocess pastes per image: 32 launches per 32-image batch,
// each with its own canvas size): blockIdx.y = image, the image's masks are rows [first, first + n) of probs / boxes and its
// canvases start at byte out_offset of `out`.
constexpr int PASTE_MAXIMG = 64;
struct PasteBatch { U2PasteImage im[PASTE_MAXIMG]; };
__global__ __launch_bounds__(256) void paste_masks_batch_kernel(const PasteBatch batch, const float* __restrict__ probs,
                                                                const float* __restrict__ boxes, uint8_t* __restrict__ out, int P,
                                                                float thr) {
  const U2PasteImage im = batch.im[blockIdx.y];
  const long long total = (long long)im.n * im.H * im.W;
  if ((long long)blockIdx.x * 2048 >= total) return;
  paste_masks_block(probs + (size_t)im.first * P * P, boxes + (size_t)im.first * 4, out + im.out_offset, total, P, im.H, im.W, thr,
                    (long long)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Panoptic merge (panoptic_fpn.py:184-269): one work-group per image walks the instances in descending score order.
// Per instance it counts, inside the region the pasted mask can occupy, the mask area and its overlap with what is
// already claimed; rejects it when overlap / area > overlap_thr (double arithmetic, as the reference's Python floats);
// otherwise claims the still-free mask pixels for the next segment id.  Stuff: areas of the free pixels per semantic
// label by an LDS histogram, ids handed out in ascending label order (torch.unique is sorted), one more pass to write.
// Every decision is integer / exact, so the map is bit-identical to the reference's.
// ---------------------------------------------------------------------------------------------
constexpr int PM_THREADS = 1024, PM_MAXSEM = 256, PM_MAXIMG = 40, PM_STRIPES = 16;
struct PanopticBatch { U2PanopticImage im[PM_MAXIMG]; };   // 40 x 80 bytes of kernel arguments: a 32-image batch in one launch

__device__ __forceinline__ void block_sum2_1024(int& a, int& b, int* red /*[34]*/) {
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) { red[wid * 2] = a; red[wid * 2 + 1] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int sa = 0, sb = 0;
    for (int i = 0; i < PM_THREADS / 64; ++i) { sa += red[i * 2]; sb += red[i * 2 + 1]; }
    red[32] = sa; red[33] = sb;
  }
  __syncthreads();
  a = red[32]; b = red[33];
}

// The merge of one image is serial in the instances (each claims what the better ones left), so that part is one work-group per
// image; the passes over the whole 800 x 1333 map around it - clearing it, the semantic histogram, writing the stuff ids - are
// plain data-parallel and run as PM_STRIPES work-groups per image (one work-group streaming 1 M pixels three times was 2 of the
// 2.8 ms the one-kernel form took per launch; and a 32-image batch was two launches of 24 + 8 images, one after the other).
__global__ __launch_bounds__(PM_THREADS) void panoptic_clear_kernel(const PanopticBatch batch) {
  const U2PanopticImage im = batch.im[blockIdx.y];
  const long long hw = (long long)im.H * im.W;
  for (long long p = (long long)blockIdx.x * PM_THREADS + threadIdx.x; p < hw; p += (long long)gridDim.x * PM_THREADS) im.panoptic[p] = 0;
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < im.num_sem; i += PM_THREADS) { im.stuff_area[i] = 0; im.stuff_segment[i] = 0; }
}

__global__ __launch_bounds__(PM_THREADS) void panoptic_merge_kernel(const PanopticBatch batch, float overlap_thr, float score_thr,
                                                                    int mask_res) {
  __shared__ int red[34];
  const U2PanopticImage im = batch.im[blockIdx.x];
  const int tid = threadIdx.x;
  const int H = im.H, W = im.W;
  const long long hw = (long long)H * W;
  int seg = 0;
  for (int rank = 0; rank < im.K; ++rank) {
    if (tid == 0) im.inst_segment[rank] = 0;
    if (im.scores_sorted[rank] < score_thr) {  // sorted: nothing after this one qualifies either
      for (int r = rank + 1 + tid; r < im.K; r += PM_THREADS) im.inst_segment[r] = 0;
      break;
    }
    const int inst = im.order[rank];
    int ry0 = 0, ry1 = H, rx0 = 0, rx1 = W;
    if (im.boxes && mask_res > 0) {  // a pasted mask reaches at most half a source texel (+ rounding slack) past its box
      const float x0 = im.boxes[inst * 4 + 0], y0 = im.boxes[inst * 4 + 1], x1 = im.boxes[inst * 4 + 2], y1 = im.boxes[inst * 4 + 3];
      const float ex = 0.5f * (x1 - x0) / (float)mask_res + 2.f, ey = 0.5f * (y1 - y0) / (float)mask_res + 2.f;
      rx0 = max(0, (int)floorf(x0 - ex)); rx1 = min(W, (int)ceilf(x1 + ex) + 1);
      ry0 = max(0, (int)floorf(y0 - ey)); ry1 = min(H, (int)ceilf(y1 + ey) + 1);
    }
    const int rw = max(rx1 - rx0, 0), rh = max(ry1 - ry0, 0);
    const unsigned char* m = im.masks + (size_t)inst * hw;
    int area = 0, inter = 0;
    for (int i = tid; i < rw * rh; i += PM_THREADS) {
      const int y = ry0 + i / rw, x = rx0 + i % rw;
      const size_t p = (size_t)y * W + x;
      if (m[p]) { ++area; if (im.panoptic[p] > 0) ++inter; }
    }
    block_sum2_1024(area, inter, red);
    if (area == 0) continue;
    if ((double)inter * 1.0 / (double)area > (double)overlap_thr) continue;
    ++seg;
    for (int i = tid; i < rw * rh; i += PM_THREADS) {
      const int y = ry0 + i / rw, x = rx0 + i % rw;
      const size_t p = (size_t)y * W + x;
      if (m[p] && im.panoptic[p] == 0) im.panoptic[p] = seg;
    }
    if (tid == 0) im.inst_segment[rank] = seg;
    __syncthreads();
  }
}

// semantic histogram of a stripe: per label the free pixels (-> stuff_area) and whether the label occurs at all (-> a flag in
// stuff_segment, replaced by the segment id in the last kernel)
__global__ __launch_bounds__(PM_THREADS) void panoptic_stuff_hist_kernel(const PanopticBatch batch) {
  __shared__ int hist_all[PM_MAXSEM], hist_free[PM_MAXSEM];
  const U2PanopticImage im = batch.im[blockIdx.y];
  const long long hw = (long long)im.H * im.W;
  for (int i = threadIdx.x; i < PM_MAXSEM; i += PM_THREADS) { hist_all[i] = 0; hist_free[i] = 0; }
  __syncthreads();
  for (long long p = (long long)blockIdx.x * PM_THREADS + threadIdx.x; p < hw; p += (long long)gridDim.x * PM_THREADS) {
    const long long lab = im.semantic[p];
    if (lab >= 0 && lab < im.num_sem) {
      atomicAdd(&hist_all[(int)lab], 1);
      if (im.panoptic[p] == 0) atomicAdd(&hist_free[(int)lab], 1);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < im.num_sem; i += PM_THREADS) {
    if (hist_free[i]) atomicAdd(&im.stuff_area[i], hist_free[i]);
    if (hist_all[i]) atomicOr(&im.stuff_segment[i], 1);
  }
}

// ids in ascending label order (torch.unique is sorted) after the instance segments; FILL: write them into the stripe,
// otherwise (one work-group per image, after the fill) into stuff_segment
template <bool FILL>
__global__ __launch_bounds__(PM_THREADS) void panoptic_stuff_ids_kernel(const PanopticBatch batch, int stuff_area_thr) {
  __shared__ int stuff_id[PM_MAXSEM];
  const U2PanopticImage im = batch.im[blockIdx.y];
  const long long hw = (long long)im.H * im.W;
  if (threadIdx.x == 0) {
    int seg = 0;
    for (int r = 0; r < im.K; ++r) seg = max(seg, im.inst_segment[r]);
    stuff_id[0] = 0;
    for (int lab = 1; lab < im.num_sem; ++lab) {  // label 0 = "thing" pixels of the semantic head
      int id = 0;
      if (im.stuff_segment[lab] != 0 && im.stuff_area[lab] >= stuff_area_thr) id = ++seg;
      stuff_id[lab] = id;
    }
  }
  __syncthreads();
  if (FILL) {
    for (long long p = (long long)blockIdx.x * PM_THREADS + threadIdx.x; p < hw; p += (long long)gridDim.x * PM_THREADS) {
      const long long lab = im.semantic[p];
      if (lab > 0 && lab < im.num_sem && im.panoptic[p] == 0) {
        const int id = stuff_id[(int)lab];
        if (id) im.panoptic[p] = id;
      }
    }
  } else {
    for (int i = threadIdx.x; i < im.num_sem; i += PM_THREADS) im.stuff_segment[i] = stuff_id[i];
  }
}

}  // namespace

extern "C" int u2_panoptic_merge(const U2PanopticImage* images, int n_images, float overlap_thr, int stuff_area_thr,
                                 float score_thr, int mask_res, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  for (int b0 = 0; b0 < n_images; b0 += PM_MAXIMG) {
    const int nb = n_images - b0 < PM_MAXIMG ? n_images - b0 : PM_MAXIMG;
    PanopticBatch batch;
    for (int i = 0; i < nb; ++i) {
      batch.im[i] = images[b0 + i];
      if (batch.im[i].num_sem > PM_MAXSEM || batch.im[i].num_sem < 0 || batch.im[i].H <= 0 || batch.im[i].W <= 0) return -1;
    }
    hipLaunchKernelGGL(panoptic_clear_kernel, dim3(PM_STRIPES, nb), dim3(PM_THREADS), 0, s, batch);
    U2_CHECK_LAUNCH();
    hipLaunchKernelGGL(panoptic_merge_kernel, dim3(nb), dim3(PM_THREADS), 0, s, batch, overlap_thr, score_thr, mask_res);
    U2_CHECK_LAUNCH();
    hipLaunchKernelGGL(panoptic_stuff_hist_kernel, dim3(PM_STRIPES, nb), dim3(PM_THREADS), 0, s, batch);
    U2_CHECK_LAUNCH();
    hipLaunchKernelGGL(panoptic_stuff_ids_kernel<true>, dim3(PM_STRIPES, nb), dim3(PM_THREADS), 0, s, batch, stuff_area_thr);
    U2_CHECK_LAUNCH();
    hipLaunchKernelGGL(panoptic_stuff_ids_kernel<false>, dim3(1, nb), dim3(PM_THREADS), 0, s, batch, stuff_area_thr);
    U2_CHECK_LAUNCH();
  }
  return 0;
}


namespace {
// ------------------------------------------------------------------------------------------------
// Semantic head at inference: F.interpolate(logits.float(), scale_factor = S, mode = "bilinear", align_corners = False)
// (meta_arch/semantic_seg.py:240-244) and the per-pixel argmax PanopticFPN.inference takes of it (panoptic_fpn.py:173), in one
// pass over the low-resolution NHWC bf16 logits: the S-times upsampled fp32 NCHW logits are written once (the caller's
// "sem_seg" result) and the argmax map with them, so the 3.9 GB of a batch-32 result are never read back.
// Arithmetic = ATen's upsample_bilinear2d: src = max((dst + 0.5) / S - 0.5, 0); i0 = floor(src); i1 = i0 + (i0 < n - 1);
// value = l0y * (l0x * v00 + l1x * v01) + l1y * (l0x * v10 + l1x * v11), fp32, contraction off.
// ------------------------------------------------------------------------------------------------
template <int KMAX>
__global__ __launch_bounds__(256) void semseg_upsample_kernel(const bf16_t* __restrict__ x, float* __restrict__ out,
                                                              long long* __restrict__ amax, int B, int H, int W, int Cp, int K,
                                                              int S) {
  const int Ho = H * S, Wo = W * S;
  const float inv = 1.0f / (float)S;
  const size_t total = (size_t)B * Ho * Wo;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ox = (int)(i % Wo);
    const size_t t = i / Wo;
    const int oy = (int)(t % Ho), b = (int)(t / Ho);
    const float sy = fmaxf(inv * ((float)oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(inv * ((float)ox + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const bf16_t* base = x + (size_t)b * H * W * Cp;
    const bf16_t* p00 = base + ((size_t)y0 * W + x0) * Cp;
    const bf16_t* p01 = base + ((size_t)y0 * W + x1) * Cp;
    const bf16_t* p10 = base + ((size_t)y1 * W + x0) * Cp;
    const bf16_t* p11 = base + ((size_t)y1 * W + x1) * Cp;
    float best = 0.f;
    int besti = 0;
#pragma unroll
    for (int c8 = 0; c8 < KMAX; c8 += 8) {
      if (c8 >= K) break;
      bf16_t v00[8], v01[8], v10[8], v11[8];
      *reinterpret_cast<uint4*>(v00) = *reinterpret_cast<const uint4*>(p00 + c8);
      *reinterpret_cast<uint4*>(v01) = *reinterpret_cast<const uint4*>(p01 + c8);
      *reinterpret_cast<uint4*>(v10) = *reinterpret_cast<const uint4*>(p10 + c8);
      *reinterpret_cast<uint4*>(v11) = *reinterpret_cast<const uint4*>(p11 + c8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = c8 + e;
        if (c >= K) break;
        const float f = hy * (hx * bf2f(v00[e]) + lx * bf2f(v01[e])) + ly * (hx * bf2f(v10[e]) + lx * bf2f(v11[e]));
        if (out) out[(((size_t)b * K + c) * Ho + oy) * Wo + ox] = f;
        if (c == 0 || f > best) { best = f; besti = c; }  // first maximum wins, like torch.argmax
      }
    }
    if (amax) amax[i] = besti;
  }
}

// sem_seg_postprocess (modeling/postprocessing.py:77-100): bilinear resize of a cropped fp32 [C][Hin][Win] window to
// [C][Hout][Wout], align_corners = False, ATen's source-index rule (fp32 scale = in / out; negative source clamped to 0)
__global__ __launch_bounds__(256) void bilinear_resize_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int C,
                                                                  int Hin, int Win, long long in_cs, long long in_rs, int Ho, int Wo,
                                                                  float sch, float scw) {
  const size_t total = (size_t)C * Ho * Wo;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ox = (int)(i % Wo);
    const size_t t = i / Wo;
    const int oy = (int)(t % Ho), c = (int)(t / Ho);
    const float sy = fmaxf(sch * ((float)oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(scw * ((float)ox + 0.5f) - 0.5f, 0.f);
    const int y0 = min((int)sy, Hin - 1), x0 = min((int)sx, Win - 1);
    const int y1 = y0 + (y0 < Hin - 1 ? 1 : 0), x1 = x0 + (x0 < Win - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* base = in + (size_t)c * in_cs;
    const float v00 = base[(size_t)y0 * in_rs + x0], v01 = base[(size_t)y0 * in_rs + x1];
    const float v10 = base[(size_t)y1 * in_rs + x0], v11 = base[(size_t)y1 * in_rs + x1];
    out[i] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
  }
}

}  // namespace

extern "C" int u2_semseg_upsample(const void* logits, float* out, long long* argmax, int B, int H, int W, int Cp, int K, int S,
                                  void* stream) {
  if ((Cp & 7) || K < 1 || K > Cp || K > 64 || S < 1) return -1;
  const size_t total = (size_t)B * H * S * W * S;
  if (!total) return 0;
  size_t g = (total + 255) / 256;
  if (g > 256 * 32) g = 256 * 32;
  hipLaunchKernelGGL(semseg_upsample_kernel<64>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits, out,
                     argmax, B, H, W, Cp, K, S);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_paste_masks_batch(const float* probs, const float* boxes, void* out, const U2PasteImage* images, int num_images,
                                    int P, float threshold, void* stream) {
  if (num_images <= 0) return 0;
  if (P < 1 || !images) return -1;
  for (int i0 = 0; i0 < num_images; i0 += PASTE_MAXIMG) {
    PasteBatch b;
    const int nb = num_images - i0 < PASTE_MAXIMG ? num_images - i0 : PASTE_MAXIMG;
    long long max_blocks = 0;
    for (int i = 0; i < nb; ++i) {
      b.im[i] = images[i0 + i];
      if (b.im[i].n < 0 || b.im[i].H < 0 || b.im[i].W < 0 || (b.im[i].out_offset & 7)) return -1;
      const long long blocks = ((long long)b.im[i].n * b.im[i].H * b.im[i].W + 2047) / 2048;
      if (blocks > max_blocks) max_blocks = blocks;
    }
    if (max_blocks == 0) continue;
    if (max_blocks > 0x7fffffffLL) return -1;
    hipLaunchKernelGGL(paste_masks_batch_kernel, dim3((unsigned)max_blocks, (unsigned)nb), dim3(256), 0, (hipStream_t)stream, b, probs,
                       boxes, (uint8_t*)out, P, threshold);
    U2_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int u2_paste_masks(const float* probs, const float* boxes, void* out, int n, int P, int H, int W, float threshold,
                              void* stream) {
  if (n <= 0 || H <= 0 || W <= 0) return 0;
  if (P < 1) return -1;
  const long long total = (long long)n * H * W;
  const long long threads = (total + 7) / 8;
  const long long blocks = (threads + 255) / 256;
  if (blocks > 0x7fffffffLL) return -1;
  hipLaunchKernelGGL(paste_masks_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, probs, boxes,
                     (uint8_t*)out, total, P, H, W, threshold);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_bilinear_resize_f32(const float* in, float* out, int C, int Hin, int Win, long long in_cs, long long in_rs,
                                      int Hout, int Wout, void* stream) {
  if (C < 1 || Hin < 1 || Win < 1 || Hout < 1 || Wout < 1) return -1;
  const size_t total = (size_t)C * Hout * Wout;
  size_t g = (total + 255) / 256;
  if (g > 256 * 32) g = 256 * 32;
  hipLaunchKernelGGL(bilinear_resize_f32_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, in, out, C, Hin, Win, in_cs,
                     in_rs, Hout, Wout, (float)Hin / (float)Hout, (float)Win / (float)Wout);
  U2_CHECK_LAUNCH();
  return 0;
}
