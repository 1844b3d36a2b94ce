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

// The canvases ([n][H][W] bytes per image, 3.4 GB for a 32-image batch of 100 masks at 800 x 1333) are zero except inside the
// boxes (~4 % of the pixels).  Written in two passes on the same stream:
//   1. paste_zero_kernel: plain 16-byte non-temporal zero fill of everything (6+ TB/s);
//   2. paste_interior_kernel: per mask, the rows and the 8-byte words of each row that the box can reach (a conservative
//      rectangle, two pixels wider than the exact condition) are computed pixel by pixel with the reference's arithmetic.
// One pass over everything (every thread 8 or 16 bytes, zero store where its bytes miss the box) ran at 1.35 TB/s: a wave covers
// most of a canvas row, so every wave on a row the box touches sat in the per-pixel loop with a fifth of its lanes busy
// (profiles/r04_paste.txt).
typedef unsigned long long paste_u64x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void paste_zero_kernel(uint8_t* __restrict__ out, long long n16) {
  paste_u64x2* o = reinterpret_cast<paste_u64x2*>(out);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256)
    __builtin_nontemporal_store(paste_u64x2{0ull, 0ull}, o + i);
}

// the 8 bytes at flat index i = (k, y, x) of an image's canvases, exactly as the reference computes them
__device__ __forceinline__ void paste_group8(const float* __restrict__ probs, const float* __restrict__ boxes,
                                             uint8_t* __restrict__ out, long long total, int P, int H, int W, float thr, int k,
                                             int y, int x, long long i) {
  if (x + 8 <= W && i + 8 <= total) {
    // eight pixels of one row (nearly every group): the row terms once, the 32 source reads of the group issued together
    // (clamped addresses, a term outside the P x P map contributes an exact 0 - the sum is the one the loop below forms)
    const float bx0 = boxes[(size_t)k * 4 + 0], by0 = boxes[(size_t)k * 4 + 1], bx1 = boxes[(size_t)k * 4 + 2], by1 = boxes[(size_t)k * 4 + 3];
    const float* pk = probs + (size_t)k * P * P;
    const float iy = paste_axis_coord((float)y, by0, by1, P);
    const float fy = floorf(iy);
    unsigned long long bits = 0ull;
    if (iy > -1.f && iy < (float)P) {
      const int yn = (int)fy;
      const float wy_s = iy - fy, wy_n = (fy + 1.f) - iy;
      const bool yn_ok = yn >= 0 && yn < P, ys_ok = yn + 1 >= 0 && yn + 1 < P;
      const float* rn = pk + min(max(yn, 0), P - 1) * P;
      const float* rs = pk + min(max(yn + 1, 0), P - 1) * P;
      float v[8];
      bool in[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float ix = paste_axis_coord((float)(x + e), bx0, bx1, P);
        in[e] = ix > -1.f && ix < (float)P;
        const float fx = floorf(in[e] ? ix : 0.f);
        const int xw = (int)fx;
        const float wx_e = ix - fx, wx_w = (fx + 1.f) - ix;
        const bool xw_ok = xw >= 0 && xw < P, xe_ok = xw + 1 >= 0 && xw + 1 < P;
        const int cw = min(max(xw, 0), P - 1), ce = min(max(xw + 1, 0), P - 1);
        const float p00 = rn[cw], p01 = rn[ce], p10 = rs[cw], p11 = rs[ce];
        float a = 0.f;
        a += (yn_ok && xw_ok) ? p00 * (wx_w * wy_n) : 0.f;
        a += (yn_ok && xe_ok) ? p01 * (wx_e * wy_n) : 0.f;
        a += (ys_ok && xw_ok) ? p10 * (wx_w * wy_s) : 0.f;
        a += (ys_ok && xe_ok) ? p11 * (wx_e * wy_s) : 0.f;
        v[e] = a;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (in[e] && v[e] >= thr) bits |= 1ull << (8 * e);
    }
    *reinterpret_cast<unsigned long long*>(out + i) = bits;
    return;
  }
  unsigned long long bits = 0ull;
  int cur_k = -1, cur_y = -1;
  float x0 = 0.f, y0 = 0.f, x1 = 0.f, y1 = 0.f, iy = 0.f, wy_n = 0.f, wy_s = 0.f;
  int yn = 0;
  bool row_in = false;
  const float* pk = probs;
  const int nvalid = (int)((total - i) < 8 ? (total - i) : 8);
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
    *reinterpret_cast<unsigned long long*>(out + i) = bits;
  } else {
    for (int e = 0; e < nvalid; ++e) out[i + e] = (uint8_t)((bits >> (8 * e)) & 1ull);
  }
}

// pixels [lo, hi] of an axis of extent n that a box side (b0, b1) can light: ix > -1 and ix < P hold only for
// b_lo - 0.5 bw / P - 0.5 < pix < b_hi + 0.5 bw / P - 0.5; two pixels of margin on either side absorb the rounding of this form
// (the pixels inside are decided by the exact expression).  A NaN side gives the whole axis.
__device__ __forceinline__ void paste_reach(float b0, float b1, int P, int n, int& lo, int& hi) {
  const float bw = fabsf(b1 - b0);
  const float flo = fminf(b0, b1) - 0.5f * bw / (float)P - 2.5f, fhi = fmaxf(b0, b1) + 0.5f * bw / (float)P + 1.5f;
  lo = (int)fminf(fmaxf(floorf(flo), 0.f), (float)n);           // n: an empty range when the box lies beyond the axis
  hi = (int)fmaxf(fminf(ceilf(fhi), (float)(n - 1)), -1.f);
}

constexpr int PASTE_MAXIMG = 64, PASTE_ROW_GROUPS = 16;
struct PasteBatch { U2PasteImage im[PASTE_MAXIMG]; };
// grid (PASTE_ROW_GROUPS, masks of the largest image, images)
__global__ __launch_bounds__(256) void paste_interior_kernel(const PasteBatch batch, const float* __restrict__ probs,
                                                             const float* __restrict__ boxes, uint8_t* __restrict__ out, int P,
                                                             float thr) {
  const U2PasteImage im = batch.im[blockIdx.z];
  const int k = blockIdx.y;
  if (k >= im.n) return;
  const int H = im.H, W = im.W;
  const float* bimg = boxes + (size_t)im.first * 4;
  const float* pimg = probs + (size_t)im.first * P * P;
  int xlo, xhi, ylo, yhi;
  paste_reach(bimg[(size_t)k * 4 + 0], bimg[(size_t)k * 4 + 2], P, W, xlo, xhi);
  paste_reach(bimg[(size_t)k * 4 + 1], bimg[(size_t)k * 4 + 3], P, H, ylo, yhi);
  if (xlo > xhi || ylo > yhi) return;
  const long long hw = (long long)H * W, total = (long long)im.n * hw;
  uint8_t* o = out + im.out_offset;
  // items = (row, 8-byte word of the row's span); a box is ~30 words wide, so the rows of a work-group are walked together
  const int wpr = (xhi - xlo) / 8 + 2;                                    // words per row, at most
  const int nrows = yhi - ylo + 1;
  const int rpg = (nrows + PASTE_ROW_GROUPS - 1) / PASTE_ROW_GROUPS;      // rows of this work-group: [r0, r1)
  const int r0 = (int)blockIdx.x * rpg, r1 = min(nrows, r0 + rpg);
  for (int t = (int)threadIdx.x; t < (r1 - r0) * wpr; t += 256) {
    const int y = ylo + r0 + t / wpr, wd = t % wpr;
    const long long rowbase = (long long)k * hw + (long long)y * W;
    const long long a = ((rowbase + xlo) & ~7LL) + 8LL * wd;               // the image's canvases start 16-byte aligned: a >= 0
    if (a > rowbase + xhi) continue;
    int dx = (int)(a - rowbase), yy = y, kk = k;
    while (dx < 0) {  // the word begins in the previous row (or the previous mask's last row)
      dx += W;
      if (--yy < 0) { yy = H - 1; --kk; }
    }
    paste_group8(pimg, bimg, o, total, P, H, W, thr, kk, yy, dx, a);
  }
}

static int paste_launch(const PasteBatch& b, int nb, const float* probs, const float* boxes, uint8_t* out, int P, float thr,
                        hipStream_t s) {
  int max_n = 0;
  long long lo = -1, hi = 0, sum = 0;
  for (int i = 0; i < nb; ++i) {
    const long long bytes = (long long)b.im[i].n * b.im[i].H * b.im[i].W;
    if (bytes <= 0) continue;
    if (b.im[i].n > max_n) max_n = b.im[i].n;
    if (lo < 0 || b.im[i].out_offset < lo) lo = b.im[i].out_offset;
    if (b.im[i].out_offset + bytes > hi) hi = b.im[i].out_offset + bytes;
    sum += bytes;
  }
  if (max_n == 0) return 0;
  // zero fill.  Canvases laid out back to back (apart from the alignment padding between images, which is zeroed with them):
  // one launch over the whole span; otherwise image by image.  Whole 16-byte words by the kernel, a ragged end by a memset.
  auto zero = [&](long long off, long long bytes) {
    const long long n16 = bytes / 16;
    if (n16 > 0) {
      long long g = (n16 + 255) / 256;
      if (g > 256 * 64) g = 256 * 64;
      hipLaunchKernelGGL(paste_zero_kernel, dim3((unsigned)g), dim3(256), 0, s, out + off, n16);
    }
    if (bytes & 15) (void)hipMemsetAsync(out + off + n16 * 16, 0, (size_t)(bytes & 15), s);
  };
  if (hi - lo <= sum + 16LL * nb) {
    zero(lo, hi - lo);
  } else {
    for (int i = 0; i < nb; ++i) {
      const long long bytes = (long long)b.im[i].n * b.im[i].H * b.im[i].W;
      if (bytes > 0) zero(b.im[i].out_offset, bytes);
    }
  }
  if (max_n > 65535) return -1;
  hipLaunchKernelGGL(paste_interior_kernel, dim3(PASTE_ROW_GROUPS, (unsigned)max_n, (unsigned)nb), dim3(256), 0, s, b, probs, boxes,
                     out, P, thr);
  return 0;
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

// semantic label of pixel p = y * W + x: the label map may be a window of a wider (padded) map, row pitch sem_stride elements
__device__ __forceinline__ long long panoptic_sem_at(const U2PanopticImage& im, long long p) {
  if (im.sem_stride == im.W) return im.semantic[p];
  const int y = (int)(p / im.W);
  return im.semantic[(long long)y * im.sem_stride + (p - (long long)y * im.W)];
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
    const long long lab = panoptic_sem_at(im, p);
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
      const long long lab = panoptic_sem_at(im, p);
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
      if (batch.im[i].num_sem > PM_MAXSEM || batch.im[i].num_sem < 0 || batch.im[i].H <= 0 || batch.im[i].W <= 0 ||
          batch.im[i].sem_stride < batch.im[i].W)
        return -1;
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
        // (non-temporal: the 3.9 GB result is far beyond the MALL and not read again on the device, 1.42 -> 1.33 ms at batch 32;
        //  four pixels per thread with 16-byte stores measured slower, 1.59 ms: 64 source loads per thread)
        if (out) __builtin_nontemporal_store(f, out + (((size_t)b * K + c) * Ho + oy) * Wo + ox);
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

// ---------------------------------------------------------------------------------------------
// mask_rcnn_inference (roi_heads/mask_head.py:115-158) without the K-channel logit map: the reference runs the 1x1 predictor to
// all K classes (800 here: 4 GB of logits for the 3200 detections of a 32-image batch) and keeps one channel per detection.
// Here only that channel is formed:  prob[n][y][x] = sigmoid(bf16(x[n][y][x][:] . bf16(Wp[cls[n]][:]) + bf16(bp[cls[n]]))),
// products of bf16 operands summed in fp32 and the logit rounded to bf16 as the conv's output is.
// `phased`: x is the 2x2/stride-2 deconvolution's GEMM output before its pixel shuffle, [N][S][S][2 (dy)][2 (dx)][C] - output
// pixel (2 h + dy, 2 w + dx) reads phase (dy, dx) of source pixel (h, w) - so the shuffled copy of the trunk output is never made.
// One work-group per detection; a wave reads two positions per load, 8 channels per lane and 256-channel chunk (C % 8 == 0, <= 1024).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_predict_prob_kernel(const bf16_t* __restrict__ x, const float* __restrict__ Wp,
                                                                const float* __restrict__ bp, const long long* __restrict__ cls,
                                                                float* __restrict__ prob, int S2, int C, int phased) {
  constexpr int MAXCH = 4;   // 256-channel chunks a lane pair row covers: C <= 1024
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int half = lane >> 5, l32 = lane & 31;
  const int k = (int)cls[n];
  float wq[MAXCH][8];
#pragma unroll
  for (int q = 0; q < MAXCH; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = q * 256 + l32 * 8 + e;
      wq[q][e] = c < C ? bf2f(f2bf(Wp[(size_t)k * C + c])) : 0.f;
    }
  const float bias = bf2f(f2bf(bp[k]));
  const int P = S2 * S2;
  const bf16_t* xn = x + (size_t)n * P * C;
  for (int p0 = wv * 2; p0 < P; p0 += 8) {   // positions in memory order (for `phased`: (h, w, dy, dx))
    const int p = p0 + half;
    float d = 0.f;
    if (p < P) {
#pragma unroll
      for (int q = 0; q < MAXCH; ++q) {
        const int c = q * 256 + l32 * 8;
        if (c < C) {
          bf16_t xv[8];
          *reinterpret_cast<uint4*>(xv) = *reinterpret_cast<const uint4*>(xn + (size_t)p * C + c);
#pragma unroll
          for (int e = 0; e < 8; ++e) d += bf2f(xv[e]) * wq[q][e];
        }
      }
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
    if (l32 == 0 && p < P) {
      const float z = bf2f(f2bf(d + bias));
      int o = p;
      if (phased) {
        const int S = S2 >> 1;
        const int dx = p & 1, dy = (p >> 1) & 1, hwi = p >> 2;
        const int h = hwi / S, w = hwi - h * S;
        o = (2 * h + dy) * S2 + 2 * w + dx;
      }
      prob[(size_t)n * P + o] = 1.f / (1.f + expf(-z));
    }
  }
}

extern "C" int u2_mask_predict_prob(const void* x, const float* Wp, const float* bp, const void* cls, float* prob, int N, int S2,
                                    int C, int phased, void* stream) {
  if (C < 8 || (C & 7) || C > 1024 || S2 < 1 || (phased && (S2 & 1))) return -1;
  if (N <= 0) return 0;
  hipLaunchKernelGGL(mask_predict_prob_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, Wp, bp,
                     (const long long*)cls, prob, S2, C, phased);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_paste_masks_batch(const float* probs, const float* boxes, void* out, const U2PasteImage* images, int num_images,
                                    int P, float threshold, void* stream) {
  if (num_images <= 0) return 0;
  if (P < 1 || !images || ((uintptr_t)out & 15)) return -1;
  for (int i = 0; i < num_images; ++i)
    if (images[i].n < 0 || images[i].H < 0 || images[i].W < 0 || (images[i].out_offset & 15)) return -1;
  for (int i0 = 0; i0 < num_images; i0 += PASTE_MAXIMG) {
    PasteBatch b;
    const int nb = num_images - i0 < PASTE_MAXIMG ? num_images - i0 : PASTE_MAXIMG;
    for (int i = 0; i < nb; ++i) b.im[i] = images[i0 + i];
    if (paste_launch(b, nb, probs, boxes, (uint8_t*)out, P, threshold, (hipStream_t)stream)) return -1;
    U2_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int u2_paste_masks(const float* probs, const float* boxes, void* out, int n, int P, int H, int W, float threshold,
                              void* stream) {
  if (n <= 0 || H <= 0 || W <= 0) return 0;
  if (P < 1 || ((uintptr_t)out & 15)) return -1;
  PasteBatch b;
  b.im[0].first = 0; b.im[0].n = n; b.im[0].H = H; b.im[0].W = W; b.im[0].out_offset = 0;
  if (paste_launch(b, 1, probs, boxes, (uint8_t*)out, P, threshold, (hipStream_t)stream)) return -1;
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
