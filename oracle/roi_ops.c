/* CPU oracle (TEST INFRASTRUCTURE ONLY - never linked or called by the product path).
 *
 * Plain-C restatement of the two torchvision ops the reference's hot path bottoms out in:
 *   - roi_align(aligned=True, sampling_ratio=0) forward/backward: call sites detectron2/layers/roi_align.py:58-65
 *     (via modeling/poolers.py:165-176,261 and structures/masks.py:211-215). torchvision is an un-vendored,
 *     un-pinned dependency; the arithmetic restated here follows the reference's own vendored equivalent,
 *     detectron2/layers/csrc/ROIAlignRotated/ROIAlignRotated_cpu.cpp:27-129 (bilinear pre-calc rules),
 *     :201-310 (forward) and :312-416 (backward) evaluated at angle 0, and is pinned by the golden 4x4 of
 *     tests/layers/test_roi_align.py:36-41 plus outputs of that vendored C++ op captured in tests/golden/.
 *   - nms / batched_nms: call sites detectron2/layers/nms.py:5-6,20. Rule restated from the python statement
 *     in the reference tests (tests/layers/test_nms_rotated.py:44-66): visit boxes by descending score, keep a
 *     box unless an already kept box overlaps it with IoU > threshold.
 * Layouts: features NCHW fp32, rois [R][5] = (batch index, x0, y0, x1, y1).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int ylo, xlo, yhi, xhi; float w1, w2, w3, w4; int ok; } tap_t;

static tap_t make_tap(float y, float x, int H, int W) {
  tap_t t;
  memset(&t, 0, sizeof(t));
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return t; /* sample contributes nothing */
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  t.ylo = (int)y;
  t.xlo = (int)x;
  if (t.ylo >= H - 1) { t.yhi = t.ylo = H - 1; y = (float)t.ylo; } else t.yhi = t.ylo + 1;
  if (t.xlo >= W - 1) { t.xhi = t.xlo = W - 1; x = (float)t.xlo; } else t.xhi = t.xlo + 1;
  {
    const float ly = y - t.ylo, lx = x - t.xlo, hy = 1.0f - ly, hx = 1.0f - lx;
    t.w1 = hy * hx; t.w2 = hy * lx; t.w3 = ly * hx; t.w4 = ly * lx;
  }
  t.ok = 1;
  return t;
}

/* dir = 0: out[r][c][ph][pw] = mean of samples; dir = 1: dfeat += scatter of dout */
static void roi_align_impl(int dir, const float* feat, float* dfeat, const float* rois, float* out, const float* dout,
                           int R, int C, int H, int W, int PH, int PW, float scale, int sampling_ratio, int aligned) {
  const float off = aligned ? 0.5f : 0.0f;
  for (int r = 0; r < R; ++r) {
    const float* roi = rois + (size_t)r * 5;
    const int b = (int)roi[0];
    const float sw = roi[1] * scale - off, sh = roi[2] * scale - off;
    const float ew = roi[3] * scale - off, eh = roi[4] * scale - off;
    float rw = ew - sw, rh = eh - sh;
    if (!aligned) { if (rw < 1.0f) rw = 1.0f; if (rh < 1.0f) rh = 1.0f; }
    const float bh = rh / (float)PH, bw = rw / (float)PW;
    const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
    const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
    const int cnt_i = gh * gw > 1 ? gh * gw : 1;
    const float cnt = (float)cnt_i;
    for (int ph = 0; ph < PH; ++ph)
      for (int pw = 0; pw < PW; ++pw)
        for (int iy = 0; iy < gh; ++iy) {
          const float y = sh + ph * bh + (iy + 0.5f) * bh / (float)gh;
          for (int ix = 0; ix < gw; ++ix) {
            const float x = sw + pw * bw + (ix + 0.5f) * bw / (float)gw;
            const tap_t t = make_tap(y, x, H, W);
            if (!t.ok) continue;
            for (int c = 0; c < C; ++c) {
              const size_t plane = ((size_t)b * C + c) * H * W;
              const size_t o = (((size_t)r * C + c) * PH + ph) * PW + pw;
              if (dir == 0) {
                const float* f = feat + plane;
                out[o] += t.w1 * f[t.ylo * W + t.xlo] + t.w2 * f[t.ylo * W + t.xhi] + t.w3 * f[t.yhi * W + t.xlo] +
                          t.w4 * f[t.yhi * W + t.xhi];
              } else {
                float* g = dfeat + plane;
                const float d = dout[o] / cnt;
                g[t.ylo * W + t.xlo] += t.w1 * d;
                g[t.ylo * W + t.xhi] += t.w2 * d;
                g[t.yhi * W + t.xlo] += t.w3 * d;
                g[t.yhi * W + t.xhi] += t.w4 * d;
              }
            }
          }
        }
    if (dir == 0) /* output = sum of samples / count (one division per output element) */
      for (size_t o = (size_t)r * C * PH * PW; o < (size_t)(r + 1) * C * PH * PW; ++o) out[o] /= cnt;
  }
}

void oracle_roi_align_fwd(const float* feat, const float* rois, float* out, int R, int C, int H, int W, int PH, int PW,
                          float scale, int sampling_ratio, int aligned) {
  memset(out, 0, sizeof(float) * (size_t)R * C * PH * PW);
  roi_align_impl(0, feat, NULL, rois, out, NULL, R, C, H, W, PH, PW, scale, sampling_ratio, aligned);
}

/* dfeat must be zeroed by the caller ([N][C][H][W]) */
void oracle_roi_align_bwd(const float* dout, const float* rois, float* dfeat, int R, int C, int H, int W, int PH, int PW,
                          float scale, int sampling_ratio, int aligned) {
  roi_align_impl(1, NULL, dfeat, rois, NULL, dout, R, C, H, W, PH, PW, scale, sampling_ratio, aligned);
}

typedef struct { float s; int i; } si_t;
static int cmp_desc(const void* a, const void* b) {
  const si_t* x = (const si_t*)a; const si_t* y = (const si_t*)b;
  if (x->s > y->s) return -1;
  if (x->s < y->s) return 1;
  return x->i - y->i; /* stable: lower index first on equal scores */
}

/* groups may be NULL (plain nms).  Returns the number kept; keep[] = original indices by descending score. */
int oracle_nms(const float* boxes, const float* scores, const int* groups, int n, float thr, int* keep) {
  si_t* ord = (si_t*)malloc(sizeof(si_t) * (size_t)(n > 0 ? n : 1));
  char* dead = (char*)calloc((size_t)(n > 0 ? n : 1), 1);
  int nk = 0;
  for (int i = 0; i < n; ++i) { ord[i].s = scores[i]; ord[i].i = i; }
  qsort(ord, (size_t)n, sizeof(si_t), cmp_desc);
  for (int a = 0; a < n; ++a) {
    if (dead[a]) continue;
    const int i = ord[a].i;
    const float* bi = boxes + (size_t)i * 4;
    const float ai = (bi[2] - bi[0]) * (bi[3] - bi[1]);
    keep[nk++] = i;
    for (int c = a + 1; c < n; ++c) {
      if (dead[c]) continue;
      const int j = ord[c].i;
      if (groups && groups[i] != groups[j]) continue;
      const float* bj = boxes + (size_t)j * 4;
      float w = fminf(bi[2], bj[2]) - fmaxf(bi[0], bj[0]);
      float h = fminf(bi[3], bj[3]) - fmaxf(bi[1], bj[1]);
      if (w < 0) w = 0;
      if (h < 0) h = 0;
      const float inter = w * h;
      const float aj = (bj[2] - bj[0]) * (bj[3] - bj[1]);
      if (inter / (ai + aj - inter) > thr) dead[c] = 1;
    }
  }
  free(ord);
  free(dead);
  return nk;
}
