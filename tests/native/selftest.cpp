// Standalone GPU self-test + micro-benchmark of the GEMM-family kernels through the C ABI
// (no torch dependency, so it starts in a second on a fresh GPU box).  Test infrastructure only.
//   build: u2seg_amd/csrc/build.sh && hipcc tests/native/selftest.cpp -Iinclude -Lu2seg_amd/csrc -lu2seg_hip ...
//   run:   selftest [bench]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#include "u2seg_hip.h"

#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t v) { uint32_t u = ((uint32_t)v) << 16; float f; memcpy(&f, &u, 4); return f; }
static uint32_t rng_state = 12345;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f; }

template <typename T> struct DBuf {
  T* d = nullptr; size_t n = 0;
  explicit DBuf(size_t n_) : n(n_) { HIPCHK(hipMalloc(&d, n * sizeof(T) + 256)); HIPCHK(hipMemset(d, 0, n * sizeof(T) + 256)); }
  ~DBuf() { (void)hipFree(d); }
  void up(const std::vector<T>& h) { HIPCHK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); }
  std::vector<T> down() { std::vector<T> h(n); HIPCHK(hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost)); return h; }
};

struct ConvCase { int B, Hin, Win, C, N, KH, KW, pad, mul, div, relu, accum, bias, stats; const char* name; };

static int out_dim(int in, int k, int pad, int mul, int div) {
  if (div > 1) return in * div;           // dgrad of a stride-`div` conv whose forward output was `in`
  return (in + 2 * pad - k) / mul + 1;
}

static int test_conv(const ConvCase& c, int variant) {
  const int Hout = out_dim(c.Hin, c.KH, c.pad, c.mul, c.div), Wout = out_dim(c.Win, c.KW, c.pad, c.mul, c.div);
  const int M = c.B * Hout * Wout, T = c.KH * c.KW;
  const int out_ld = ((c.N + 31) / 32) * 32;
  std::vector<uint16_t> hin((size_t)c.B * c.Hin * c.Win * c.C), hw((size_t)c.N * T * c.C), hout((size_t)M * out_ld);
  std::vector<float> hb(c.N);
  for (auto& v : hin) v = f2bf(frand());
  for (auto& v : hw) v = f2bf(frand() * 0.25f);
  for (auto& v : hout) v = f2bf(frand());
  for (auto& v : hb) v = frand();
  DBuf<uint16_t> din(hin.size()), dw(hw.size()), dout(hout.size());
  DBuf<float> db(c.N), dst(2 * c.N);
  din.up(hin); dw.up(hw); dout.up(hout); db.up(hb);
  int rc = u2_conv_igemm(din.d, dw.d, dout.d, c.bias ? db.d : nullptr, c.stats ? dst.d : nullptr, c.B, c.Hin, c.Win, c.C, c.C,
                         Hout, Wout, c.N, out_ld, c.KH, c.KW, c.pad, c.pad, c.mul, c.div, c.relu, c.accum, variant, nullptr);
  HIPCHK(hipDeviceSynchronize());
  if (rc) { printf("FAIL %-28s v%d launch rc=%d\n", c.name, variant, rc); return 1; }
  auto got = dout.down();
  auto gst = dst.down();
  double max_err = 0, max_ref = 0, st_err = 0;
  std::vector<double> s0(c.N, 0.0), s1(c.N, 0.0);
  for (int m = 0; m < M; ++m) {
    const int img = m / (Hout * Wout), rem = m % (Hout * Wout), oy = rem / Wout, ox = rem % Wout;
    for (int n = 0; n < c.N; ++n) {
      double acc = 0;
      for (int kh = 0; kh < c.KH; ++kh)
        for (int kw = 0; kw < c.KW; ++kw) {
          int sy = oy * c.mul - c.pad + kh, sx = ox * c.mul - c.pad + kw;
          if (sy < 0 || sx < 0) continue;
          if (c.div > 1) { if (sy % c.div || sx % c.div) continue; sy /= c.div; sx /= c.div; }
          if (sy >= c.Hin || sx >= c.Win) continue;
          const uint16_t* ip = &hin[((size_t)(img * c.Hin + sy) * c.Win + sx) * c.C];
          const uint16_t* wp = &hw[((size_t)n * T + kh * c.KW + kw) * c.C];
          for (int k = 0; k < c.C; ++k) acc += (double)bf2f(ip[k]) * bf2f(wp[k]);
        }
      if (c.bias) acc += hb[n];
      float ref = (float)acc;
      if (c.accum) { ref = bf2f(f2bf(ref)) + bf2f(hout[(size_t)m * out_ld + n]); }
      if (c.relu) ref = fmaxf(ref, 0.f);
      const float g = bf2f(got[(size_t)m * out_ld + n]);
      max_err = fmax(max_err, fabs((double)g - ref));
      max_ref = fmax(max_ref, fabs((double)ref));
      s0[n] += g; s1[n] += (double)g * g;
    }
  }
  if (c.stats)
    for (int n = 0; n < c.N; ++n) {
      st_err = fmax(st_err, fabs(s0[n] - gst[n]) / (1.0 + fabs(s0[n])));
      st_err = fmax(st_err, fabs(s1[n] - gst[c.N + n]) / (1.0 + fabs(s1[n])));
    }
  const bool ok = max_err <= 0.01 * max_ref + 1e-3 && st_err < 1e-3;
  printf("%s %-28s v%d  max_err %.4g (max_ref %.4g) stats_relerr %.3g\n", ok ? "PASS" : "FAIL", c.name, variant, max_err, max_ref, st_err);
  return ok ? 0 : 1;
}

struct WgCase { int B, Hin, Win, C, N, KH, KW, pad, stride; const char* name; };

static int test_wgrad(const WgCase& c, int variant, bool into = false) {
  const int Hout = (c.Hin + 2 * c.pad - c.KH) / c.stride + 1, Wout = (c.Win + 2 * c.pad - c.KW) / c.stride + 1;
  const int M = c.B * Hout * Wout, T = c.KH * c.KW;
  std::vector<uint16_t> hx((size_t)c.B * c.Hin * c.Win * c.C), hdy((size_t)M * c.N);
  for (auto& v : hx) v = f2bf(frand());
  for (auto& v : hdy) v = f2bf(frand());
  DBuf<uint16_t> dx(hx.size()), ddy(hdy.size());
  DBuf<float> ddw((size_t)c.N * T * c.C);
  dx.up(hx); ddy.up(hdy);
  const int nv = into ? c.N - 5 : c.N, cv = into ? c.C - 3 : c.C;  // the reference's [N][Cin][KH][KW] with unpadded N, Cin
  int rc = into ? u2_conv_wgrad_into(dx.d, ddy.d, ddw.d, c.B, c.Hin, c.Win, c.C, c.C, Hout, Wout, c.N, c.N, c.KH, c.KW, c.pad,
                                     c.pad, c.stride, nv, cv, (long long)cv * T, 1, T, variant, nullptr)
                : u2_conv_wgrad(dx.d, ddy.d, ddw.d, c.B, c.Hin, c.Win, c.C, c.C, Hout, Wout, c.N, c.N, c.KH, c.KW, c.pad, c.pad,
                                c.stride, variant, nullptr);
  HIPCHK(hipDeviceSynchronize());
  if (rc) { printf("FAIL %-28s v%d launch rc=%d\n", c.name, variant, rc); return 1; }
  auto got = ddw.down();
  if (into) {  // back to [N][T][C] for the comparison; everything past the valid block must still be zero
    std::vector<float> re((size_t)c.N * T * c.C, 0.f);
    for (int n = 0; n < nv; ++n)
      for (int k = 0; k < cv; ++k)
        for (int t = 0; t < T; ++t) re[((size_t)n * T + t) * c.C + k] = got[((size_t)n * cv + k) * T + t];
    for (size_t i = (size_t)nv * cv * T; i < got.size(); ++i)
      if (got[i] != 0.f) { printf("FAIL %-28s wrote past the valid block\n", c.name); return 1; }
    got = re;
  }
  std::vector<double> ref((size_t)c.N * T * c.C, 0.0);
  for (int m = 0; m < M; ++m) {
    const int img = m / (Hout * Wout), rem = m % (Hout * Wout), oy = rem / Wout, ox = rem % Wout;
    for (int kh = 0; kh < c.KH; ++kh)
      for (int kw = 0; kw < c.KW; ++kw) {
        const int sy = oy * c.stride - c.pad + kh, sx = ox * c.stride - c.pad + kw;
        if (sy < 0 || sx < 0 || sy >= c.Hin || sx >= c.Win) continue;
        const uint16_t* ip = &hx[((size_t)(img * c.Hin + sy) * c.Win + sx) * c.C];
        for (int n = 0; n < c.N; ++n) {
          const double g = bf2f(hdy[(size_t)m * c.N + n]);
          double* r = &ref[((size_t)n * T + kh * c.KW + kw) * c.C];
          for (int k = 0; k < c.C; ++k) r[k] += g * bf2f(ip[k]);
        }
      }
  }
  double max_err = 0, max_ref = 0;
  for (size_t i = 0; i < ref.size(); ++i) {
    if (into && ((int)(i % c.C) >= cv || (int)(i / ((size_t)T * c.C)) >= nv)) continue;
    max_err = fmax(max_err, fabs(ref[i] - got[i])); max_ref = fmax(max_ref, fabs(ref[i]));
  }
  const bool ok = max_err <= 2e-3 * max_ref + 1e-3;
  printf("%s %-28s v%d%s  max_err %.4g (max_ref %.4g)\n", ok ? "PASS" : "FAIL", c.name, variant, into ? " into" : "", max_err, max_ref);
  return ok ? 0 : 1;
}

// u2_conv1x1_bwd_fused: dx = dy . W and dW += dy^T x in one pass over dy (wgrad_stream_kernel<.., DG>)
static int test_wdgrad(int M, int C, int N, int cv, int nv, int variant, const char* name) {
  std::vector<uint16_t> hx((size_t)M * C), hdy((size_t)M * N), hwt((size_t)C * N);
  for (auto& v : hx) v = f2bf(frand());
  for (auto& v : hdy) v = f2bf(frand());
  for (auto& v : hwt) v = f2bf(frand() * 0.25f);      // wt[c][n]
  for (int m = 0; m < M; ++m) for (int n = nv; n < N; ++n) hdy[(size_t)m * N + n] = 0;   // pad channels of dy are zero
  DBuf<uint16_t> dx_in(hx.size()), ddy(hdy.size()), dwt(hwt.size()), ddx((size_t)M * C);
  DBuf<float> ddw((size_t)nv * cv);
  dx_in.up(hx); ddy.up(hdy); dwt.up(hwt);
  std::vector<uint16_t> junk((size_t)M * C, 0x7fc0); ddx.up(junk);
  int rc = u2_conv1x1_bwd_fused(dx_in.d, ddy.d, dwt.d, ddx.d, ddw.d, M, C, C, N, N, N, C, nv, cv, (long long)cv, 1, variant, nullptr);
  HIPCHK(hipDeviceSynchronize());
  if (rc) { printf("FAIL %-28s v%d launch rc=%d\n", name, variant, rc); return 1; }
  auto gdx = ddx.down(); auto gdw = ddw.down();
  double e1 = 0, r1 = 0, e2 = 0, r2 = 0;
  for (int m = 0; m < M; ++m)
    for (int c = 0; c < C; ++c) {
      double acc = 0;
      for (int n = 0; n < N; ++n) acc += (double)bf2f(hdy[(size_t)m * N + n]) * bf2f(hwt[(size_t)c * N + n]);
      e1 = fmax(e1, fabs(acc - bf2f(gdx[(size_t)m * C + c]))); r1 = fmax(r1, fabs(acc));
    }
  std::vector<double> ref((size_t)nv * cv, 0.0);
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < nv; ++n) {
      const double g = bf2f(hdy[(size_t)m * N + n]);
      for (int c = 0; c < cv; ++c) ref[(size_t)n * cv + c] += g * bf2f(hx[(size_t)m * C + c]);
    }
  for (size_t i = 0; i < ref.size(); ++i) { e2 = fmax(e2, fabs(ref[i] - gdw[i])); r2 = fmax(r2, fabs(ref[i])); }
  const bool ok = e1 <= 0.01 * r1 + 1e-3 && e2 <= 2e-3 * r2 + 1e-3 && u2_conv_last_kernel() / 10 == 275;
  printf("%s %-28s v%d  dx err %.4g (max %.4g)  dW err %.4g (max %.4g)  kernel %d\n", ok ? "PASS" : "FAIL", name, variant, e1, r1, e2, r2, u2_conv_last_kernel());
  return ok ? 0 : 1;
}


// u2_conv1x1_bwd_fused_bn against its definition: u2_norm_bwd_apply (relu = 0) into a dy buffer, then u2_conv1x1_bwd_fused on it.
// The data gradient must agree bit for bit (same rounded dy, same MFMA order per block shape is NOT guaranteed between the 4- and
// 8-wave blocks, so a tolerance on dx / dW and bit-equality of the dy the kernel formed, probed through an identity filter).
static int test_wdgrad_bn(int M, int C, int N, int cv, int nv, int variant, const char* name) {
  std::vector<uint16_t> hx((size_t)M * C), hdz((size_t)M * N), hy((size_t)M * N), hwt((size_t)C * N);
  std::vector<float> k1(N), k2(N), k3(N);
  for (auto& v : hx) v = f2bf(frand());
  for (auto& v : hdz) v = f2bf(frand());
  for (auto& v : hy) v = f2bf(frand() * 2.f);
  for (auto& v : hwt) v = f2bf(frand() * 0.25f);
  for (int n = 0; n < N; ++n) { k1[n] = 0.5f + frand(); k2[n] = 0.1f * frand(); k3[n] = 0.05f * frand(); }
  for (int n = nv; n < N; ++n) k1[n] = k2[n] = k3[n] = 0.f;
  for (int m = 0; m < M; ++m) for (int n = nv; n < N; ++n) hdz[(size_t)m * N + n] = hy[(size_t)m * N + n] = 0;
  DBuf<uint16_t> dx_in(hx.size()), ddz(hdz.size()), dy(hy.size()), ddy(hdz.size()), dwt(hwt.size()), ddx((size_t)M * C), ddx2((size_t)M * C);
  DBuf<float> ddw((size_t)nv * cv), ddw2((size_t)nv * cv), dk1(N), dk2(N), dk3(N);
  dx_in.up(hx); ddz.up(hdz); dy.up(hy); dwt.up(hwt); dk1.up(k1); dk2.up(k2); dk3.up(k3);
  std::vector<uint16_t> junk((size_t)M * C, 0x7fc0); ddx.up(junk); ddx2.up(junk);
  int rc = u2_norm_bwd_apply(ddz.d, nullptr, dy.d, dk1.d, dk2.d, dk3.d, ddy.d, nullptr, 1, M, N, N, 0, nullptr, nullptr, nullptr);
  if (!rc) rc = u2_conv1x1_bwd_fused(dx_in.d, ddy.d, dwt.d, ddx2.d, ddw2.d, M, C, C, N, N, N, C, nv, cv, (long long)cv, 1, variant, nullptr);
  HIPCHK(hipDeviceSynchronize());
  if (rc) { printf("FAIL %-28s v%d reference launches rc=%d\n", name, variant, rc); return 1; }
  rc = u2_conv1x1_bwd_fused_bn(dx_in.d, ddz.d, dy.d, dk1.d, dk2.d, dk3.d, dwt.d, ddx.d, ddw.d, M, C, C, N, N, N, C, nv, cv, (long long)cv, 1, variant, nullptr);
  HIPCHK(hipDeviceSynchronize());
  if (rc) { printf("FAIL %-28s v%d launch rc=%d\n", name, variant, rc); return 1; }
  const int code = u2_conv_last_kernel();
  auto gdx = ddx.down(); auto gdx2 = ddx2.down(); auto gdw = ddw.down(); auto gdw2 = ddw2.down();
  double e1 = 0, r1 = 0, e2 = 0, r2 = 0; size_t ndiff = 0;
  for (size_t i = 0; i < gdx.size(); ++i) {
    e1 = fmax(e1, fabs((double)bf2f(gdx[i]) - bf2f(gdx2[i]))); r1 = fmax(r1, fabs((double)bf2f(gdx2[i])));
    ndiff += gdx[i] != gdx2[i];
  }
  for (size_t i = 0; i < gdw.size(); ++i) { e2 = fmax(e2, fabs((double)gdw[i] - gdw2[i])); r2 = fmax(r2, fabs((double)gdw2[i])); }
  // fp32 accumulation order differs between the block shapes: one bf16 ulp on dx, 1e-4 relative on dW
  const bool ok = e1 <= 0.008 * r1 + 1e-6 && e2 <= 2e-4 * r2 + 1e-4 && code == (N > 256 ? 2763 : (variant & 8) ? 2762 : 2761);
  printf("%s %-28s v%d  dx diff %.4g (max %.4g, %zu of %zu differ)  dW diff %.4g (max %.4g)  kernel %d\n", ok ? "PASS" : "FAIL", name, variant, e1,
         r1, ndiff, gdx.size(), e2, r2, code);
  return ok ? 0 : 1;
}

static void bench_conv(const char* name, int B, int H, int W, int C, int N, int K, int pad, int stride, int variant) {
  const int Hout = (H + 2 * pad - K) / stride + 1, Wout = (W + 2 * pad - K) / stride + 1;
  const size_t M = (size_t)B * Hout * Wout;
  DBuf<uint16_t> din((size_t)B * H * W * C), dw((size_t)N * K * K * C), dout(M * N), ddy(M * N);
  DBuf<float> dgw((size_t)N * K * K * C), dst(2 * N);
  std::vector<uint16_t> h(din.n); for (auto& v : h) v = f2bf(frand()); din.up(h);
  std::vector<uint16_t> hw(dw.n); for (auto& v : hw) v = f2bf(frand() * 0.1f); dw.up(hw);
  std::vector<uint16_t> hy(ddy.n); for (auto& v : hy) v = f2bf(frand()); ddy.up(hy);
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  const double flop = 2.0 * M * N * K * K * C;
  auto wgrad_into = [&]() {
    u2_conv_wgrad_into(din.d, ddy.d, dgw.d, B, H, W, C, C, Hout, Wout, N, N, K, K, pad, pad, stride, N, C, (long long)C * K * K, 1,
                       K * K, variant, nullptr);
  };
  for (int pass = 0; pass < 3; ++pass) {
    if (pass == 2 && K == 1) break;
    const int iters = 10;
    for (int i = 0; i < 2; ++i) {
      if (pass == 0) u2_conv_igemm(din.d, dw.d, dout.d, nullptr, dst.d, B, H, W, C, C, Hout, Wout, N, N, K, K, pad, pad, stride, 1, 0, 0, variant, nullptr);
      else if (pass == 1) u2_conv_wgrad(din.d, ddy.d, dgw.d, B, H, W, C, C, Hout, Wout, N, N, K, K, pad, pad, stride, variant, nullptr);
      else wgrad_into();
    }
    HIPCHK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) {
      if (pass == 0) u2_conv_igemm(din.d, dw.d, dout.d, nullptr, dst.d, B, H, W, C, C, Hout, Wout, N, N, K, K, pad, pad, stride, 1, 0, 0, variant, nullptr);
      else if (pass == 1) u2_conv_wgrad(din.d, ddy.d, dgw.d, B, H, W, C, C, Hout, Wout, N, N, K, K, pad, pad, stride, variant, nullptr);
      else wgrad_into();
    }
    HIPCHK(hipEventRecord(e1)); HIPCHK(hipEventSynchronize(e1));
    float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    const double bytes = pass == 0 ? 2.0 * ((double)B * H * W * C + (double)M * N) : 2.0 * ((double)B * H * W * C + (double)M * N);
    printf("BENCH %-26s %-5s v%d  %8.3f ms  %8.1f TFLOP/s  %7.1f GB/s(min traffic)\n", name, pass == 0 ? "fwd" : pass == 1 ? "wgrad" : "wginto", variant, ms,
           flop / ms * 1e-9, bytes / ms * 1e-6);
  }
}


// ---- bench2: every conv shape of the u2seg_R50_800 training step x a list of kernel variants (one allocation, tiled random fill) ----
struct LayerShape { const char* name; int B, H, W, C, N, K, pad, stride, relu, bias, stats; };

static void bench2(int argc, char** argv, bool wgrad) {
  const LayerShape layers[] = {
      {"p2 3x3 256->256 200x336", 16, 200, 336, 256, 256, 3, 1, 1, 0, 0, 1},
      {"p3 3x3 256->256 100x168", 16, 100, 168, 256, 256, 3, 1, 1, 0, 0, 1},
      {"p4 3x3 256->256 50x84", 16, 50, 84, 256, 256, 3, 1, 1, 0, 0, 1},
      {"p5 3x3 256->256 25x42", 16, 25, 42, 256, 256, 3, 1, 1, 0, 0, 1},
      {"rpn 3x3 256->256 200x336 b+r", 16, 200, 336, 256, 256, 3, 1, 1, 1, 1, 0},
      {"sem 3x3 256->128 200x336", 16, 200, 336, 256, 128, 3, 1, 1, 0, 0, 0},
      {"sem 3x3 128->256 200x336", 16, 200, 336, 128, 256, 3, 1, 1, 0, 0, 0},
      {"res3 3x3 128->128 100x168", 16, 100, 168, 128, 128, 3, 1, 1, 0, 0, 1},
      {"res5 3x3 512->512 25x42", 16, 25, 42, 512, 512, 3, 1, 1, 0, 0, 1},
      {"res2 3x3 64->64 200x336", 16, 200, 336, 64, 64, 3, 1, 1, 0, 0, 1},
      {"mask 3x3 256->256 261x14x14", 261, 14, 14, 256, 256, 3, 1, 1, 1, 1, 0},
      {"res4 1x1 256->1024 50x84", 16, 50, 84, 256, 1024, 1, 0, 1, 0, 0, 1},
      {"res4 1x1 1024->256 50x84", 16, 50, 84, 1024, 256, 1, 0, 1, 0, 0, 1},
      {"res3 1x1 128->512 100x168", 16, 100, 168, 128, 512, 1, 0, 1, 0, 0, 1},
      {"res3 1x1 512->128 100x168", 16, 100, 168, 512, 128, 1, 0, 1, 0, 0, 1},
      {"res2 1x1 256->64 200x336", 16, 200, 336, 256, 64, 1, 0, 1, 0, 0, 1},
      {"res2 1x1 64->256 200x336", 16, 200, 336, 64, 256, 1, 0, 1, 0, 0, 1},
      {"lat2 1x1 256->256 200x336", 16, 200, 336, 256, 256, 1, 0, 1, 0, 0, 1},
      {"sem 1x1 256->128 200x336", 16, 200, 336, 256, 128, 1, 0, 1, 0, 0, 1},
      {"res2 1x1 64->64 200x336", 16, 200, 336, 64, 64, 1, 0, 1, 0, 0, 1},
      {"lat3 1x1 512->256 100x168", 16, 100, 168, 512, 256, 1, 0, 1, 0, 0, 1},
      {"res3 1x1 256->512 100x168", 16, 100, 168, 256, 512, 1, 0, 1, 0, 0, 0},
      {"res5 1x1 512->2048 25x42", 16, 25, 42, 512, 2048, 1, 0, 1, 0, 0, 1},
      {"res5 1x1 2048->512 25x42", 16, 25, 42, 2048, 512, 1, 0, 1, 0, 0, 1},
      {"res4 1x1 s2 512->1024 100x168", 16, 100, 168, 512, 1024, 1, 0, 2, 0, 0, 1},
      {"fc1 fwd 7x7 256->1024 M8192", 8192, 7, 7, 256, 1024, 7, 0, 1, 1, 1, 0},
      {"fc1 dgrad gemm 1024->12544", 1, 8192, 1, 1024, 12544, 1, 0, 1, 0, 0, 0},
      {"fc2 gemm 1024->1024 M8192", 1, 8192, 1, 1024, 1024, 1, 0, 1, 1, 1, 0},
      {"gemm 8192^3", 1, 8192, 1, 8192, 8192, 1, 0, 1, 0, 0, 0},
      {"stem gemm 160->64 M4.3M stats", 1, 4300800, 1, 160, 64, 1, 0, 1, 0, 0, 1},
      {"stem gemm 160->64 M4.3M plain", 1, 4300800, 1, 160, 64, 1, 0, 1, 0, 0, 0},
      {"res2 1x1 64->256 plain", 16, 200, 336, 64, 256, 1, 0, 1, 0, 0, 0},
      {"res4 1x1 1024->256 plain", 16, 50, 84, 1024, 256, 1, 0, 1, 0, 0, 0},
      {"res4 1x1 256->1024 plain", 16, 50, 84, 256, 1024, 1, 0, 1, 0, 0, 0},
      {"res4 3x3 256->256 50x84 stats", 16, 50, 84, 256, 256, 3, 1, 1, 0, 0, 1},
      {"res4 3x3 256->256 50x84 plain", 16, 50, 84, 256, 256, 3, 1, 1, 0, 0, 0},
      {"res3 s2 3x3 128->128 stats", 16, 200, 336, 128, 128, 3, 1, 2, 0, 0, 1},
      {"res3 s2 3x3 128->128 plain", 16, 200, 336, 128, 128, 3, 1, 2, 0, 0, 0},
  };
  std::vector<int> variants;
  for (int i = 2; i < argc; ++i) variants.push_back((int)strtol(argv[i], nullptr, 0));
  if (variants.empty()) {
    if (wgrad) variants = {2048, 0, 256, 4096, 4096 | (1 << 14)};
    else variants = {15 << 12, 0, 1 << 12, 2 << 12, 3 << 12, 4 << 12, 5 << 12, 6 << 12};
  }
  size_t in_elems = (size_t)16 * 200 * 336 * 256, out_elems = (size_t)16 * 200 * 336 * 256;
  const size_t w_elems = (size_t)8192 * 8192;
  for (const auto& L : layers) {   // the selected layers may need more (the stem GEMM reads 4.3 M x 160)
    const char* f = getenv("U2_BENCH_LAYERS");
    if (f && !strstr(L.name, f)) continue;
    in_elems = std::max(in_elems, (size_t)L.B * L.H * L.W * L.C);
    out_elems = std::max(out_elems, (size_t)L.B * L.H * L.W * L.N);
  }
  DBuf<uint16_t> din(in_elems), dw(wgrad ? 16 : w_elems), dout(out_elems);
  DBuf<float> db(16384), dst(2 * 16384), dgw(wgrad ? w_elems : 16);
  if (wgrad) {  // the output gradient operand
    std::vector<uint16_t> pat((size_t)1 << 22);
    for (auto& v : pat) v = f2bf(frand());
    for (size_t o = 0; o < out_elems; o += pat.size())
      HIPCHK(hipMemcpy(dout.d + o, pat.data(), std::min(pat.size(), out_elems - o) * 2, hipMemcpyHostToDevice));
  }
  {
    std::vector<uint16_t> pat((size_t)1 << 22);
    for (auto& v : pat) v = f2bf(frand());
    for (size_t o = 0; o < in_elems; o += pat.size())
      HIPCHK(hipMemcpy(din.d + o, pat.data(), std::min(pat.size(), in_elems - o) * 2, hipMemcpyHostToDevice));
    for (auto& v : pat) v = f2bf(frand() * 0.05f);
    for (size_t o = 0; !wgrad && o < w_elems; o += pat.size())
      HIPCHK(hipMemcpy(dw.d + o, pat.data(), std::min(pat.size(), w_elems - o) * 2, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  const char* only = getenv("U2_BENCH_LAYERS");   // substring filter on the layer names
  for (const auto& L : layers) {
    if (only && !strstr(L.name, only)) continue;
    const int Hout = (L.H + 2 * L.pad - L.K) / L.stride + 1, Wout = (L.W + 2 * L.pad - L.K) / L.stride + 1;
    const double M = (double)L.B * Hout * Wout;
    const double flop = 2.0 * M * L.N * L.K * L.K * L.C;
    const double bytes = 2.0 * ((double)L.B * L.H * L.W * L.C + M * L.N + (double)L.N * L.K * L.K * L.C);
    printf("LAYER %-32s GFLOP %8.1f  MB %7.1f |", L.name, flop * 1e-9, bytes * 1e-6);
    // interleaved A/B: ROUNDS passes over the variant list, each pass 1 untimed + 3 timed launches per variant; the MEDIAN
    // pass of every variant is reported (the first thing measured after a pause runs on a cold clock: order effects of
    // 10 % and more otherwise)
    const int ROUNDS = 5;
    std::vector<std::vector<float>> times(variants.size());
    std::vector<int> rcs(variants.size(), 0);
    for (int r = 0; r < ROUNDS; ++r)
      for (size_t vi = 0; vi < variants.size(); ++vi) {
        const int v = variants[vi];
        auto run = [&]() {
          if (wgrad)
            return u2_conv_wgrad(din.d, dout.d, dgw.d, L.B, L.H, L.W, L.C, L.C, Hout, Wout, L.N, L.N, L.K, L.K, L.pad, L.pad, L.stride,
                                 v, nullptr);
          return u2_conv_igemm(din.d, dw.d, dout.d, L.bias ? db.d : nullptr, L.stats ? dst.d : nullptr, L.B, L.H, L.W, L.C, L.C, Hout,
                               Wout, L.N, L.N, L.K, L.K, L.pad, L.pad, L.stride, 1, L.relu, 0, v, nullptr);
        };
        rcs[vi] |= run();
        HIPCHK(hipEventRecord(e0));
        const int iters = 3;
        for (int i = 0; i < iters; ++i) rcs[vi] |= run();
        HIPCHK(hipEventRecord(e1)); HIPCHK(hipEventSynchronize(e1));
        float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        times[vi].push_back(ms / iters);
      }
    for (size_t vi = 0; vi < variants.size(); ++vi) {
      std::sort(times[vi].begin(), times[vi].end());
      const float ms = times[vi][ROUNDS / 2];
      if (rcs[vi]) printf("  v%x rc=%d", variants[vi], rcs[vi]);
      else printf("  v%x %6.3fms %6.0fTF %4.1fTB/s", variants[vi], ms, flop / ms * 1e-9, bytes / ms * 1e-9);
    }
    printf("\n");
    fflush(stdout);
  }
}

int main(int argc, char** argv) {
  int fails = 0;
  if (argc > 1 && !strcmp(argv[1], "bench2")) { bench2(argc, argv, false); return 0; }
  if (argc > 1 && !strcmp(argv[1], "bench2w")) { bench2(argc, argv, true); return 0; }
  if (argc > 1 && !strcmp(argv[1], "one")) {  // a single shape, for PMC profiling
    const int v = argc > 2 ? atoi(argv[2]) : 0;
    bench_conv("fpn_out2 3x3 256->256 B16", 16, 200, 336, 256, 256, 3, 1, 1, v);
    return 0;
  }
  const ConvCase tconvs[] = {
      {2, 36, 32, 64, 256, 3, 3, 1, 1, 1, 0, 0, 0, 1, "tile 3x3 c64 n256 stats"},
      {2, 36, 32, 96, 320, 3, 3, 1, 1, 1, 1, 0, 1, 0, "tile 3x3 c96 n320 bias relu"},
      {1, 700, 3, 256, 512, 1, 1, 0, 1, 1, 0, 0, 0, 1, "tile 1x1 c256 n512 stats"},
      {1, 700, 3, 256, 512, 1, 1, 0, 1, 1, 1, 1, 1, 0, "tile 1x1 c256 n512 accumulate bias relu"},
      {2, 33, 21, 64, 264, 1, 1, 0, 1, 1, 1, 1, 0, 0, "tile 1x1 c64 n264 accumulate relu"},
      {2, 37, 41, 160, 128, 1, 1, 0, 2, 1, 0, 0, 0, 1, "tile 1x1 s2 c160 n128 stats"},
      {2, 18, 20, 128, 128, 3, 3, 1, 1, 2, 0, 0, 0, 0, "tile dgrad(3x3 s2) c128 n128"},
      {1, 45, 23, 64, 72, 3, 3, 1, 1, 1, 0, 0, 1, 1, "tile 3x3 c64 n72 bias stats"},
      {3, 7, 7, 32, 136, 7, 7, 0, 1, 1, 1, 0, 1, 0, "tile fc 7x7 c32 n136 bias relu"},
  };
  // conv_halo.hip (variant bit 24): 3x3 / stride 1 / pad 1; partial patches in both directions, one and many slabs, channel
  // tails, several tiles per work-group (bit 16), bias / ReLU / statistics
  const ConvCase hconvs[] = {
      {2, 16, 32, 32, 128, 3, 3, 1, 1, 1, 0, 0, 0, 1, "halo 16x32 c32 n128 stats"},
      {1, 19, 45, 64, 128, 3, 3, 1, 1, 1, 0, 0, 0, 1, "halo 19x45 c64 n128 stats"},
      {2, 33, 70, 96, 256, 3, 3, 1, 1, 1, 1, 0, 1, 0, "halo 33x70 c96 n256 bias relu"},
      {1, 40, 31, 128, 136, 3, 3, 1, 1, 1, 0, 0, 0, 1, "halo 40x31 c128 n136 stats"},
      {3, 7, 9, 32, 72, 3, 3, 1, 1, 1, 0, 0, 1, 1, "halo 7x9 c32 n72 bias stats"},
      {1, 50, 84, 256, 256, 3, 3, 1, 1, 1, 0, 0, 0, 1, "halo 50x84 c256 n256 stats"},
      // <= 64 output channels: the 64-channel tiles (code 301)
      {2, 33, 70, 64, 64, 3, 3, 1, 1, 1, 0, 0, 0, 1, "halo 33x70 c64 n64 stats"},
      {1, 19, 45, 96, 40, 3, 3, 1, 1, 1, 1, 0, 1, 0, "halo 19x45 c96 n40 bias relu"},
      {3, 7, 9, 32, 8, 3, 3, 1, 1, 1, 0, 0, 1, 1, "halo 7x9 c32 n8 bias stats"},
      {1, 50, 84, 256, 64, 3, 3, 1, 1, 1, 0, 0, 0, 1, "halo 50x84 c256 n64 stats"},
  };
  if (argc > 1 && !strcmp(argv[1], "chalo")) {
    const int extra = argc > 2 ? (int)strtol(argv[2], nullptr, 0) : 0;
    for (int tiny = 0; tiny < 2; ++tiny)
      for (const auto& c : hconvs) {
        fails += test_conv(c, (1 << 24) | (tiny << 16) | extra);
        const int want = (c.N <= 64 && !((extra >> 17) & 1)) ? 301 : 300;
        if (u2_conv_last_kernel() != want) { printf("FAIL %-28s did not take the halo conv kernel %d (%d)\n", c.name, want, u2_conv_last_kernel()); ++fails; }
      }
    printf("SELFTEST chalo %s (%d failures)\n", fails ? "FAILED" : "OK", fails);
    return fails ? 1 : 0;
  }
  if (argc > 2 && !strcmp(argv[1], "tile")) {  // one persistent-tile configuration only
    const int cfg = atoi(argv[2]);
    const int extra = argc > 3 ? (int)strtol(argv[3], nullptr, 0) : 0;  // further variant bits (bit 17: slab-major order)
    for (int tiny = 0; tiny < 2; ++tiny)
      for (const auto& c : tconvs) fails += test_conv(c, (cfg << 12) | (tiny << 16) | extra);
    printf("SELFTEST tile %d %s (%d failures)\n", cfg, fails ? "FAILED" : "OK", fails);
    return fails ? 1 : 0;
  }
  const WgCase halo_cases[] = {  // 3x3 / stride 1 / pad 1 only: wgrad_halo.hip (variant bit 12), bits 14-15 = rounds - 1
      {2, 14, 14, 64, 64, 3, 3, 1, 1, "halo 14x14 c64 n64"},
      {1, 13, 17, 128, 136, 3, 3, 1, 1, "halo 13x17 c128 n136"},
      {3, 25, 42, 96, 200, 3, 3, 1, 1, "halo 25x42 c96 n200"},
      {2, 40, 70, 64, 256, 3, 3, 1, 1, "halo 40x70 c64 n256"},
      {2, 3, 11, 32, 8, 3, 3, 1, 1, "halo 3x11 c32 n8 (1 step)"},
      {1, 5, 101, 32, 72, 3, 3, 1, 1, "halo 5x101 c32 n72 (2 steps)"},
      {1, 8, 95, 64, 40, 3, 3, 1, 1, "halo 8x95 c64 n40 (3 steps)"},
      {1, 12, 84, 32, 16, 3, 3, 1, 1, "halo 12x84 c32 n16 (4 steps)"},
      {1, 5, 300, 32, 72, 3, 3, 1, 1, "halo 5x300 c32 n72"},
  };
  auto run_halo = [&]() {
    int f = 0;
    for (const auto& c : halo_cases) {
      f += test_wgrad(c, 4096);
      if (u2_conv_last_kernel() != 2900) { printf("FAIL %-28s did not take the halo kernel (%d)\n", c.name, u2_conv_last_kernel()); ++f; }
      f += test_wgrad(c, 4096 | (1 << 14));
      f += test_wgrad(c, 4096, true);
      // partial blocks + reduction pass instead of the atomic epilogue (bit 17), plain and into a strided gradient
      f += test_wgrad(c, 4096 | (1 << 17));
      f += test_wgrad(c, 4096 | (1 << 17), true);
      f += test_wgrad(c, 4096 | (1 << 17) | (1 << 14), true);
    }
    return f;
  };
  // wgrad_stream.hip (variant bit 18 forces it on small maps; bits 21-23 block configuration, bit 20: 8 pixel ranges, bits 24-25
  // work-groups per CU): M not a multiple of 32, channel tails on both sides, blocks tiled over N and over C, 1 .. many steps per range
  const WgCase ws_cases[] = {
      {1, 37, 29, 64, 256, 1, 1, 0, 1, "wstream 37x29 c64 n256"},
      {2, 33, 21, 256, 64, 1, 1, 0, 1, "wstream 33x21 c256 n64"},
      {1, 45, 23, 128, 136, 1, 1, 0, 1, "wstream 45x23 c128 n136"},
      {1, 19, 45, 264, 520, 1, 1, 0, 1, "wstream 19x45 c264 n520"},
      {3, 7, 9, 32, 32, 1, 1, 0, 1, "wstream 7x9 c32 n32 (few steps)"},
      {1, 61, 67, 160, 64, 1, 1, 0, 1, "wstream 61x67 c160 n64 (stem)"},
      {2, 40, 70, 512, 128, 1, 1, 0, 1, "wstream 40x70 c512 n128"},
      {1, 1, 5, 64, 64, 1, 1, 0, 1, "wstream 5 px"},
  };
  if (argc > 1 && !strcmp(argv[1], "bench_wd")) {
    // both gradients of a 1x1 layer: two launches (u2_conv_igemm as the data gradient + u2_conv_wgrad) vs u2_conv1x1_bwd_fused
    const int shapes[][3] = {{16 * 200 * 336, 64, 256}, {16 * 100 * 168, 128, 512}, {16 * 200 * 336, 32, 256}, {16 * 100 * 168, 64, 512}};
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (const auto& sh : shapes) {
      const int M = sh[0], C = sh[1], N = sh[2];
      DBuf<uint16_t> dx_in((size_t)M * C), ddy((size_t)M * N), dwt((size_t)C * N), ddx((size_t)M * C);
      DBuf<float> ddw((size_t)N * C);
      std::vector<uint16_t> pat((size_t)1 << 22);
      for (auto& v : pat) v = f2bf(frand());
      for (size_t o = 0; o < ddy.n; o += pat.size()) HIPCHK(hipMemcpy(ddy.d + o, pat.data(), std::min(pat.size(), ddy.n - o) * 2, hipMemcpyHostToDevice));
      for (size_t o = 0; o < dx_in.n; o += pat.size()) HIPCHK(hipMemcpy(dx_in.d + o, pat.data(), std::min(pat.size(), dx_in.n - o) * 2, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(dwt.d, pat.data(), dwt.n * 2, hipMemcpyHostToDevice));
      std::vector<float> t[2];
      for (int r = 0; r < 5; ++r)
        for (int mode = 0; mode < 2; ++mode) {
          auto run = [&]() {
            if (mode == 0) {
              u2_conv_igemm(ddy.d, dwt.d, ddx.d, nullptr, nullptr, 1, M, 1, N, N, M, 1, C, C, 1, 1, 0, 0, 1, 1, 0, 0, 0, nullptr);
              u2_conv_wgrad(dx_in.d, ddy.d, ddw.d, 1, M, 1, C, C, M, 1, N, N, 1, 1, 0, 0, 1, 0, nullptr);
            } else {
              u2_conv1x1_bwd_fused(dx_in.d, ddy.d, dwt.d, ddx.d, ddw.d, M, C, C, N, N, N, C, N, C, (long long)C, 1, 1, nullptr);
            }
          };
          run();
          HIPCHK(hipEventRecord(e0));
          for (int i = 0; i < 3; ++i) run();
          HIPCHK(hipEventRecord(e1)); HIPCHK(hipEventSynchronize(e1));
          float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
          t[mode].push_back(ms / 3);
        }
      std::sort(t[0].begin(), t[0].end()); std::sort(t[1].begin(), t[1].end());
      const double by2 = 2.0 * ((double)M * N * 2 + (double)M * C * 2), by1 = 2.0 * ((double)M * N + (double)M * C * 2);
      printf("BENCH_WD M=%d %d->%d  two launches %.3f ms (%.2f TB/s of their %.0f MB)   fused %.3f ms (%.2f TB/s of its %.0f MB)\n", M, C, N,
             t[0][2], by2 / t[0][2] * 1e-9, by2 * 1e-6, t[1][2], by1 / t[1][2] * 1e-9, by1 * 1e-6);
    }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "tile7")) {
    // conv_tile configuration 7 (whole K tiles of 64 channels per ring stage): whole tiles / stream-K shares, full grid / 8 groups
    const ConvCase k64[] = {
        {1, 700, 3, 256, 512, 1, 1, 0, 1, 1, 0, 0, 0, 1, "tile7 1x1 c256 n512 stats"},
        {1, 700, 3, 256, 512, 1, 1, 0, 1, 1, 1, 0, 1, 0, "tile7 1x1 c256 n512 bias relu"},
        {1, 300, 3, 1024, 256, 1, 1, 0, 1, 1, 0, 0, 0, 1, "tile7 1x1 c1024 n256 stats"},
        {2, 36, 32, 64, 256, 3, 3, 1, 1, 1, 0, 0, 0, 1, "tile7 3x3 c64 n256 stats"},
        {2, 18, 20, 128, 128, 3, 3, 1, 1, 2, 0, 0, 0, 0, "tile7 dgrad(3x3 s2) c128 n128"},
        {2, 37, 41, 192, 136, 1, 1, 0, 2, 1, 0, 0, 0, 1, "tile7 1x1 s2 c192 n136 stats"},
        {1, 1100, 1, 512, 264, 1, 1, 0, 1, 1, 0, 0, 0, 0, "tile7 gemm k512 n264"},
    };
    for (int v : {7 << 12, (7 << 12) | (1 << 16), (7 << 12) | (1 << 27), (7 << 12) | (1 << 27) | (1 << 16), (7 << 12) | (1 << 28)})
      for (const auto& c : k64) {
        fails += test_conv(c, v);
        if (u2_conv_last_kernel() % 100 != 7) { printf("FAIL %-28s v%x did not take configuration 7 (%d)\n", c.name, v, u2_conv_last_kernel()); ++fails; }
      }
    printf("SELFTEST tile7 %s (%d failures)\n", fails ? "FAILED" : "OK", fails);
    return fails ? 1 : 0;
  }
  if (argc > 1 && !strcmp(argv[1], "wdgrad_bn")) {
    for (int v : {1, 3, 1 | 8, 3 | 8}) {   // forced; | 2: 8 pixel ranges; | 8: the 4-wave block, two work-groups per CU
      fails += test_wdgrad_bn(37 * 29, 64, 256, 64, 256, v, "wdgrad_bn 64->256");
      fails += test_wdgrad_bn(33 * 21 * 2, 40, 200, 37, 196, v, "wdgrad_bn 40->200 (tails)");
      fails += test_wdgrad_bn(5, 32, 136, 32, 136, v, "wdgrad_bn 5 px");
      fails += test_wdgrad_bn(40 * 70, 8, 16, 8, 16, v, "wdgrad_bn 8->16");
      fails += test_wdgrad_bn(64 * 131, 64, 256, 64, 256, v, "wdgrad_bn 64->256 8384 px");
      if (!(v & 8)) {
        fails += test_wdgrad_bn(19 * 45, 128, 512, 128, 512, v | 16, "wdgrad_bn 128->512");
        fails += test_wdgrad_bn(45 * 23, 72, 264, 70, 260, v | 16, "wdgrad_bn 72->264 (wide, tails)");
      }
    }
    printf("SELFTEST wdgrad_bn %s (%d failures)\n", fails ? "FAILED" : "OK", fails);
    return fails ? 1 : 0;
  }
  if (argc > 1 && !strcmp(argv[1], "bench_wd_bn")) {
    // the tail of a res2 block: apply + fused (two launches) vs the one launch (argv[2] == "res3": the tail of a res3 block)
    const bool res3 = argc > 2 && !strcmp(argv[2], "res3");
    const int M = res3 ? 16 * 100 * 168 : 16 * 200 * 336, C = res3 ? 128 : 64, N = res3 ? 512 : 256;
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    DBuf<uint16_t> dx_in((size_t)M * C), ddz((size_t)M * N), dy((size_t)M * N), ddy((size_t)M * N), dwt((size_t)C * N), ddx((size_t)M * C);
    DBuf<float> ddw((size_t)N * C), dk((size_t)3 * N);
    std::vector<uint16_t> pat((size_t)1 << 22);
    for (auto& v : pat) v = f2bf(frand());
    for (size_t o = 0; o < ddz.n; o += pat.size()) {
      HIPCHK(hipMemcpy(ddz.d + o, pat.data(), std::min(pat.size(), ddz.n - o) * 2, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(dy.d + o, pat.data() + 17, std::min(pat.size() - 32, dy.n - o) * 2, hipMemcpyHostToDevice));
    }
    for (size_t o = 0; o < dx_in.n; o += pat.size()) HIPCHK(hipMemcpy(dx_in.d + o, pat.data(), std::min(pat.size(), dx_in.n - o) * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dwt.d, pat.data(), dwt.n * 2, hipMemcpyHostToDevice));
    std::vector<float> hk((size_t)3 * N, 0.25f); dk.up(hk);
    std::vector<float> t[3];
    for (int r = 0; r < 5; ++r)
      for (int mode = 0; mode < 3; ++mode) {
        auto run = [&]() {
          if (mode == 0) {
            u2_norm_bwd_apply(ddz.d, nullptr, dy.d, dk.d, dk.d + N, dk.d + 2 * N, ddy.d, nullptr, 1, M, N, N, 0, nullptr, nullptr, nullptr);
            u2_conv1x1_bwd_fused(dx_in.d, ddy.d, dwt.d, ddx.d, ddw.d, M, C, C, N, N, N, C, N, C, (long long)C, 1, 1, nullptr);
          } else {
            u2_conv1x1_bwd_fused_bn(dx_in.d, ddz.d, dy.d, dk.d, dk.d + N, dk.d + 2 * N, dwt.d, ddx.d, ddw.d, M, C, C, N, N, N, C, N, C, (long long)C, 1,
                                    res3 ? 1 | 16 : mode == 1 ? 1 : 1 | 8, nullptr);
          }
        };
        run();
        HIPCHK(hipEventRecord(e0));
        for (int i = 0; i < 3; ++i) run();
        HIPCHK(hipEventRecord(e1)); HIPCHK(hipEventSynchronize(e1));
        float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        t[mode].push_back(ms / 3);
      }
    for (auto& v : t) std::sort(v.begin(), v.end());
    const double by2 = 2.0 * ((double)M * N * 4 + (double)M * C * 2), by1 = 2.0 * ((double)M * N * 2 + (double)M * C * 2);
    printf("BENCH_WD_BN M=%d %d->%d  apply + fused %.3f ms (%.2f TB/s of their %.0f MB)   one launch %.3f ms (%.2f TB/s of its %.0f MB)   "
           "4-wave form %.3f ms\n", M, C, N, t[0][2], by2 / t[0][2] * 1e-9, by2 * 1e-6, t[1][2], by1 / t[1][2] * 1e-9, by1 * 1e-6, t[2][2]);
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "wdgrad")) {
    for (int v = 1; v <= 3; v += 2) {   // forced, full grid / forced, 8 pixel ranges
      fails += test_wdgrad(37 * 29, 64, 256, 64, 256, v, "wdgrad 64->256");
      fails += test_wdgrad(33 * 21 * 2, 40, 200, 37, 196, v, "wdgrad 40->200 (tails)");
      fails += test_wdgrad(19 * 45, 128, 512, 128, 512, v, "wdgrad 128->512");
      fails += test_wdgrad(45 * 23, 72, 264, 70, 260, v, "wdgrad 72->264 (wide, tails)");
      fails += test_wdgrad(5, 32, 136, 32, 136, v, "wdgrad 5 px");
      fails += test_wdgrad(40 * 70, 8, 16, 8, 16, v, "wdgrad 8->16");
    }
    printf("SELFTEST wdgrad %s (%d failures)\n", fails ? "FAILED" : "OK", fails);
    return fails ? 1 : 0;
  }
  if (argc > 1 && !strcmp(argv[1], "wstream")) {
    for (int cfg = 0; cfg <= 4; ++cfg)
      for (int tiny = 0; tiny < 2; ++tiny)
        for (const auto& c : ws_cases) {
          const int v = (1 << 18) | (tiny << 20) | (cfg << 21);
          fails += test_wgrad(c, v);
          if (u2_conv_last_kernel() / 100 != 27) { printf("FAIL %-28s did not take the streaming kernel (%d)\n", c.name, u2_conv_last_kernel()); ++fails; }
          if (cfg == 0 || tiny) fails += test_wgrad(c, v, true);
          if (cfg >= 1 && cfg <= 3 && !tiny) fails += test_wgrad(c, v | (1 << 24));   // one work-group per CU
        }
    printf("SELFTEST wstream %s (%d failures)\n", fails ? "FAILED" : "OK", fails);
    return fails ? 1 : 0;
  }
  if (argc > 1 && !strcmp(argv[1], "halo")) {
    fails = run_halo();
    printf("SELFTEST halo %s (%d failures)\n", fails ? "FAILED" : "OK", fails);
    return fails ? 1 : 0;
  }
  hipDeviceProp_t prop; HIPCHK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs %d  abi %d\n", prop.gcnArchName, prop.multiProcessorCount, u2_abi_version());
  const ConvCase convs[] = {
      {2, 9, 11, 64, 64, 1, 1, 0, 1, 1, 0, 0, 0, 1, "1x1 c64 n64 stats"},
      {1, 13, 17, 128, 200, 3, 3, 1, 1, 1, 1, 0, 1, 0, "3x3 c128 n200 bias relu"},
      {2, 14, 18, 64, 96, 3, 3, 1, 2, 1, 0, 0, 0, 1, "3x3 s2 c64 n96 stats"},
      {2, 7, 9, 96, 64, 3, 3, 1, 1, 2, 0, 0, 0, 0, "dgrad(3x3 s2) c96(bk32) n64"},
      {1, 8, 10, 256, 40, 1, 1, 0, 2, 1, 0, 0, 0, 0, "1x1 s2 c256 n40"},
      {1, 300, 1, 160, 64, 1, 1, 0, 1, 1, 0, 0, 0, 1, "gemm k160(bk32) n64 stats"},
      {1, 200, 1, 128, 801, 1, 1, 0, 1, 1, 0, 0, 1, 0, "linear k128 n801 bias"},
      {1, 9, 9, 64, 64, 3, 3, 1, 1, 1, 1, 1, 0, 0, "3x3 accumulate relu"},
  };
  const int cvs[] = {0, 1, 32, 48, 4, 4 | 32, 8, 8 | 4 | 32, 256, 256 | 1024};
  for (int v : cvs)
    for (const auto& c : convs) fails += test_conv(c, v);
  // persistent tile kernels (conv_tile.hip): variant bits 12-15 pick the configuration, bit 16 caps the grid at 8
  // work-groups so that every work-group walks several tiles (cross-tile prefetch, accumulator reset, tail waits)
  for (int cfg = 1; cfg <= 6; ++cfg)
    for (int tiny = 0; tiny < 2; ++tiny)
      for (const auto& c : tconvs) fails += test_conv(c, (cfg << 12) | (tiny << 16));
  const WgCase wgs[] = {
      {2, 9, 11, 64, 64, 1, 1, 0, 1, "wgrad 1x1 c64 n64"},
      {1, 13, 17, 128, 136, 3, 3, 1, 1, "wgrad 3x3 c128 n136"},
      {2, 14, 18, 72, 96, 3, 3, 1, 2, "wgrad 3x3 s2 c72 n96"},
      {1, 700, 1, 160, 64, 1, 1, 0, 1, "wgrad gemm k160 n64"},
  };
  for (int v = 0; v < 4; ++v)
    for (const auto& c : wgs) fails += test_wgrad(c, v);
  for (const auto& c : wgs) fails += test_wgrad(c, 64);
  for (const auto& c : wgs) fails += test_wgrad(c, 1 << 9);
  for (const auto& c : wgs) fails += test_wgrad(c, 2 << 9);
  for (const auto& c : wgs) fails += test_wgrad(c, 256);
  for (const auto& c : wgs) fails += test_wgrad(c, 256, true);
  for (const auto& c : wgs) fails += test_wgrad(c, 0, true);
  fails += run_halo();
  printf("SELFTEST %s (%d failures)\n", fails ? "FAILED" : "OK", fails);
  if (argc > 1 && !strcmp(argv[1], "bench")) {
    for (int v = 0; v < 1; ++v) {
      bench_conv("res2 3x3 64->64 B16", 16, 200, 336, 64, 64, 3, 1, 1, v);
      bench_conv("res2 1x1 64->256 B16", 16, 200, 336, 64, 256, 1, 0, 1, v);
      bench_conv("res2 1x1 256->64 B16", 16, 200, 336, 256, 64, 1, 0, 1, v);
      bench_conv("res3 3x3 128->128 B16", 16, 100, 168, 128, 128, 3, 1, 1, v);
      bench_conv("res4 3x3 256->256 B16", 16, 50, 84, 256, 256, 3, 1, 1, v);
      bench_conv("res4 1x1 1024->256 B16", 16, 50, 84, 1024, 256, 1, 0, 1, v);
      bench_conv("res5 3x3 512->512 B16", 16, 25, 42, 512, 512, 3, 1, 1, v);
      bench_conv("fpn_out2 3x3 256->256 B16", 16, 200, 336, 256, 256, 3, 1, 1, v | 512);
      bench_conv("fpn_out3 3x3 256->256 B16", 16, 100, 168, 256, 256, 3, 1, 1, v | 512);
      bench_conv("lat2 1x1 256->256 B16", 16, 200, 336, 256, 256, 1, 0, 1, v | 512);
      bench_conv("fc1 12544->1024 M8192", 1, 8192, 1, 12544, 1024, 1, 0, 1, v | 512);
      bench_conv("gemm 8192x8192x8192", 1, 8192, 1, 8192, 8192, 1, 0, 1, v | 512);
    }
    for (int v : {512 | 4 | 16}) {  // 128 x 128 tile, BK 32, two-stage ring: 32 KB LDS, three work-groups per CU
      bench_conv("fpn_out2 3x3 256->256 B16", 16, 200, 336, 256, 256, 3, 1, 1, v);
      bench_conv("res3 3x3 128->128 B16", 16, 100, 168, 128, 128, 3, 1, 1, v);
      bench_conv("res4 3x3 256->256 B16", 16, 50, 84, 256, 256, 3, 1, 1, v);
      bench_conv("res4 1x1 1024->256 B16", 16, 50, 84, 1024, 256, 1, 0, 1, v);
      bench_conv("res4 1x1 256->1024 B16", 16, 50, 84, 256, 1024, 1, 0, 1, v);
      bench_conv("res3 1x1 512->128 B16", 16, 100, 168, 512, 128, 1, 0, 1, v);
    }
    for (int v : {512}) {
      bench_conv("res4 1x1 256->1024 B16", 16, 50, 84, 256, 1024, 1, 0, 1, v);
      bench_conv("res3 1x1 512->128 B16", 16, 100, 168, 512, 128, 1, 0, 1, v);
    }
    for (int v : {256, 256 | 1024, 256 | 1024 | 2048}) {  // 256 x 256 tile, 8 waves; | 1024: staggered wave groups
      bench_conv("fpn_out2 3x3 256->256 B16", 16, 200, 336, 256, 256, 3, 1, 1, v);
      bench_conv("fpn_out3 3x3 256->256 B16", 16, 100, 168, 256, 256, 3, 1, 1, v);
      bench_conv("res4 3x3 256->256 B16", 16, 50, 84, 256, 256, 3, 1, 1, v);
      bench_conv("lat2 1x1 256->256 B16", 16, 200, 336, 256, 256, 1, 0, 1, v);
      bench_conv("res4 1x1 1024->256 B16", 16, 50, 84, 1024, 256, 1, 0, 1, v);
      bench_conv("res5 3x3 512->512 B16", 16, 25, 42, 512, 512, 3, 1, 1, v);
      bench_conv("fc1 12544->1024 M8192", 1, 8192, 1, 12544, 1024, 1, 0, 1, v);
      bench_conv("gemm 8192x8192x8192", 1, 8192, 1, 8192, 8192, 1, 0, 1, v);
    }
    const int vs[] = {512};  // wgrad: 128 x 128 tiles only (the default picks 256 x 256 when N and C are multiples of 256)
    for (int v : vs) {
      bench_conv("res2 3x3 64->64 B16", 16, 200, 336, 64, 64, 3, 1, 1, v);
      bench_conv("res3 3x3 128->128 B16", 16, 100, 168, 128, 128, 3, 1, 1, v);
      bench_conv("fpn_out2 3x3 256->256 B16", 16, 200, 336, 256, 256, 3, 1, 1, v);
      bench_conv("res4 3x3 256->256 B16", 16, 50, 84, 256, 256, 3, 1, 1, v);
      bench_conv("res4 1x1 1024->256 B16", 16, 50, 84, 1024, 256, 1, 0, 1, v);
      bench_conv("res5 3x3 512->512 B16", 16, 25, 42, 512, 512, 3, 1, 1, v);
      bench_conv("res2 1x1 64->256 B16", 16, 200, 336, 64, 256, 1, 0, 1, v);
      bench_conv("fc1 12544->1024 M8192", 1, 8192, 1, 12544, 1024, 1, 0, 1, v);
      bench_conv("gemm 8192x8192x8192", 1, 8192, 1, 8192, 8192, 1, 0, 1, v);
    }
  }
  return fails ? 1 : 0;
}
