// Implicit-GEMM convolution family on CDNA4 MFMA (gfx950), bf16 in / fp32 accumulate.
//
// Replaces the F.conv2d / F.linear / ConvTranspose2d calls behind
//   detectron2/layers/wrappers.py:127-134 (Conv2d.forward), roi_heads/box_head.py:94-97,
//   roi_heads/fast_rcnn.py:288-305, roi_heads/mask_head.py:287-290   (reference file:line)
// and their autograd backward (dgrad = the same gather with mul/div swapped and a flipped,
// transposed filter; wgrad = pixel-reduction GEMM with split-K + fp32 atomics).
//
// Data layout (HBM): activations NHWC bf16 ("pixel rows" of C channels, pitch in_ld), weights
// [N][KH*KW][C] bf16 (K contiguous per output channel), outputs [M][out_ld] bf16, M = B*Hout*Wout.
//
// Tile: 128 pixels x 128 output channels x BK reduction per workgroup of 4 waves (2x2, 64x64 per
// wave, 4x4 v_mfma_f32_16x16x32_bf16).  Operands reach LDS by global_load_lds (16 B per lane, the
// LDS image is lane-linear, the XOR bank swizzle is applied on the *source* chunk index and again
// on the fragment read).  Zero padding / row tails read a 16-byte zero page instead of branching.
// The MFMA is issued with the weight tile as the "A" operand and the pixel tile as "B", so every
// lane ends up with 4 consecutive channels of one pixel; the tile is then transposed through LDS
// and written with 16-byte coalesced stores (bias / accumulate / ReLU / BN column statistics fused).
#include <stdlib.h>
#include <mutex>
#include <vector>

#include "common.h"
#include "conv_args.h"
#include "u2seg_hip.h"

using namespace u2conv;
namespace {

template <int BK, bool GLDS, int TM, int TN, int NST>
__global__ __launch_bounds__((TM / 64) * (TN / 64) * 64, 2) void conv_igemm_kernel(const ConvArgs a) {
  constexpr int NW = (TM / 64) * (TN / 64);   // waves per work-group, one 64x64 sub-tile each
  constexpr int NT = NW * 64;
  constexpr int CPR = BK / 8;                 // 16-byte chunks per tile row
  constexpr int NCHP = (TM * CPR) / NT;       // chunks per thread, pixel operand
  constexpr int NCHW = (TN * CPR) / NT;       // chunks per thread, weight operand
  static_assert(NCHP >= 1 && NCHW >= 1, "tile too small for the thread count");
  constexpr int ROWS_PER_WAVE_INSTR = 64 / CPR;
  constexpr int PTILE_BYTES = TM * BK * 2;
  constexpr int WTILE_BYTES = TN * BK * 2;
  constexpr int STAGE_BYTES = PTILE_BYTES + WTILE_BYTES;
  constexpr int OPITCH = TN + 8;              // bf16 elements per row of the staged output tile
  constexpr int WM = TM / 64;                 // waves along the pixel dimension
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = a.tiles_m * a.tiles_n;
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int tile_m = tile / a.tiles_n;
  const int tile_n = tile - tile_m * a.tiles_n;
  const int m0 = tile_m * TM;
  const int n0 = tile_n * TN;

  // ---- per-thread chunk bookkeeping (rows are fixed across the K loop) ----
  int p_img[NCHP], p_by[NCHP], p_bx[NCHP];
  bool p_ok[NCHP];
  const bf16_t* w_row[NCHW];
  int cc;  // data chunk (8 channels) this lane fetches within a row; identical for all NCH rows
  {
    const int row_in_instr = lane / CPR;
    const int slot = lane % CPR;
    cc = slot ^ swz<BK>(row_in_instr);  // (instr,wave) offsets are multiples of the swizzle period
    const int hw = a.Hout * a.Wout;
#pragma unroll
    for (int i = 0; i < NCHP; ++i) {
      const int row = (i * NW + w) * ROWS_PER_WAVE_INSTR + row_in_instr;
      const int m = m0 + row;
      p_ok[i] = m < a.M;
      const int mm = p_ok[i] ? m : 0;
      const int img = mm / hw;
      const int rem = mm - img * hw;
      const int oy = rem / a.Wout;
      const int ox = rem - oy * a.Wout;
      p_img[i] = img;
      p_by[i] = oy * a.mul;
      p_bx[i] = ox * a.mul;
    }
#pragma unroll
    for (int i = 0; i < NCHW; ++i) {
      const int n = n0 + (i * NW + w) * ROWS_PER_WAVE_INSTR + row_in_instr;
      w_row[i] = (n < a.N) ? a.wt + (size_t)n * ((size_t)a.wt_taps * a.C) + cc * 8 : nullptr;
    }
  }

  const int kc_per_tap = a.C / BK;
  const int nk = a.ntaps * kc_per_tap;

  uint4 stage_regs[GLDS ? 1 : NCHP + NCHW];

  // Source pointers are (re)derived once per filter tap; inside a tap consecutive K tiles only advance by BK channels
  // (rows that read the zero page do not advance).  issue() is always called with kt = 0, 1, 2, ... in order.
  const bf16_t* p_src[NCHP];
  int p_inc[NCHP];
  const bf16_t* w_src[NCHW];
#pragma unroll
  for (int i = 0; i < NCHW; ++i) w_src[i] = w_row[i] ? w_row[i] : a.zero;
  int kc_in_tap = 0, tap_cur = 0, reg_kh = 0, reg_kw = 0;

  auto issue = [&](int kt, int buf) {
    (void)kt;
    if (kc_in_tap == 0) {
      int dy, dx, wtap;
      if (a.remap_out) {
        dy = a.tap_dy[tap_cur]; dx = a.tap_dx[tap_cur]; wtap = a.tap_w[tap_cur];
      } else {
        dy = reg_kh - a.pad_h; dx = reg_kw - a.pad_w; wtap = tap_cur;
        if (++reg_kw == a.KW) { reg_kw = 0; ++reg_kh; }
      }
#pragma unroll
      for (int i = 0; i < NCHP; ++i) {
        const int sy = p_by[i] + dy, sx = p_bx[i] + dx;
        const bool ok = p_ok[i] && sy >= 0 && sx >= 0 && sy < a.Hin && sx < a.Win;
        p_src[i] = ok ? a.in + ((size_t)(p_img[i] * a.Hin + sy) * a.Win + sx) * a.in_ld + cc * 8 : a.zero;
        p_inc[i] = ok ? BK : 0;
      }
#pragma unroll
      for (int i = 0; i < NCHW; ++i) w_src[i] = w_row[i] ? w_row[i] + (size_t)wtap * a.C : a.zero;
    }
    unsigned char* pbase = smem + buf * STAGE_BYTES;
    unsigned char* wbase = pbase + PTILE_BYTES;
#pragma unroll
    for (int i = 0; i < NCHP; ++i) {
      if constexpr (GLDS) glds16(p_src[i], pbase + (i * NW + w) * 1024);
      else stage_regs[i] = *reinterpret_cast<const uint4*>(p_src[i]);
      p_src[i] += p_inc[i];
    }
#pragma unroll
    for (int i = 0; i < NCHW; ++i) {
      if constexpr (GLDS) glds16(w_src[i], wbase + (i * NW + w) * 1024);
      else stage_regs[NCHP + i] = *reinterpret_cast<const uint4*>(w_src[i]);
      if (w_row[i]) w_src[i] += BK;
    }
    if (++kc_in_tap == kc_per_tap) { kc_in_tap = 0; ++tap_cur; }
  };
  auto commit = [&](int buf) {  // register-staged path only
    if constexpr (!GLDS) {
      unsigned char* pbase = smem + buf * STAGE_BYTES;
      unsigned char* wbase = pbase + PTILE_BYTES;
#pragma unroll
      for (int i = 0; i < NCHP; ++i)
        *reinterpret_cast<uint4*>(pbase + (i * NW + w) * 1024 + lane * 16) = stage_regs[i];
#pragma unroll
      for (int i = 0; i < NCHW; ++i)
        *reinterpret_cast<uint4*>(wbase + (i * NW + w) * 1024 + lane * 16) = stage_regs[NCHP + i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wr = w / WM;  // which 64-channel slice of the tile
  const int wc = w % WM;  // which 64-pixel slice of the tile
  const int fr = lane & 15;
  const int fg = lane >> 4;

  auto compute = [&](int buf) {
    const unsigned char* pbase = smem + buf * STAGE_BYTES;
    const unsigned char* wbase = pbase + PTILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      s16x8 wf[4], pf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int nrow = wr * 64 + t * 16 + fr;
        const int prow = wc * 64 + t * 16 + fr;
        const int kc = ks * 4 + fg;
        wf[t] = *reinterpret_cast<const s16x8*>(wbase + nrow * (BK * 2) + ((kc ^ swz<BK>(nrow)) << 4));
        pf[t] = *reinterpret_cast<const s16x8*>(pbase + prow * (BK * 2) + ((kc ^ swz<BK>(prow)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], pf[j], acc[i][j], 0, 0, 0);
    }
#ifdef U2_IGLP
    __builtin_amdgcn_iglp_opt(U2_IGLP - 1);
#endif
  };

  if constexpr (GLDS) {
    // NST-deep LDS ring: tile kt+NST-1 is in flight while tile kt is multiplied.  LDS-DMA completion is tracked
    // with counted vmcnt (each tile is LPT global_load_lds per thread, retired in order); a raw s_barrier (not
    // __syncthreads, which would drain vmcnt to 0) publishes tile kt and frees the slot of tile kt-1.
    constexpr int LPT = NCHP + NCHW;
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0)
      if (s0 < nk) issue(s0, s0);
    for (int kt = 0; kt < nk; ++kt) {
      const int ahead = min(nk, kt + NST - 1) - (kt + 1);
      if (NST > 3 && ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPT) : "memory");
      else if (NST > 2 && ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + NST - 1 < nk) issue(kt + NST - 1, (kt + NST - 1) % NST);
      compute(kt % NST);
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      issue(kt, 0);
      __syncthreads();
      commit(0);
      __syncthreads();
      compute(0);
    }
  }

  // ---- epilogue: accumulators -> LDS (bf16, [pixel][channel]) -> coalesced 16-byte stores ----
  __syncthreads();
  bf16_t* otile = reinterpret_cast<bf16_t*>(smem);
  float* red = reinterpret_cast<float*>(smem + TM * OPITCH * 2);  // [2][NW][TN] floats
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int nl = wr * 64 + i * 16 + fg * 4;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    if (a.bias) {
      const int n = n0 + nl;
      b0 = (n + 0 < a.N) ? a.bias[n + 0] : 0.f;
      b1 = (n + 1 < a.N) ? a.bias[n + 1] : 0.f;
      b2 = (n + 2 < a.N) ? a.bias[n + 2] : 0.f;
      b3 = (n + 3 < a.N) ? a.bias[n + 3] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ml = wc * 64 + j * 16 + fr;
      float v0 = acc[i][j][0] + b0, v1 = acc[i][j][1] + b1, v2 = acc[i][j][2] + b2, v3 = acc[i][j][3] + b3;
      if (a.relu && !a.accumulate) {
        v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
      }
      uint2 pk;
      pk.x = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
      pk.y = (uint32_t)f2bf(v2) | ((uint32_t)f2bf(v3) << 16);
      *reinterpret_cast<uint2*>(otile + ml * OPITCH + nl) = pk;
    }
  }
  __syncthreads();

  constexpr int CPO = TN / 8;           // 16-byte chunks per output row
  constexpr int RPI = NT / CPO;         // rows covered per iteration
  const int cchunk = tid % CPO;
  const int rbase = tid / CPO;
  float s[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
  const bool vec_ok = ((a.out_ld & 7) == 0) && (n0 + cchunk * 8 + 8 <= a.N);
  const bool nt_out = !a.accumulate && (size_t)a.M * a.out_ld * 2 > ((size_t)160 << 20) && (a.abl & 8);
#pragma unroll
  for (int it = 0; it < TM / RPI; ++it) {
    const int row = it * RPI + rbase;
    const int m = m0 + row;
    if (m >= a.M) continue;
    uint4 v = *reinterpret_cast<const uint4*>(otile + row * OPITCH + cchunk * 8);
    bf16_t e8[8];
    *reinterpret_cast<uint4*>(e8) = v;
    size_t orow = (size_t)m;
    if (a.remap_out) {
      const int hw = a.Hout * a.Wout;
      const int img = m / hw;
      const int rem = m - img * hw;
      const int qy = rem / a.Wout;
      const int qx = rem - qy * a.Wout;
      orow = ((size_t)img * a.Hfull + qy * a.out_sy + a.out_y0) * a.Wfull + qx * a.out_sx + a.out_x0;
    }
    bf16_t* dst = a.out + orow * a.out_ld + n0 + cchunk * 8;
    if (a.accumulate) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (n0 + cchunk * 8 + e < a.N) {
          float f = bf2f(e8[e]) + bf2f(dst[e]);
          if (a.relu) f = fmaxf(f, 0.f);
          e8[e] = f2bf(f);
        }
      }
    }
    if (a.stats) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = bf2f(e8[e]);
        s[e] += f;
        ss[e] += f * f;
      }
    }
    if (vec_ok) {
      if (nt_out) {  // outputs far beyond the MALL size: do not let them evict what the next layer can still reuse
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
        const uint4 v = *reinterpret_cast<const uint4*>(e8);
        u32x4_t t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<u32x4_t*>(dst));
      } else {
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(e8);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (n0 + cchunk * 8 + e < a.N) dst[e] = e8[e];
    }
  }
  if (a.stats) {
    // lanes sharing (lane % CPO) hold partial sums of the same 8 channels
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o = CPO; o < 64; o <<= 1) {
        s[e] += __shfl_xor(s[e], o, 64);
        ss[e] += __shfl_xor(ss[e], o, 64);
      }
    }
    if (lane < CPO) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(0 * NW + w) * TN + lane * 8 + e] = s[e];
        red[(1 * NW + w) * TN + lane * 8 + e] = ss[e];
      }
    }
    __syncthreads();
    if (tid < TN && n0 + tid < a.N) {
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) {
        t0 += red[(0 * NW + ww) * TN + tid];
        t1 += red[(1 * NW + ww) * TN + tid];
      }
      atomicAdd(a.stats + n0 + tid, t0);
      atomicAdd(a.stats + a.N + n0 + tid, t1);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 tile, 8 waves (4 along channels x 2 along pixels, 64 x 128 per wave), for the wide layers.
//
// Why a second structure: with 64 x 64 wave tiles a K step needs 16 ds_read_b128 per 32 MFMA and a barrier pair
// per K tile; here a wave needs 12 reads per 32 MFMA, the reduction advances in half K tiles (32 channels) through FOUR
// LDS buffers (global_load_lds runs four half tiles = two K tiles ahead, counted vmcnt(8), never drained in the loop),
// there is ONE raw barrier per half tile, and the fragments of the next phase are read from LDS before the MFMAs of
// the current one are issued (register double buffering), so LDS latency hides behind the wave's own MFMA clusters.
//
//   phase A(h): read W fragments (channels 32-63 of the wave) of half tile h; stage 2nd half of half tile h+3; 16 MFMA
//   phase B(h): vmcnt(8), barrier [publishes h+1, frees buffer h]; read W (channels 0-31) + 8 pixel fragments of h+1;
//               stage 1st half of half tile h+4 into the freed buffer; 16 MFMA
// ------------------------------------------------------------------------------------------------
constexpr int C256_THREADS = 512;
constexpr int C256_HALF_BYTES = 256 * 64;                       // one operand, one half K tile: 256 rows x 32 bf16
constexpr int C256_BUF_BYTES = 2 * C256_HALF_BYTES;             // pixels + weights
constexpr int C256_OPITCH = 256 + 8;
constexpr int C256_LDS_BYTES = 256 * C256_OPITCH * 2 + 2 * 8 * 256 * 4;  // epilogue staging + stats scratch (151 KB)
static_assert(C256_LDS_BYTES >= 4 * C256_BUF_BYTES, "ring must fit in the epilogue allocation");

template <bool STAGGER>
__global__ __launch_bounds__(C256_THREADS, 2) void conv_igemm256_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = a.tiles_m * a.tiles_n;
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int tile_m = tile / a.tiles_n;
  const int tile_n = tile - tile_m * a.tiles_n;
  const int m0 = tile_m * 256;
  const int n0 = tile_n * 256;

  // ---- staging bookkeeping: per half tile a thread moves 2 pixel chunks and 2 weight chunks (rows fixed) ----
  int p_img[2], p_by[2], p_bx[2];
  bool p_ok[2];
  const bf16_t* w_src[2];  // weights are [n][tap][C]: consecutive half tiles are consecutive 32-channel slabs, across taps too
  int w_inc[2];
  const int row_in = lane >> 2;
  const int cc = (lane & 3) ^ swz<32>(row_in);
  {
    const int hw = a.Hout * a.Wout;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (i * 8 + w) * 16 + row_in;
      const int m = m0 + row;
      p_ok[i] = m < a.M;
      const int mm = p_ok[i] ? m : 0;
      const int img = mm / hw;
      const int rem = mm - img * hw;
      const int oy = rem / a.Wout;
      p_img[i] = img;
      p_by[i] = oy * a.mul;
      p_bx[i] = (rem - oy * a.Wout) * a.mul;
      const int n = n0 + row;
      w_src[i] = (n < a.N) ? a.wt + (size_t)n * ((size_t)a.wt_taps * a.C) + cc * 8 : a.zero;
      w_inc[i] = (n < a.N) ? 32 : 0;
    }
  }
  const int kh_per_tap = a.C >> 5;      // half tiles (32 channels) per filter tap
  const int nkh = a.ntaps * kh_per_tap; // even (C % 64 == 0), >= 4
  const bf16_t* p_src[2];
  int p_inc[2];
  int kc_in_tap = 0, reg_kh = 0, reg_kw = 0;
  // Staging order is P(0) W(0) P(1) W(1) ...: the pixel half carries the (branchy, once per tap) pointer set-up and is
  // issued right after a barrier, where no LDS read is outstanding; the weight half is straight-line code.
  auto stage_pixels = [&](int hb) {
    if (kc_in_tap == 0) {
      const int dy = reg_kh - a.pad_h, dx = reg_kw - a.pad_w;
      if (++reg_kw == a.KW) { reg_kw = 0; ++reg_kh; }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int sy = p_by[i] + dy, sx = p_bx[i] + dx;
        const bool ok = p_ok[i] && sy >= 0 && sx >= 0 && sy < a.Hin && sx < a.Win;
        p_src[i] = ok ? a.in + ((size_t)(p_img[i] * a.Hin + sy) * a.Win + sx) * a.in_ld + cc * 8 : a.zero;
        p_inc[i] = ok ? 32 : 0;
      }
    }
    if (++kc_in_tap == kh_per_tap) kc_in_tap = 0;
    unsigned char* pbase = smem + hb * C256_BUF_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      glds16(p_src[i], pbase + (i * 8 + w) * 1024);
      p_src[i] += p_inc[i];
    }
  };
  auto stage_weights = [&](int hb) {
    unsigned char* wbase = smem + hb * C256_BUF_BYTES + C256_HALF_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      glds16(w_src[i], wbase + (i * 8 + w) * 1024);
      w_src[i] += w_inc[i];
    }
  };

  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wr = w >> 1;  // 64-channel slice
  const int wc = w & 1;   // 128-pixel slice
  const int fr = lane & 15;
  const int fg = lane >> 4;
  // the swizzle term has a period of 16 rows, so fragment t of a wave is fragment 0 + t * 1024 bytes (an immediate offset)
  const int wbase0 = C256_HALF_BYTES + (wr * 64 + fr) * 64 + ((fg ^ swz<32>(fr)) << 4);
  const int pbase0 = (wc * 128 + fr) * 64 + ((fg ^ swz<32>(fr)) << 4);
  auto ldw = [&](int hb, int t) { return *reinterpret_cast<const s16x8*>(smem + hb * C256_BUF_BYTES + wbase0 + t * 1024); };
  auto ldp = [&](int hb, int t) { return *reinterpret_cast<const s16x8*>(smem + hb * C256_BUF_BYTES + pbase0 + t * 1024); };

  s16x8 wfA[2], wfB[2], pf[8];
#define U2_C256_MFMA(I, WF, J)                                                                             \
  acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF, pf[J], acc[I][J], 0, 0, 0)
#define U2_BAR()                                                                                           \
  do {                                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    asm volatile("" ::: "memory");                                                                         \
    __builtin_amdgcn_s_barrier();                                                                          \
    asm volatile("" ::: "memory");                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
  } while (0)
  if constexpr (STAGGER) {
    // Two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run the same LOAD | MFMA | LOAD | MFMA sequence one
    // barrier apart, so that while one group issues LDS reads and LDS-DMA the other keeps the matrix pipe busy:
    //   L_A(h): read W frags (ch 0-31) + 8 pixel frags of half tile h; stage weights(h+2)
    //   M_A(h): 16 MFMA
    //   L_B(h): read W frags (ch 32-63); stage pixels(h+3)      [group 1: vmcnt for half tile h+1]
    //   M_B(h): 16 MFMA                                         [group 0: vmcnt for half tile h+1]
    // with a raw barrier after every part.  Half tile k's pixels are staged in L_B(k-3) and its weights in L_A(k-2),
    // into the buffer of half tile k-4, whose last reads (group 1's L_B(k-4)) retired two barriers earlier.
    const int grp = a.stagger_by_parity ? (w & 1) : (w >> 2);
    stage_pixels(0); stage_weights(0);
    stage_pixels(1); stage_weights(1);
    stage_pixels(2);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    U2_BAR();
    if (grp == 1) U2_BAR();  // the lag: group 1 starts one part later
#define U2_C256_HALF(H, SW, SP, VMC, NEXT)                                                                  \
  do {                                                                                                     \
    const int hb_ = (H) & 3;                                                                               \
    /* L_A */                                                                                              \
    wfA[0] = ldw(hb_, 0); wfA[1] = ldw(hb_, 1);                                                            \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) pf[j] = ldp(hb_, j);                                     \
    if (SW) stage_weights(((H) + 2) & 3);                                                                  \
    U2_BAR();                                                                                              \
    /* M_A */                                                                                              \
    __builtin_amdgcn_s_setprio(1);                                                                         \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) { U2_C256_MFMA(0, wfA[0], j); U2_C256_MFMA(1, wfA[1], j); } \
    __builtin_amdgcn_s_setprio(0);                                                                         \
    U2_BAR();                                                                                              \
    /* L_B */                                                                                              \
    wfB[0] = ldw(hb_, 2); wfB[1] = ldw(hb_, 3);                                                            \
    if (SP) stage_pixels(((H) + 3) & 3);                                                                   \
    if (NEXT && grp == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMC) : "memory");                       \
    U2_BAR();                                                                                              \
    /* M_B */                                                                                              \
    __builtin_amdgcn_s_setprio(1);                                                                         \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) { U2_C256_MFMA(2, wfB[0], j); U2_C256_MFMA(3, wfB[1], j); } \
    __builtin_amdgcn_s_setprio(0);                                                                         \
    if (NEXT && grp == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMC) : "memory");                       \
    U2_BAR();                                                                                              \
  } while (0)
    int h = 0;
    for (; h + 3 < nkh; ++h) U2_C256_HALF(h, true, true, 6, true);
    U2_C256_HALF(h, true, false, 4, true); ++h;    // h = nkh - 3: weights(nkh-1) are the last loads issued
    U2_C256_HALF(h, false, false, 0, true); ++h;   // h = nkh - 2
    U2_C256_HALF(h, false, false, 0, false);       // h = nkh - 1
#undef U2_C256_HALF
    if (grp == 0) U2_BAR();  // pairs with group 1's last barrier
  } else {
  // ---- prologue: P0 W0 P1 W1 P2 W2 P3 in flight (W3 follows in phase A of half tile 0), publish half tile 0 ----
  stage_pixels(0); stage_weights(0);
  stage_pixels(1); stage_weights(1);
  stage_pixels(2); stage_weights(2);
  stage_pixels(3);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  wfA[0] = ldw(0, 0); wfA[1] = ldw(0, 1);
#pragma unroll
  for (int j = 0; j < 8; ++j) pf[j] = ldp(0, j);

  // One half tile.  SW / SP: stage weights(h+3) / pixels(h+4); VMC: loads that may stay in flight at the barrier;
  // NEXT: half tile h+1 exists.  LDS reads are placed so that nothing waits on a read issued less than ~8 MFMAs
  // earlier: phase A starts with MFMAs on fragments loaded in the middle of the previous phase B, reloads pixel
  // fragments 4-7 and the second weight pair after its first four MFMAs, and phase B reloads pixel fragments 0-3
  // (for the next half tile) between its two MFMA groups.
#define U2_C256_HALF(H, SW, SP, VMC, NEXT)                                                                  \
  do {                                                                                                     \
    const int hb_ = (H) & 3, nb_ = ((H) + 1) & 3;                                                          \
    /* phase A */                                                                                          \
    __builtin_amdgcn_s_setprio(1);                                                                         \
    U2_C256_MFMA(0, wfA[0], 0); U2_C256_MFMA(1, wfA[1], 0); U2_C256_MFMA(0, wfA[0], 1); U2_C256_MFMA(1, wfA[1], 1); \
    __builtin_amdgcn_s_setprio(0);                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    pf[4] = ldp(hb_, 4); pf[5] = ldp(hb_, 5); pf[6] = ldp(hb_, 6); pf[7] = ldp(hb_, 7);                    \
    wfB[0] = ldw(hb_, 2); wfB[1] = ldw(hb_, 3);                                                            \
    if (SW) stage_weights(((H) + 3) & 3);                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                                         \
    U2_C256_MFMA(0, wfA[0], 2); U2_C256_MFMA(1, wfA[1], 2); U2_C256_MFMA(0, wfA[0], 3); U2_C256_MFMA(1, wfA[1], 3); \
    _Pragma("unroll") for (int j = 4; j < 8; ++j) { U2_C256_MFMA(0, wfA[0], j); U2_C256_MFMA(1, wfA[1], j); } \
    __builtin_amdgcn_s_setprio(0);                                                                         \
    /* phase B */                                                                                          \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMC) : "memory");                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
    __builtin_amdgcn_s_barrier();                                                                          \
    asm volatile("" ::: "memory");                                                                         \
    if (SP) stage_pixels(hb_);                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                                         \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) { U2_C256_MFMA(2, wfB[0], j); U2_C256_MFMA(3, wfB[1], j); } \
    __builtin_amdgcn_s_setprio(0);                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    if (NEXT) {                                                                                            \
      wfA[0] = ldw(nb_, 0); wfA[1] = ldw(nb_, 1);                                                          \
      pf[0] = ldp(nb_, 0); pf[1] = ldp(nb_, 1); pf[2] = ldp(nb_, 2); pf[3] = ldp(nb_, 3);                  \
    }                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                                         \
    _Pragma("unroll") for (int j = 4; j < 8; ++j) { U2_C256_MFMA(2, wfB[0], j); U2_C256_MFMA(3, wfB[1], j); } \
    __builtin_amdgcn_s_setprio(0);                                                                         \
  } while (0)

  int h = 0;
  for (; h + 4 < nkh; ++h) U2_C256_HALF(h, true, true, 8, true);
  U2_C256_HALF(h, true, false, 8, true); ++h;    // h = nkh - 4: weights(nkh-1) are the last loads issued
  U2_C256_HALF(h, false, false, 4, true); ++h;   // h = nkh - 3
  U2_C256_HALF(h, false, false, 0, true); ++h;   // h = nkh - 2
  U2_C256_HALF(h, false, false, 0, false);       // h = nkh - 1
#undef U2_C256_HALF
  }
#undef U2_C256_MFMA
#undef U2_BAR

  // ---- epilogue: accumulators -> LDS (bf16, [pixel][channel]) -> coalesced 16-byte stores ----
  __syncthreads();
  bf16_t* otile = reinterpret_cast<bf16_t*>(smem);
  float* red = reinterpret_cast<float*>(smem + 256 * C256_OPITCH * 2);  // [2][8][256] floats
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int nl = wr * 64 + i * 16 + fg * 4;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    if (a.bias) {
      const int n = n0 + nl;
      b0 = (n + 0 < a.N) ? a.bias[n + 0] : 0.f;
      b1 = (n + 1 < a.N) ? a.bias[n + 1] : 0.f;
      b2 = (n + 2 < a.N) ? a.bias[n + 2] : 0.f;
      b3 = (n + 3 < a.N) ? a.bias[n + 3] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ml = wc * 128 + j * 16 + fr;
      float v0 = acc[i][j][0] + b0, v1 = acc[i][j][1] + b1, v2 = acc[i][j][2] + b2, v3 = acc[i][j][3] + b3;
      if (a.relu && !a.accumulate) {
        v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
      }
      uint2 pk;
      pk.x = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
      pk.y = (uint32_t)f2bf(v2) | ((uint32_t)f2bf(v3) << 16);
      *reinterpret_cast<uint2*>(otile + ml * C256_OPITCH + nl) = pk;
    }
  }
  __syncthreads();
  const int cchunk = tid & 31;   // 32 chunks of 8 channels per output row
  const int rbase = tid >> 5;    // 16 rows per iteration
  float s[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
  const bool vec_ok = ((a.out_ld & 7) == 0) && (n0 + cchunk * 8 + 8 <= a.N);
  const bool nt_out = !a.accumulate && (size_t)a.M * a.out_ld * 2 > ((size_t)160 << 20) && (a.abl & 8);
  for (int it = 0; it < 16; ++it) {
    const int row = it * 16 + rbase;
    const int m = m0 + row;
    if (m >= a.M) continue;
    bf16_t e8[8];
    *reinterpret_cast<uint4*>(e8) = *reinterpret_cast<const uint4*>(otile + row * C256_OPITCH + cchunk * 8);
    bf16_t* dst = a.out + (size_t)m * a.out_ld + n0 + cchunk * 8;
    if (a.accumulate) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (n0 + cchunk * 8 + e < a.N) {
          float f = bf2f(e8[e]) + bf2f(dst[e]);
          if (a.relu) f = fmaxf(f, 0.f);
          e8[e] = f2bf(f);
        }
      }
    }
    if (a.stats) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = bf2f(e8[e]);
        s[e] += f;
        ss[e] += f * f;
      }
    }
    if (vec_ok) {
      if (nt_out) {  // outputs far beyond the MALL size: do not let them evict what the next layer can still reuse
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
        const uint4 v = *reinterpret_cast<const uint4*>(e8);
        u32x4_t t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<u32x4_t*>(dst));
      } else {
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(e8);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (n0 + cchunk * 8 + e < a.N) dst[e] = e8[e];
    }
  }
  if (a.stats) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {  // lanes l and l ^ 32 hold the same 8 channels
      s[e] += __shfl_xor(s[e], 32, 64);
      ss[e] += __shfl_xor(ss[e], 32, 64);
    }
    if (lane < 32) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(0 * 8 + w) * 256 + lane * 8 + e] = s[e];
        red[(1 * 8 + w) * 256 + lane * 8 + e] = ss[e];
      }
    }
    __syncthreads();
    if (tid < 256 && n0 + tid < a.N) {
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) {
        t0 += red[(0 * 8 + ww) * 256 + tid];
        t1 += red[(1 * 8 + ww) * 256 + tid];
      }
      atomicAdd(a.stats + n0 + tid, t0);
      atomicAdd(a.stats + a.N + n0 + tid, t1);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// wgrad: dW[n][tap][c] += sum_m dY[m][n] * X[src(m,tap)][c]    (fp32 atomics, split over pixels)
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
  const bf16_t* x;    // NHWC input of the forward conv (pixel pitch x_ld)
  const bf16_t* dy;   // [M][dy_ld] output gradient
  float* dw;          // fp32, accumulated atomically at dw[n * dw_sn + tap * dw_st + c * dw_sc], n < n_valid, c < c_valid
  long long dw_sn;
  int dw_st, dw_sc, n_valid, c_valid;
  const bf16_t* zero;
  int B, Hin, Win, C, x_ld;
  int Hout, Wout, N, dy_ld;
  int KH, KW, pad_h, pad_w, stride;
  int M, tiles_n, tiles_c, pix_per_wg;
  int tiles, xcd_group;  // tiles = tiles_n * tiles_c * taps; xcd_group: 1-D launch, all tiles of a pixel split on one XCD
  // Partial tiles instead of atomics (round 4): when `part` is set a work-group stores its fp32 tile, [c][n] with n fastest, at
  // part + (split * tiles + tile) * TILE * TILE and wgrad_reduce_kernel sums the splits into dw afterwards.  fp32 atomics retire
  // at ~0.25 T operations/s on this part whatever their scope (tests/native/atomics_bench: ~1 TB/s of partial sums) - the
  // 64 KB x 512 work-group epilogue of a small-map 1x1 layer was 30 of its 77 us - plain 16-byte stores + one pass that reads
  // the partials back from the MALL move the same bytes at 5+ TB/s.
  float* part;
};

constexpr int WP = 32;  // pixels per reduction step

// XOR swizzle of the 16-byte chunk index of the wgrad LDS image [32 px][16 chunks].  A half-wave of a
// ds_read_b64_tr_b16 touches pixels {4h..4h+3, 8+4h..8+4h+3} x one 32-byte chunk pair; mapping those 8 pixels to 8
// different chunk pairs (bits 1-3) makes the 8 x 32 B of a half-wave cover one 256-byte bank row exactly.
__device__ __forceinline__ int wg_swz(int pix) { return ((pix & 3) << 1) | (pix & 8); }

template <bool GLDS, bool TR, int RING = 2>
__global__ __launch_bounds__(256, 4) void conv_wgrad_kernel(const WgradArgs a) {
  constexpr int TILE_BYTES = WP * 128 * 2;  // 8 KB per operand per stage
  constexpr int STAGE_BYTES = 2 * TILE_BYTES;
  static_assert(RING == 2 || (GLDS && TR), "the deeper rings exist for the LDS-DMA + transposing-read form only");
  __shared__ __attribute__((aligned(16))) unsigned char smem[RING * STAGE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t = blockIdx.x, split = blockIdx.y;
  if (a.xcd_group) {
    // every (n-tile, c-tile, tap) of one pixel split streams the same pixels: keep them on one XCD so that its L2 serves
    // all but the first read (hardware sends work-group i to XCD i % 8)
    const int logical = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    split = logical / a.tiles;
    t = logical - split * a.tiles;
  }
  const int tile_c = t % a.tiles_c; t /= a.tiles_c;
  const int tile_n = t % a.tiles_n; t /= a.tiles_n;
  const int tap = t;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int n0 = tile_n * 128, c0 = tile_c * 128;
  const int mbeg = split * a.pix_per_wg;
  const int mend = min(a.M, mbeg + a.pix_per_wg);
  if (mbeg >= mend) return;
  const int nsteps = (mend - mbeg + WP - 1) / WP;
  const int hw = a.Hout * a.Wout;
  const bool direct = (a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad_h == 0 && a.pad_w == 0);

  // chunk assignment: instr i in {0,1}: linear chunk p = (i*4+w)*64 + lane; pix = p>>4, slot = p&15
  const int pix_in_instr = lane >> 4;
  const int slot = lane & 15;

  uint4 stage_regs[GLDS ? 1 : 4];
  // per-chunk pixel cursor, advanced by WP pixels per step with carries instead of divisions
  int cm[2], cimg[2], coy[2], cox[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pix = (i * 4 + w) * 4 + pix_in_instr;
    cm[i] = mbeg + pix;
    cimg[i] = cm[i] / hw;
    const int rem = cm[i] - cimg[i] * hw;
    coy[i] = rem / a.Wout;
    cox[i] = rem - coy[i] * a.Wout;
  }
  auto issue = [&](int step, int buf) {
    (void)step;
    unsigned char* ybase = smem + buf * STAGE_BYTES;
    unsigned char* xbase = ybase + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pix = (i * 4 + w) * 4 + pix_in_instr;
      const int cc = slot ^ wg_swz(pix);
      const int m = cm[i];
      const bool mok = m < mend;
      const bf16_t* ysrc = (mok && n0 + cc * 8 < a.N) ? a.dy + (size_t)m * a.dy_ld + n0 + cc * 8 : a.zero;
      const bf16_t* xsrc = a.zero;
      if (mok && c0 + cc * 8 < a.C) {
        if (direct) {
          xsrc = a.x + (size_t)m * a.x_ld + c0 + cc * 8;
        } else {
          const int sy = coy[i] * a.stride - a.pad_h + kh;
          const int sx = cox[i] * a.stride - a.pad_w + kw;
          if (sy >= 0 && sy < a.Hin && sx >= 0 && sx < a.Win)
            xsrc = a.x + ((size_t)(cimg[i] * a.Hin + sy) * a.Win + sx) * a.x_ld + c0 + cc * 8;
        }
      }
      if constexpr (GLDS) {
        glds16(ysrc, ybase + (i * 4 + w) * 1024);
        glds16(xsrc, xbase + (i * 4 + w) * 1024);
      } else {
        stage_regs[2 * i] = *reinterpret_cast<const uint4*>(ysrc);
        stage_regs[2 * i + 1] = *reinterpret_cast<const uint4*>(xsrc);
      }
      cm[i] += WP;
      if (!direct) {
        if (a.Wout >= WP) {
          cox[i] += WP;
          while (cox[i] >= a.Wout) { cox[i] -= a.Wout; if (++coy[i] == a.Hout) { coy[i] = 0; ++cimg[i]; } }
        } else {
          cimg[i] = cm[i] / hw;
          const int rem = cm[i] - cimg[i] * hw;
          coy[i] = rem / a.Wout;
          cox[i] = rem - coy[i] * a.Wout;
        }
      }
    }
  };
  auto commit = [&](int buf) {
    if constexpr (!GLDS) {
      unsigned char* ybase = smem + buf * STAGE_BYTES;
      unsigned char* xbase = ybase + TILE_BYTES;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        *reinterpret_cast<uint4*>(ybase + (i * 4 + w) * 1024 + lane * 16) = stage_regs[2 * i];
        *reinterpret_cast<uint4*>(xbase + (i * 4 + w) * 1024 + lane * 16) = stage_regs[2 * i + 1];
      }
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wr = w >> 1, wc = w & 1;
  const int fr = lane & 15, fg = lane >> 4;

  // Fragment for a 16-channel block starting at ch0: lane gets, for channel ch0+fr, the 8 pixels 8*fg .. 8*fg+7 of the
  // step.  LDS image: [pix][128 ch] bf16, 16-byte chunks XOR-swizzled by wg_swz(pix).  All byte offsets are loop
  // invariant and computed once.
  // The 16-channel block index q only occupies bits 1-2 of the 16-byte chunk number (disjoint from the bits set by
  // wr and the lane), so offset(q) = offset(0) ^ (q << 5): two base offsets per operand instead of a table.
  constexpr int NOFF = TR ? 2 : 8;
  int yoff0[NOFF], xoff0[NOFF];
#pragma unroll
  for (int e = 0; e < NOFF; ++e) {
    // TR: ds_read_b64_tr_b16 - lane fr of a 16-lane group supplies the address of 4 contiguous bf16
    // (pixel p0 + fr/4, channels ch0 + (fr%4)*4 ..+3) and receives channel ch0+fr of pixels p0..p0+3.
    const int pix = TR ? fg * 8 + e * 4 + (fr >> 2) : fg * 8 + e;
    const int chl = TR ? (fr & 3) * 4 : fr;
    const int chy = wr * 64 + chl, chx = wc * 64 + chl;
    yoff0[e] = pix * 256 + (((chy >> 3) ^ wg_swz(pix)) << 4) + (chy & 7) * 2;
    xoff0[e] = pix * 256 + (((chx >> 3) ^ wg_swz(pix)) << 4) + (chx & 7) * 2;
  }
  // GLDS + TR: the transposing reads are inline asm.  Through the builtin the compiler assumes every LDS read may alias
  // the LDS-DMA writes in flight and drains them (vmcnt(0)) right after they are issued, i.e. the next step's loads
  // never overlap this step's MFMAs.  Ordering is explicit instead: vmcnt(0) + barrier before a step is read (the only
  // loads in flight at that point are that step's), lgkmcnt(0) after the reads.
  constexpr bool ASM_TR = GLDS && TR;
  const unsigned lds0 = (unsigned)(size_t)U2_LDS_PTR(smem);
  auto load_frag = [&](const unsigned char* base, const int* off0, int q) -> s16x8 {
    s16x8 r;
    if constexpr (ASM_TR) {
      union { unsigned long long u[2]; s16x8 v; } rr;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned addr = lds0 + (unsigned)(base - smem) + (unsigned)(off0[h] ^ (q << 5));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(rr.u[h]) : "v"(addr) : "memory");
      }
      r = rr.v;
    } else if constexpr (TR) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(base + (off0[h] ^ (q << 5))));
        r[h * 4 + 0] = v[0]; r[h * 4 + 1] = v[1]; r[h * 4 + 2] = v[2]; r[h * 4 + 3] = v[3];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = *reinterpret_cast<const short*>(base + (off0[e] ^ (q << 5)));
    }
    return r;
  };

  // a wave whose 64 x 64 sub-tile lies outside the valid N x C block (64-channel layers fill a quarter of the tile) only
  // helps with the staging: its MFMAs would multiply the zero page
  const bool wave_active = (n0 + wr * 64 < a.n_valid) && (c0 + wc * 64 < a.c_valid);
  auto compute = [&](int buf) {
    if (!wave_active) return;
    const unsigned char* ybase = smem + buf * STAGE_BYTES;
    const unsigned char* xbase = ybase + TILE_BYTES;
    s16x8 yf[4], xf[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      yf[q] = load_frag(ybase, yoff0, q);
      xf[q] = load_frag(xbase, xoff0, q);
    }
    if constexpr (ASM_TR) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], xf[j], acc[i][j], 0, 0, 0);
  };

  if constexpr (GLDS && RING > 2) {
    // RING-deep LDS ring: the loads of steps st+1 .. st+RING-1 are in flight while step st is multiplied (a step is only 16
    // MFMAs per wave, ~0.1 us: with the two-buffer form every step waited out most of a memory round trip).  4 LDS-DMA
    // loads per thread per step, retired in order: counted vmcnt, raw barrier (__syncthreads would drain them).
#pragma unroll
    for (int s0 = 0; s0 < RING - 1; ++s0)
      if (s0 < nsteps) issue(s0, s0);
    int buf = 0;
    for (int st = 0; st < nsteps; ++st) {
      const int ahead = min(nsteps, st + RING - 1) - (st + 1);  // steps issued behind step st
      if (ahead >= 2) {
        if (RING > 3 && ahead >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else if (ahead == 1) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (st + RING - 1 < nsteps) issue(st + RING - 1, buf == 0 ? RING - 1 : buf - 1);
      compute(buf);
      buf = (buf + 1 == RING) ? 0 : buf + 1;
    }
  } else if constexpr (GLDS) {
    issue(0, 0);
    for (int st = 0; st < nsteps; ++st) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (st + 1 < nsteps) issue(st + 1, (st + 1) & 1);
      compute(st & 1);
    }
  } else {
    for (int st = 0; st < nsteps; ++st) {
      issue(st, 0);
      __syncthreads();
      commit(0);
      __syncthreads();
      compute(0);
    }
  }

  // D[i = n][j = c]: lane holds column c = fr, rows n = fg*4 + r
  const int T = a.KH * a.KW;
  if (a.part) {  // 16 B per lane: rows fg * 4 .. + 3 of column c are consecutive in the [c][n] partial tile
    const int tlin = (tap * a.tiles_n + tile_n) * a.tiles_c + tile_c;
    float* pt = a.part + ((size_t)split * a.tiles + tlin) * (128 * 128);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(pt + (wc * 64 + j * 16 + fr) * 128 + wr * 64 + i * 16 + fg * 4) = acc[i][j];
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c0 + wc * 64 + j * 16 + fr;
      if (c >= a.c_valid) continue;
      float* dst = a.dw + (size_t)tap * a.dw_st + (size_t)c * a.dw_sc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wr * 64 + i * 16 + fg * 4 + r;
        if (n < a.n_valid) atomicAdd(dst + n * a.dw_sn, acc[i][j][r]);
      }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad, 256(n) x 256(c) tile per tap, 8 waves of 64(n) x 128(c).  The 128 x 128 kernel re-streams its pixel range for every
// (n-tile, c-tile) pair - PMC: 829 MB fetched per launch against ~250 MB algorithmic - so for layers with >= 256 channels
// on both sides this halves (256 x 256 layers: both operands once per tap) the traffic that bounds it.  Each operand's
// [32 px][256 ch] step image is stored as two of the 128-channel images of the kernel above (same swizzle, same transposing
// reads); three-stage LDS ring, counted vmcnt, one raw barrier per step (one resident work-group per CU cannot hide a
// drained pipeline behind another group).
// ------------------------------------------------------------------------------------------------
constexpr int WG256_IMG = WP * 128 * 2;          // 8 KB: one [32 px][128 ch] image
constexpr int WG256_STAGE = 4 * WG256_IMG;       // dy lo / dy hi / x lo / x hi
#ifndef U2_WG256_PIPE
// 1: fragment reads half a step ahead of their MFMAs in a four-stage ring (round 6).  Built on the reading that the drained
// wait behind a step's 24 transposing reads is what holds the kernel at ~2 100 cycles per 1 024-cycle step - and measured EQUAL
// (tests/native/selftest bench2w, same box: fc1 0.340 -> 0.351 ms, res4 1x1 0.070 -> 0.071, 8192^3 890 TFLOP/s either way;
// profiles/r06_wgrad_halo_part.txt), so the read latency is not what the step waits for; fc1's 600 TFLOP/s is this kernel's
// 890 x the 196 tiles on 256 CUs.  Default 0: the round-2 loop (all reads, one wait, 32 MFMAs) in a three-stage ring.
#define U2_WG256_PIPE 0
#endif
#ifndef U2_WG256_STAGES
// LDS ring depth of the default loop (3 ... 5: 96 ... 160 KB, stages - 1 of them in flight behind the step).  Round 6, with the counters
// of fc1's launch in hand (38 % of the wave cycles in s_waitcnt, no LDS bank conflicts, hardly any wait for LDS issue): 3 / 4 / 5
// stages measure the same on fc1 (0.351-0.359 ms), the stride-16 1x1 layers (0.069-0.074) and 8192^3 (0.90-0.91 PFLOP/s) -
// tools/exp/wg256_stages_ab.sh - so it is not bytes in flight either.
#define U2_WG256_STAGES 3
#endif
constexpr int WG256_STAGES = U2_WG256_PIPE ? 4 : U2_WG256_STAGES;

__global__ __launch_bounds__(512, 2) void conv_wgrad256_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t = blockIdx.x, split = blockIdx.y;
  if (a.xcd_group) {
    const int logical = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    split = logical / a.tiles;
    t = logical - split * a.tiles;
  }
  const int tile_c = t % a.tiles_c; t /= a.tiles_c;
  const int tile_n = t % a.tiles_n; t /= a.tiles_n;
  const int tap = t;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int n0 = tile_n * 256, c0 = tile_c * 256;
  const int mbeg = split * a.pix_per_wg;
  const int mend = min(a.M, mbeg + a.pix_per_wg);
  if (mbeg >= mend) return;
  const int nsteps = (mend - mbeg + WP - 1) / WP;
  const int hw = a.Hout * a.Wout;
  const bool direct = (a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad_h == 0 && a.pad_w == 0);
  // a "fully connected" conv (one output position per image, no padding: fc1's 7 x 7): pixel m reads input row (m, kh, kw) - no
  // coordinates to carry (the general path below re-derives them with two integer divisions per step when Wout < 32)
  const bool fc = !direct && a.Hout == 1 && a.Wout == 1 && a.pad_h == 0 && a.pad_w == 0;
  const size_t fc_pitch = (size_t)a.Hin * a.Win * a.x_ld, fc_off = ((size_t)kh * a.Win + kw) * a.x_ld;

  // staging: wave w moves pixels 4w .. 4w+3 of the step (16 chunks each) of all four images
  const int pix = w * 4 + (lane >> 4);
  const int cc = (lane & 15) ^ wg_swz(pix);
  int cm = mbeg + pix, cimg, coy, cox;
  {
    cimg = cm / hw;
    const int rem = cm - cimg * hw;
    coy = rem / a.Wout;
    cox = rem - coy * a.Wout;
  }
  auto issue = [&](int buf) {
    unsigned char* base = smem + buf * WG256_STAGE;
    const bool mok = cm < mend;
    const bf16_t* xrow = nullptr;
    if (mok) {
      if (direct) {
        xrow = a.x + (size_t)cm * a.x_ld;
      } else if (fc) {
        xrow = a.x + (size_t)cm * fc_pitch + fc_off;
      } else {
        const int sy = coy * a.stride - a.pad_h + kh;
        const int sx = cox * a.stride - a.pad_w + kw;
        if (sy >= 0 && sy < a.Hin && sx >= 0 && sx < a.Win) xrow = a.x + ((size_t)(cimg * a.Hin + sy) * a.Win + sx) * a.x_ld;
      }
    }
#pragma unroll
    for (int im = 0; im < 2; ++im) {
      const int nch = n0 + im * 128 + cc * 8, cch = c0 + im * 128 + cc * 8;
      const bf16_t* ysrc = (mok && nch < a.N) ? a.dy + (size_t)cm * a.dy_ld + nch : a.zero;
      const bf16_t* xsrc = (xrow && cch < a.C) ? xrow + cch : a.zero;
      glds16(ysrc, base + im * WG256_IMG + w * 1024);
      glds16(xsrc, base + (2 + im) * WG256_IMG + w * 1024);
    }
    cm += WP;
    if (!direct && !fc) {
      if (a.Wout >= WP) {
        cox += WP;
        while (cox >= a.Wout) { cox -= a.Wout; if (++coy == a.Hout) { coy = 0; ++cimg; } }
      } else {
        cimg = cm / hw;
        const int rem = cm - cimg * hw;
        coy = rem / a.Wout;
        cox = rem - coy * a.Wout;
      }
    }
  };

  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wr = w >> 1, wc = w & 1;  // wr: 64-channel slice of dy (image wr >> 1, half wr & 1); wc: x image (128 channels)
  const int fr = lane & 15, fg = lane >> 4;
  // transposing-read offsets inside one image (see conv_wgrad_kernel); the 16-channel block q is offset ^ (q << 5)
  int yoff0[2], xoff0[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int px = fg * 8 + e * 4 + (fr >> 2);
    const int chl = (fr & 3) * 4;
    const int chy = (wr & 1) * 64 + chl;
    yoff0[e] = (wr >> 1) * WG256_IMG + px * 256 + (((chy >> 3) ^ wg_swz(px)) << 4) + (chy & 7) * 2;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int chx = hh * 64 + chl;
      xoff0[hh][e] = (2 + wc) * WG256_IMG + px * 256 + (((chx >> 3) ^ wg_swz(px)) << 4) + (chx & 7) * 2;
    }
  }
  // The transposing reads are issued as inline asm: for the builtin the compiler assumes that an LDS read may alias the
  // LDS-DMA writes still in flight and drains them (s_waitcnt vmcnt(0)) before every step, which turns the ring into a
  // synchronous load.  Ordering is explicit instead: counted vmcnt + barrier before the reads, lgkmcnt(0) after them.
  const unsigned lds0 = (unsigned)(size_t)U2_LDS_PTR(smem);
  auto load_frag = [&](unsigned base, const int* off0, int q) -> s16x8 {
    union { unsigned long long u[2]; s16x8 v; } r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned addr = base + (unsigned)(off0[h] ^ (q << 5));
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r.u[h]) : "v"(addr) : "memory");
    }
    return r.v;
  };
  auto compute = [&](int buf) {
    const unsigned base = lds0 + buf * WG256_STAGE;
    s16x8 yf[4], xf[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      yf[q] = load_frag(base, yoff0, q);
      xf[q] = load_frag(base, xoff0[0], q);
      xf[4 + q] = load_frag(base, xoff0[1], q);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], xf[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

#if U2_WG256_PIPE
  // Round 6: fragment reads one half step ahead of their MFMAs in a four-stage ring.  A step is two batches of 16 MFMAs per
  // wave (dY fragments x the first / the second 64 channels of x); the second batch's x fragments are requested in front of
  // the first batch, the NEXT step's dY and first-batch fragments in front of the second one - behind the step's barrier,
  // which sits between the batches -, so every transposing read has 16 MFMAs (256 cycles of the matrix pipe) to come back and
  // the pipe has work queued while the waves meet at the barrier.  All waits are counted (LDS returns in order; no scalar loads
  // inside the loop - checked in the generated code).  Results identical to the default loop, time too (see U2_WG256_PIPE).
  union Frag { unsigned long long u[2]; s16x8 v; };
  Frag yA[4], yB[4], xl[4], xh[4];
#define U2_W_RD(F, BASE, OFF, Q)                                                                                        \
  do {                                                                                                                  \
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(F.u[0]) : "v"((BASE) + (unsigned)((OFF)[0] ^ ((Q) << 5))) : "memory"); \
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(F.u[1]) : "v"((BASE) + (unsigned)((OFF)[1] ^ ((Q) << 5))) : "memory"); \
  } while (0)
#define U2_W_TIE(F) "+v"(F[0].u[0]), "+v"(F[0].u[1]), "+v"(F[1].u[0]), "+v"(F[1].u[1]), "+v"(F[2].u[0]), "+v"(F[2].u[1]), "+v"(F[3].u[0]), "+v"(F[3].u[1])
#define U2_W_STEP(YC, YN)                                                                                      \
  do {                                                                                                         \
    const unsigned sb = lds0 + (unsigned)(buf * WG256_STAGE);                                                  \
    const unsigned sn = lds0 + (unsigned)(((buf + 1) & 3) * WG256_STAGE);                                      \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) U2_W_RD(xh[q], sb, xoff0[1], q);                             \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                              \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                            \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(YC[i].v, xl[j].v, acc[i][j], 0, 0, 0);             \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    if (st + 1 < nsteps) {                                                                                     \
      /* stage st + 1 landed (stage st + 2 may be in flight); every wave has left stage st - 1 */              \
      if (st + 2 < nsteps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                    \
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                    \
      __builtin_amdgcn_s_barrier();                                                                            \
      asm volatile("" ::: "memory");                                                                           \
      if (st + 3 < nsteps) issue((buf + 3) & 3);                                                               \
    }                                                                                                          \
    /* The next step's dY fragments, the wait for xh (lgkmcnt counts to 15: not all 16 requests in front of it), then the  \
       next step's first x fragments.  The last step requests them too (stale ring contents, never multiplied): no        \
       register written by a read in flight meets a compiler-made copy at a control-flow join, and the counts stay exact */ \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) U2_W_RD(YN[q], sn, yoff0, q);                                \
    asm volatile("s_waitcnt lgkmcnt(8)" : U2_W_TIE(xh)::"memory");                                             \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) U2_W_RD(xl[q], sn, xoff0[0], q);                             \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                              \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                            \
        acc[i][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(YC[i].v, xh[j].v, acc[i][4 + j], 0, 0, 0);     \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" : U2_W_TIE(YN), U2_W_TIE(xl)::"memory");                               \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)
  static_assert(WG256_STAGES == 4, "the ring indices of the pipelined loop are written for four stages");
  (void)compute;
  issue(0);
  if (nsteps > 1) issue(1);
  if (nsteps > 2) issue(2);
  if (nsteps > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (nsteps == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    U2_W_RD(yA[q], lds0, yoff0, q);
    U2_W_RD(xl[q], lds0, xoff0[0], q);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : U2_W_TIE(yA), U2_W_TIE(xl)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  {
    int buf = 0, st = 0;
    for (;;) {
      U2_W_STEP(yA, yB);
      ++st; buf = (buf + 1) & 3;
      if (st >= nsteps) break;
      U2_W_STEP(yB, yA);
      ++st; buf = (buf + 1) & 3;
      if (st >= nsteps) break;
    }
  }
#undef U2_W_STEP
#undef U2_W_TIE
#undef U2_W_RD
#else
  // ring of WG256_STAGES stages: steps st+1 .. st+STAGES-1 are in flight while step st is multiplied (4 LDS-DMA loads per thread
  // per step); at the barrier of step st everything but the stages behind it has landed
  static_assert(WG256_STAGES >= 3 && WG256_STAGES <= 5, "ring depth");
#pragma unroll
  for (int s0 = 0; s0 < WG256_STAGES - 1; ++s0)
    if (s0 < nsteps) issue(s0);
  for (int st = 0; st < nsteps; ++st) {
    const int behind = min(WG256_STAGES - 2, nsteps - 1 - st);   // stages issued behind step st that may still be in flight
    if (behind >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (behind == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (behind == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (st + WG256_STAGES - 1 < nsteps) issue((st + WG256_STAGES - 1) % WG256_STAGES);
    compute(st % WG256_STAGES);
  }

#endif

  // D[i = n][j = c]: lane holds column c = fr, rows n = fg*4 + r
  if (a.part) {
    const int tlin = (tap * a.tiles_n + tile_n) * a.tiles_c + tile_c;
    float* pt = a.part + ((size_t)split * a.tiles + tlin) * (256 * 256);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        *reinterpret_cast<f32x4*>(pt + (wc * 128 + (j >> 2) * 64 + (j & 3) * 16 + fr) * 256 + wr * 64 + i * 16 + fg * 4) = acc[i][j];
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + wc * 128 + (j >> 2) * 64 + (j & 3) * 16 + fr;
      if (c >= a.c_valid) continue;
      float* dst = a.dw + (size_t)tap * a.dw_st + (size_t)c * a.dw_sc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wr * 64 + i * 16 + fg * 4 + r;
        if (n < a.n_valid) atomicAdd(dst + n * a.dw_sn, acc[i][j][r]);
      }
    }
}

// Sums the partial tiles of a weight-gradient launch over its pixel splits and adds the result into dw (single owner per
// element: plain read-modify-write).  TILE x TILE partials, [c][n] with n fastest; a work-group takes 8 columns c of one tile:
// 32 lanes x 16 B per column and split (coalesced), the sums go through LDS so that the dw accesses run along c (the contiguous
// direction of both gradient layouts).
template <int TILE>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int splits, int tiles, int per_group,
                                                           WgradArgs a) {
  constexpr int NQ = TILE / 4;            // 16-byte groups per column
  constexpr int CPW = 256 / NQ;           // columns per pass of the work-group (8 for 128, 4 for 256)
  constexpr int COLS = 8;                 // columns per work-group
  __shared__ float sm[COLS][TILE + 1];
  const int tlin = blockIdx.x, slab = blockIdx.y;
  int t = tlin;
  const int tile_c = t % a.tiles_c; t /= a.tiles_c;
  const int tile_n = t % a.tiles_n; t /= a.tiles_n;
  const int tap = t;
  const int tid = threadIdx.x;
  const int nq = tid % NQ, ci = tid / NQ;
#pragma unroll
  for (int cc = 0; cc < COLS / CPW; ++cc) {
    const int c_local = slab * COLS + cc * CPW + ci;
    const float* src = part + (size_t)tlin * (TILE * TILE) + c_local * TILE + nq * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    // blockIdx.z = group of `per_group` consecutive splits (launches with few tiles and hundreds of splits: a work-group per
    // (tile, 8 columns) alone would be 32 work-groups reading 2 MB each); several groups meet in dw through atomics
    int sp = blockIdx.z * per_group;
    const int sp_end = min(splits, sp + per_group);
    const size_t sstride = (size_t)tiles * (TILE * TILE);
    // round 6: eight loads in flight per thread (the partials come back from the MALL at ~1 us per dependent round trip; with
    // pairs a work-group summing 16 splits spent eight round trips on 2 KB of data per thread)
    for (; sp + 7 < sp_end; sp += 8) {
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)(sp + k) * sstride));
      s0 += (v[0] + v[2]) + (v[4] + v[6]);
      s1 += (v[1] + v[3]) + (v[5] + v[7]);
    }
    for (; sp + 1 < sp_end; sp += 2) {   // two independent chains: the loads of a pair are in flight together
      s0 += *reinterpret_cast<const f32x4*>(src + (size_t)sp * sstride);
      s1 += *reinterpret_cast<const f32x4*>(src + (size_t)(sp + 1) * sstride);
    }
    if (sp < sp_end) s0 += *reinterpret_cast<const f32x4*>(src + (size_t)sp * sstride);
    s0 += s1;
    float* row = sm[cc * CPW + ci] + nq * 4;
    row[0] = s0[0]; row[1] = s0[1]; row[2] = s0[2]; row[3] = s0[3];
  }
  __syncthreads();
  const int c_rel = tid & (COLS - 1);
  const int c = tile_c * TILE + slab * COLS + c_rel;
  if (c >= a.c_valid) return;
  float* dst = a.dw + (size_t)tap * a.dw_st + (size_t)c * a.dw_sc;
  for (int nl = tid / COLS; nl < TILE; nl += 256 / COLS) {
    const int n = tile_n * TILE + nl;
    if (n < a.n_valid) {
      if (gridDim.z == 1) dst[n * a.dw_sn] += sm[c_rel][nl];
      else atomicAdd(dst + n * a.dw_sn, sm[c_rel][nl]);
    }
  }
}

constexpr size_t WG_SCRATCH_LIMIT = (size_t)96 << 20;
float* wgrad_scratch(hipStream_t s, size_t need_bytes) {
  return (float*)u2conv::scratch_get(2, s, need_bytes, WG_SCRATCH_LIMIT, false);
}

__device__ __attribute__((aligned(256))) bf16_t g_zero_page[128];
}  // namespace
namespace u2conv {
int g_last_conv_kernel = 0;

namespace {
struct ScratchEnt { int device; hipStream_t stream; int kind; void* buf; size_t cap; };
ScratchEnt g_scratch[128];
int g_scratch_used = 0;
std::mutex g_scratch_mu;
// blocks replaced by a larger one: another host thread (main / autograd) may hold the old pointer between scratch_get's return
// and its launch, which no stream synchronisation here can see - so an outgrown block is only FREED by u2_release_scratch().
// Growth is geometric (x 1.5) under a per-kind limit: the retired blocks of an entry sum to less than twice its final size.
struct RetiredEnt { int device; void* buf; };
std::vector<RetiredEnt> g_retired;
}  // namespace

void* scratch_get(int kind, hipStream_t s, size_t need_bytes, size_t limit_bytes, bool zero_on_alloc) {
  if (need_bytes == 0 || need_bytes > limit_bytes) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  ScratchEnt* e = nullptr;
  for (int i = 0; i < g_scratch_used; ++i)
    if (g_scratch[i].device == dev && g_scratch[i].stream == s && g_scratch[i].kind == kind) { e = &g_scratch[i]; break; }
  if (e && e->cap >= need_bytes) return e->buf;
  if (!e) {
    if (g_scratch_used == 128) return nullptr;
    e = &g_scratch[g_scratch_used];
    *e = ScratchEnt{dev, s, kind, nullptr, 0};
  }
  // grow: 1.5 x the old block at least (a sequence of slowly growing launches must not reallocate every time), 2 MB granules
  size_t cap = need_bytes > e->cap + e->cap / 2 ? need_bytes : e->cap + e->cap / 2;
  cap = (cap + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
  if (cap > limit_bytes) cap = limit_bytes;
  if (e->buf) {
    g_retired.push_back(RetiredEnt{dev, e->buf});   // freed by u2_release_scratch (see g_retired)
    e->buf = nullptr; e->cap = 0;
  }
  void* p = nullptr;
  if (hipMalloc(&p, cap) != hipSuccess) return nullptr;
  if (zero_on_alloc && hipMemset(p, 0, cap) != hipSuccess) { (void)hipFree(p); return nullptr; }
  e->buf = p; e->cap = cap;
  if (e == &g_scratch[g_scratch_used]) ++g_scratch_used;
  return p;
}
}  // namespace u2conv

extern "C" int u2_release_scratch(void) {
  int cur = 0;
  (void)hipGetDevice(&cur);
  std::lock_guard<std::mutex> lock(u2conv::g_scratch_mu);
  int freed = 0;
  for (int i = 0; i < u2conv::g_scratch_used; ++i) {
    u2conv::ScratchEnt& e = u2conv::g_scratch[i];
    if (!e.buf) continue;
    (void)hipSetDevice(e.device);
    (void)hipStreamSynchronize(e.stream);
    (void)hipFree(e.buf);
    ++freed;
  }
  u2conv::g_scratch_used = 0;
  for (const u2conv::RetiredEnt& r : u2conv::g_retired) {
    (void)hipSetDevice(r.device);
    (void)hipDeviceSynchronize();
    (void)hipFree(r.buf);
    ++freed;
  }
  u2conv::g_retired.clear();
  (void)hipSetDevice(cur);
  return freed;
}
extern "C" int u2_conv_last_kernel(void) { return u2conv::g_last_conv_kernel; }
namespace {

const bf16_t* zero_page_ptr() {
  static const bf16_t* p = nullptr;
  if (!p) {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_zero_page)) != hipSuccess) return nullptr;
    p = reinterpret_cast<const bf16_t*>(q);
  }
  return p;
}

}  // namespace

namespace {

// a caller that passes variant 0 (the product path always does) can be steered from the environment: tests and A/B runs
int env_variant(const char* name, int variant) {
  if (variant != 0) return variant;
  const char* e = getenv(name);
  return e ? (int)strtol(e, nullptr, 0) : 0;
}

int launch_conv(ConvArgs& a, int N, int C, int variant, hipStream_t s) {
  variant = env_variant("U2_CONV_VARIANT", variant);
  a.abl = (variant >> 18) & 63;  // measurement switches (conv_args.h); bit 3 = `nt` output stores, applies to every kernel below
  {  // 1x1 / stride 1 layers with <= 256 input channels on large maps: weights-resident streaming kernel (conv_stream.hip)
    const int rc = launch_conv_stream(a, N, C, variant, s);
    if (rc == 1) return 0;
    if (rc < 0) return rc;
  }
  {  // 3x3 / stride 1 / pad 1 layers on large maps: halo-staged kernel (conv_halo.hip)
    const int rc = launch_conv_halo(a, N, C, variant, s);
    if (rc == 1) return 0;
    if (rc < 0) return rc;
  }
  {  // persistent tile kernels (conv_tile.hip) first; 0 = shape / variant not served there
    const int rc = launch_conv_tile(a, N, C, variant, s);
    if (rc == 1) return 0;
    if (rc < 0) return rc;
  }
  // wide layers: 256 x 256 tile, 8 waves, four-deep half-K-tile ring (variant bit 8 forces it, bit 9 forbids it)
  const bool wide_ok = !a.remap_out && C % 64 == 0 && a.ntaps * (C / 32) >= 4;
  // measured win: deep reductions (K >= 1024) with at least two full waves of 256 x 256 tiles; short-K 1x1 layers lose
  const bool wide_auto = N % 256 == 0 && a.ntaps * C >= 1024 && (long long)a.M * N >= 256LL * 256 * 512;
  if (wide_ok && !(variant & 512) && ((variant & 256) || wide_auto)) {
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = (N + 255) / 256;
    static PerDeviceOnce attr_set;
    if (auto once_guard = attr_set.first()) {
      (void)hipFuncSetAttribute((const void*)conv_igemm256_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)conv_igemm256_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    a.stagger_by_parity = (variant & 2048) ? 1 : 0;
    g_last_conv_kernel = 256 + ((variant & 1024) ? 1024 : 0);
    if (variant & 1024)  // staggered two-group schedule
      hipLaunchKernelGGL(conv_igemm256_kernel<true>, dim3(a.tiles_m * a.tiles_n), dim3(C256_THREADS), C256_LDS_BYTES, s, a);
    else
      hipLaunchKernelGGL(conv_igemm256_kernel<false>, dim3(a.tiles_m * a.tiles_n), dim3(C256_THREADS), C256_LDS_BYTES, s, a);
    U2_CHECK_LAUNCH();
    return 0;
  }
  const bool narrow = N <= 64;  // 256 x 64 tile
  const bool big = !narrow && (variant & 8);  // 256 x 128 tile, 8 waves
  const int TM = (narrow || big) ? 256 : 128, TN = narrow ? 64 : 128;
  a.tiles_m = (a.M + TM - 1) / TM;
  a.tiles_n = (N + TN - 1) / TN;
  const dim3 grid(a.tiles_m * a.tiles_n), block(big ? 512 : 256);
  const bool glds = (variant & 1) == 0;
  // variant: bit0 register staging, bit2 force BK = 32, bit3 256x128 tile, bits 4-5 LDS ring depth override (0 = default)
  // Layers whose whole reduction is <= 256 deep (1x1 convs on <= 256 channels) are HBM-bound and never reach a steady
  // K loop: a small BK = 32 / 2-stage footprint (38 KB) keeps 4 work-groups per CU in flight instead of 2.
  const bool shallow = (long long)a.ntaps * C <= 256;  // K = 256: 440 vs 358 TFLOP/s on the 50x84 256->1024 layer
  const int bk = (C % 64 == 0 && !(variant & 4) && !shallow) ? 64 : 32;
  int nst = (variant >> 4) & 3;
  if (nst == 0) nst = (bk == 64 || shallow) ? 2 : 4;
  else nst += 1;  // 1 -> 2 stages, 2 -> 3, 3 -> 4
  if (!glds) nst = 2;
  size_t lds = (size_t)nst * (TM + TN) * bk * 2;
  const size_t epi = (size_t)TM * (TN + 8) * 2 + 2 * 8 * TN * 4;
  if (lds < epi) lds = epi;
  if (lds > 160 * 1024) return -3;
#define U2_LAUNCH_CONV(BK_, GL_, TM_, TN_, NST_)                                                                 \
  do {                                                                                                           \
    g_last_conv_kernel = 1000000 + BK_ * 10000 + (TM_ / 64) * 1000 + (TN_ / 64) * 100 + NST_ * 10 + (GL_ ? 1 : 0); \
    static PerDeviceOnce attr_set;                                                                                \
    if (auto once_guard = attr_set.first()) {                                                                                             \
      (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<BK_, GL_, TM_, TN_, NST_>,                        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                         \
    }                                                                                                            \
    hipLaunchKernelGGL((conv_igemm_kernel<BK_, GL_, TM_, TN_, NST_>), grid, block, lds, s, a);                   \
  } while (0)
#define U2_PICK_NST(BK_, TM_, TN_)                                                                               \
  do {                                                                                                           \
    if (!glds) U2_LAUNCH_CONV(BK_, false, TM_, TN_, 2);                                                          \
    else if (nst == 2) U2_LAUNCH_CONV(BK_, true, TM_, TN_, 2);                                                   \
    else if (nst == 3) U2_LAUNCH_CONV(BK_, true, TM_, TN_, 3);                                                   \
    else U2_LAUNCH_CONV(BK_, true, TM_, TN_, 4);                                                                 \
  } while (0)
  if (narrow) {
    if (bk == 64) U2_PICK_NST(64, 256, 64); else U2_PICK_NST(32, 256, 64);
  } else if (big) {
    if (bk == 64) U2_PICK_NST(64, 256, 128); else U2_PICK_NST(32, 256, 128);
  } else {
    if (bk == 64) U2_PICK_NST(64, 128, 128); else U2_PICK_NST(32, 128, 128);
  }
#undef U2_PICK_NST
#undef U2_LAUNCH_CONV
  U2_CHECK_LAUNCH();
  return 0;
}

inline int floor_div(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// Data gradient of a strided conv: the output pixels whose parity class (y % div, x % div) meets no filter tap (three of the four
// classes of a 1x1 / stride-2 layer) are zero.  One fill launch over all such classes (bit py * div + px of `classes`) instead of
// a GEMM-shaped launch per class that multiplies nothing (round 5).
__global__ __launch_bounds__(256) void parity_zero_fill_kernel(bf16_t* __restrict__ out, long long pixels, int H, int W, int ld, int chunks,
                                                               int div, unsigned classes) {
  const long long total = pixels * chunks;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long p = i / chunks;
    const int c = (int)(i - p * chunks);
    const int x = (int)(p % W), y = (int)((p / W) % H);
    if ((classes >> ((y % div) * div + (x % div))) & 1u)
      *reinterpret_cast<uint4*>(out + (size_t)p * ld + c * 8) = make_uint4(0u, 0u, 0u, 0u);
  }
}

}  // namespace

extern "C" int u2_conv_igemm(const void* in, const void* wt, void* out, const float* bias, float* stats,
                             int B, int Hin, int Win, int C, int in_ld, int Hout, int Wout, int N, int out_ld,
                             int KH, int KW, int pad_h, int pad_w, int mul, int div, int relu, int accumulate,
                             int variant, void* stream) {
  if (C % 32 != 0 || (in_ld & 7) != 0 || KH * KW > 64 || (div > 1 && mul != 1)) return -1;
  if (B <= 0 || Hout <= 0 || Wout <= 0 || N <= 0) return 0;
  ConvArgs a;
  a.in = (const bf16_t*)in; a.wt = (const bf16_t*)wt; a.out = (bf16_t*)out; a.bias = bias; a.stats = stats;
  a.zero = zero_page_ptr();
  if (!a.zero) return -2;
  a.abl = 0; a.stagger_by_parity = 0; a.sk_ws = nullptr; a.sk_flags = nullptr;
  a.B = B; a.Hin = Hin; a.Win = Win; a.C = C; a.in_ld = in_ld;
  a.N = N; a.out_ld = out_ld; a.mul = mul; a.relu = relu; a.accumulate = accumulate;
  a.wt_taps = KH * KW;
  a.KW = KW; a.pad_h = pad_h; a.pad_w = pad_w;
  a.Hfull = Hout; a.Wfull = Wout;
  hipStream_t s = (hipStream_t)stream;
  if (div <= 1) {
    a.Hout = Hout; a.Wout = Wout; a.M = B * Hout * Wout;
    a.ntaps = KH * KW;
    for (int kh = 0; kh < KH; ++kh)
      for (int kw = 0; kw < KW; ++kw) {
        const int t = kh * KW + kw;
        a.tap_dy[t] = (short)(kh - pad_h); a.tap_dx[t] = (short)(kw - pad_w); a.tap_w[t] = (short)t;
        a.tap_pk[t] = ((kh - pad_h) & 0xff) | (((kw - pad_w) & 0xff) << 8) | (t << 16);
      }
    a.remap_out = 0; a.out_sy = a.out_sx = 1; a.out_y0 = a.out_x0 = 0;
    return launch_conv(a, N, C, variant, s);
  }
  // Data gradient of a stride-`div` conv: output pixels of parity class (py, px) only meet the taps with
  // (py - pad + kh) % div == 0, so each class is its own small stride-1 conv (1, 2, 2 and 4 taps for 3x3 / stride 2)
  // instead of 9 taps of which 3/4 would multiply the zero page.
  // classes without a tap: one fill launch (a class with a bias, statistics or an accumulating epilogue keeps its own launch)
  unsigned empty = 0;
  if (!bias && !stats && !accumulate && div * div <= 32 && (N & 7) == 0) {
    for (int py = 0; py < div; ++py)
      for (int px = 0; px < div; ++px) {
        bool any = false;
        for (int kh = 0; kh < KH && !any; ++kh)
          for (int kw = 0; kw < KW && !any; ++kw)
            any = ((((py - pad_h + kh) % div) + div) % div) == 0 && ((((px - pad_w + kw) % div) + div) % div) == 0;
        if (!any && py < Hout && px < Wout) empty |= 1u << (py * div + px);
      }
    if (empty) {
      const long long pixels = (long long)B * Hout * Wout;
      long long g = (pixels * (N / 8) + 255) / 256;
      if (g > 256 * 32) g = 256 * 32;
      hipLaunchKernelGGL(parity_zero_fill_kernel, dim3((unsigned)g), dim3(256), 0, s, a.out, pixels, Hout, Wout, out_ld, N / 8, div, empty);
      U2_CHECK_LAUNCH();
    }
  }
  for (int py = 0; py < div; ++py)
    for (int px = 0; px < div; ++px) {
      if ((empty >> (py * div + px)) & 1u) continue;
      const int Hq = (Hout - py + div - 1) / div, Wq = (Wout - px + div - 1) / div;
      if (Hq <= 0 || Wq <= 0) continue;
      int nt = 0;
      for (int kh = 0; kh < KH; ++kh) {
        const int vy = py - pad_h + kh;
        if (((vy % div) + div) % div) continue;
        for (int kw = 0; kw < KW; ++kw) {
          const int vx = px - pad_w + kw;
          if (((vx % div) + div) % div) continue;
          a.tap_dy[nt] = (short)floor_div(vy, div); a.tap_dx[nt] = (short)floor_div(vx, div);
          a.tap_w[nt] = (short)(kh * KW + kw);
          a.tap_pk[nt] = (a.tap_dy[nt] & 0xff) | ((a.tap_dx[nt] & 0xff) << 8) | ((kh * KW + kw) << 16);
          ++nt;
        }
      }
      a.ntaps = nt;
      a.Hout = Hq; a.Wout = Wq; a.M = B * Hq * Wq;
      a.remap_out = 1; a.out_sy = a.out_sx = div; a.out_y0 = py; a.out_x0 = px;
      const int rc = launch_conv(a, N, C, variant, s);
      if (rc) return rc;
    }
  return 0;
}

extern "C" int u2_conv1x1_bwd_fused(const void* x, const void* dy, const void* wt, void* dx, float* dw, int M, int C, int x_ld,
                                    int N, int dy_ld, int wt_ld, int dx_ld, int n_valid, int c_valid, long long dw_stride_n,
                                    int dw_stride_c, int variant, void* stream) {
  if (n_valid > N || c_valid > C) return -1;
  const bf16_t* zero = zero_page_ptr();
  if (!zero) return -2;
  if (M <= 0) return 0;
  variant = env_variant("U2_WDGRAD_VARIANT", variant);
  if (variant & 4) return 1;   // switched off (A/B runs): the caller takes the two separate launches
  const int rc = launch_wdgrad_stream((const bf16_t*)x, (const bf16_t*)dy, (const bf16_t*)wt, (bf16_t*)dx, dw, dw_stride_n,
                                      dw_stride_c, n_valid, c_valid, zero, M, C, x_ld, N, dy_ld, wt_ld, dx_ld, variant & 1,
                                      (variant >> 1) & 1, (hipStream_t)stream);
  if (rc == 1) return 0;
  return rc < 0 ? rc : 1;
}

extern "C" int u2_conv1x1_bwd_fused_bn(const void* x, const void* dz, const void* y, const float* k1, const float* k2,
                                       const float* k3, const void* wt, void* dx, float* dw, int M, int C, int x_ld, int N, int dy_ld,
                                       int wt_ld, int dx_ld, int n_valid, int c_valid, long long dw_stride_n, int dw_stride_c,
                                       int variant, void* stream) {
  if (n_valid > N || c_valid > C || !y || !k1 || !k2 || !k3) return -1;
  const bf16_t* zero = zero_page_ptr();
  if (!zero) return -2;
  if (M <= 0) return 0;
  variant = env_variant("U2_WDGRAD_VARIANT", variant);
  if (variant & 4) return 1;
  const int rc = launch_wdgrad_stream((const bf16_t*)x, (const bf16_t*)dz, (const bf16_t*)wt, (bf16_t*)dx, dw, dw_stride_n,
                                      dw_stride_c, n_valid, c_valid, zero, M, C, x_ld, N, dy_ld, wt_ld, dx_ld,
                                      (variant & 1) | ((variant >> 2) & 2) | ((variant >> 2) & 4), (variant >> 1) & 1, (hipStream_t)stream, (const bf16_t*)y,
                                      k1, k2, k3);
  if (rc == 1) return 0;
  return rc < 0 ? rc : 1;
}

extern "C" int u2_conv_wgrad(const void* x, const void* dy, float* dw, int B, int Hin, int Win, int C, int x_ld,
                             int Hout, int Wout, int N, int dy_ld, int KH, int KW, int pad_h, int pad_w,
                             int stride, int variant, void* stream) {
  return u2_conv_wgrad_into(x, dy, dw, B, Hin, Win, C, x_ld, Hout, Wout, N, dy_ld, KH, KW, pad_h, pad_w, stride, N, C,
                            (long long)KH * KW * C, C, 1, variant, stream);
}

extern "C" int u2_conv_wgrad_into(const void* x, const void* dy, float* dw, int B, int Hin, int Win, int C, int x_ld,
                                  int Hout, int Wout, int N, int dy_ld, int KH, int KW, int pad_h, int pad_w,
                                  int stride, int n_valid, int c_valid, long long dw_stride_n, int dw_stride_tap,
                                  int dw_stride_c, int variant, void* stream) {
  if ((C & 7) != 0 || (x_ld & 7) != 0 || (dy_ld & 7) != 0 || (N & 7) != 0) return -1;
  if (n_valid > N || c_valid > C) return -1;
  if (B <= 0 || Hout <= 0 || Wout <= 0) return 0;
  WgradArgs a;
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.dw = dw; a.zero = zero_page_ptr();
  if (!a.zero) return -2;
  a.dw_sn = dw_stride_n; a.dw_st = dw_stride_tap; a.dw_sc = dw_stride_c; a.n_valid = n_valid; a.c_valid = c_valid;
  a.B = B; a.Hin = Hin; a.Win = Win; a.C = C; a.x_ld = x_ld;
  a.Hout = Hout; a.Wout = Wout; a.N = N; a.dy_ld = dy_ld;
  a.KH = KH; a.KW = KW; a.pad_h = pad_h; a.pad_w = pad_w; a.stride = stride;
  a.M = B * Hout * Wout;
  // 256 x 256 tiles: only on request (variant bit 8).  Measured 516 vs 805 TFLOP/s on the 200x336 3x3 256->256 layer: with
  // one resident work-group per CU the transposing reads of a step are not hidden behind anything, while four resident
  // 128 x 128 groups hide them behind each other, which outweighs the halved operand traffic.
  variant = env_variant("U2_WGRAD_VARIANT", variant);
  // 1x1 / stride 1 over large maps: streaming kernel with the whole dW block in registers (wgrad_stream.hip); variant bit 18
  // forces it on any size, bit 19 forbids it (as does any of the per-tap kernel's own variant bits 0-11), bit 20: 8 pixel ranges
  // only (tests), bits 21-23: block configuration (0 = automatic), bits 24-25: work-groups per CU (0 = the configuration's default)
  if (KH * KW == 1 && stride == 1 && pad_h == 0 && pad_w == 0 && Hin == Hout && Win == Wout && !(variant & (1 << 19)) &&
      (variant & 0xfff) == 0) {
    const int rc = launch_wgrad_stream(a.x, a.dy, dw, dw_stride_n, dw_stride_c, n_valid, c_valid, a.zero, a.M, C, x_ld, N, dy_ld,
                                       (variant >> 18) & 1, (variant >> 20) & 1, (variant >> 21) & 7, (variant >> 24) & 3,
                                       (hipStream_t)stream);
    if (rc == 1) return 0;
    if (rc < 0) return rc;
  }
  // 3x3 / stride 1 / pad 1: all nine taps per work-group, input halo in LDS (wgrad_halo.hip); variant bit 12 forces it,
  // bit 13 forbids it (as does any of the per-tap kernel's own variant bits 0-10), bits 14-15: rounds of work-groups (0 = one per CU);
  // automatic when a work-group gets >= 4000 positions to reduce
  const bool halo_ok = KH == 3 && KW == 3 && stride == 1 && pad_h == 1 && pad_w == 1 && Hin == Hout && Win == Wout;
  if (halo_ok && !(variant & 8192) && (variant & 0x7ff) == 0) {
    const int rc = launch_wgrad_halo(a.x, a.dy, dw, dw_stride_n, dw_stride_tap, dw_stride_c, n_valid, c_valid, a.zero, B, Hin, Win,
                                     C, x_ld, N, dy_ld, ((variant >> 14) & 3) + 1, (variant & 4096) ? 1 : 0,
                                     (variant & 65536) ? 1 : ((variant & 131072) ? 2 : 0), (hipStream_t)stream);
    if (rc == 1) { g_last_conv_kernel = 2900; return 0; }
    if (rc < 0) return rc;
  }
  // 256 x 256 tiles (half the operand traffic per flop) win where nothing else hides the memory stream: 1x1 layers over the
  // stride-4 / stride-8 maps and the 7x7 "fully connected" fc1 (tests/native/selftest bench2w, profiles/r02_wgrad_variants.txt);
  // variant bit 8 forces them, bit 11 forbids them
  // Round 4, with the partial-tile reduction in place of the 256 KB atomic epilogue (bench2w 0 256 2048, profiles/r04_wgrad_partials.txt):
  // the 256-wide tiles also win on the stride-16 1x1 layers (res4 256 <-> 1024: 71 -> 67-69 us on two boxes, the stride-2 512 -> 1024
  // 124 -> 112) - wherever both channel counts fill 256-wide tiles; with a 128-channel side (128 <-> 512, 256 -> 128) the result
  // flipped between two boxes (+-10 %), so those stay on the 128-wide tiles; below ~50 k pixels (res5, the FC layers) they lose.
  const bool wide_1x1 = KH * KW == 1 && a.M >= 50000 && N % 256 == 0 && C % 256 == 0;
  const bool wide_auto = !(variant & 2048) &&
                         (wide_1x1 || ((N % 256 == 0) && (C % 256 == 0) && KH * KW > 1 && a.M <= 16384 && Hout == 1 && Wout == 1));
  const bool wide = ((variant & 256) || wide_auto) && (variant & 3) == 0;
  const int tw = wide ? 256 : 128;
  a.tiles_n = (N + tw - 1) / tw;
  a.tiles_c = (C + tw - 1) / tw;
  const int tiles = a.tiles_n * a.tiles_c * KH * KW;
  // Pixel splits: fill the chip once (4 resident work-groups per CU = 1024 slots) but keep >= 2048 pixels per
  // work-group so that the 64 KB fp32-atomic epilogue stays a small fraction; tiny layers fall back to >= 512 groups.
  int slots = (wide ? 256 : 1024) >> ((variant >> 4) & 3);  // resident work-groups; variant bits 4-5: tuning knob
  int splits = slots / tiles;
  if (splits < 1) splits = 1;
  // (1024 and 512 pixels per work-group measured 25 % slower on the stride-16 / 32 layers: the epilogue grows with the splits)
  const int by_pixels = a.M / 2048 > 1 ? a.M / 2048 : 1;
  if (splits > by_pixels) splits = by_pixels;
  const int min_groups = wide ? 256 : 512;
  if (tiles * splits < min_groups) {
    int s2 = (min_groups + tiles - 1) / tiles;
    const int cap = a.M / 512 > 1 ? a.M / 512 : 1;
    if (s2 > cap) s2 = cap;
    if (s2 > splits) splits = s2;
  }
  int ppw = (a.M + splits - 1) / splits;
  ppw = ((ppw + WP - 1) / WP) * WP;
  splits = (a.M + ppw - 1) / ppw;
  a.pix_per_wg = ppw;
  a.tiles = tiles;
  // measured: +29% on res3 3x3, +6% on the 200x336 3x3 layers, -3..-10% on small-M layers and plain GEMMs
  a.xcd_group = ((variant & 64) || (!(variant & 128) && KH * KW > 1 && a.M >= 200000)) ? 1 : 0;
  const dim3 grid = a.xcd_group ? dim3(tiles * splits) : dim3(tiles, splits);
  const dim3 block(256);
  hipStream_t s = (hipStream_t)stream;
  // partial tiles + one reduction pass instead of the fp32-atomic epilogue (variant bit 16 keeps the atomics); a single split
  // has nothing to reduce and little to gain
  a.part = nullptr;
  // Measured (tests/native/selftest bench2w 0 65536, round 4): the atomics of a long kernel hide behind the work-groups that are
  // still multiplying, so the pass only pays where the launch is short or has many tiles - 1x1 layers on the stride-8 ... 32 maps
  // (res4 78 -> 72 us, res5 67 -> 58, lat3 126 -> 113, res3 256->512 138 -> 114), mid-size 3x3 layers (+3...10 %); it loses on
  // the stride-4 1x1 layers with one or two tiles and 512 splits (64 MB of partials: 64->256 134 -> 156 us) and on the big 3x3
  // layers (-2...4 %).  variant bit 17 forces the pass wherever splits >= 2.
  const long long dw_elems = (long long)a.tiles_n * a.tiles_c * tw * tw;
  const bool part_auto = dw_elems >= 4LL * 128 * 128 && !(KH * KW > 1 && a.M >= 200000);
  if (!(variant & 65536) && splits >= 2 && (part_auto || (variant & 131072)))
    a.part = wgrad_scratch(s, (size_t)tiles * splits * tw * tw * sizeof(float));
  auto reduce = [&]() -> int {
    if (!a.part) return 0;
    const int slabs = tw / 8;
    int groups = (512 + tiles * slabs - 1) / (tiles * slabs);   // >= ~512 work-groups in the reduction pass
    if (groups > splits / 4) groups = splits / 4 > 1 ? splits / 4 : 1;
    const int per_group = (splits + groups - 1) / groups;
    groups = (splits + per_group - 1) / per_group;
    if (wide) hipLaunchKernelGGL(wgrad_reduce_kernel<256>, dim3(tiles, slabs, groups), dim3(256), 0, s, a.part, splits, tiles, per_group, a);
    else hipLaunchKernelGGL(wgrad_reduce_kernel<128>, dim3(tiles, slabs, groups), dim3(256), 0, s, a.part, splits, tiles, per_group, a);
    U2_CHECK_LAUNCH();
    return 0;
  };
  const bool glds = (variant & 1) == 0, tr = (variant & 2) == 0;
  g_last_conv_kernel = (wide ? 2256 : 2000 + (glds ? 2 : 0) + (tr ? 1 : 0)) + (a.xcd_group ? 100 : 0);
  if (wide) {
    static PerDeviceOnce attr_set;
    if (auto once_guard = attr_set.first()) {
      (void)hipFuncSetAttribute((const void*)conv_wgrad256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL(conv_wgrad256_kernel, grid, dim3(512), WG256_STAGES * WG256_STAGE, s, a);
    U2_CHECK_LAUNCH();
    return reduce();
  }
  const int ring = (variant >> 9) & 3;  // bits 9-10: LDS ring depth of the LDS-DMA form (0 = two buffers, 1 = 3, 2 = 4)
  if (glds && tr && ring == 1)      hipLaunchKernelGGL((conv_wgrad_kernel<true, true, 3>), grid, block, 0, s, a);
  else if (glds && tr && ring == 2) hipLaunchKernelGGL((conv_wgrad_kernel<true, true, 4>), grid, block, 0, s, a);
  else if (glds && tr)  hipLaunchKernelGGL((conv_wgrad_kernel<true, true>), grid, block, 0, s, a);
  else if (glds && !tr) hipLaunchKernelGGL((conv_wgrad_kernel<true, false>), grid, block, 0, s, a);
  else if (!glds && tr) hipLaunchKernelGGL((conv_wgrad_kernel<false, true>), grid, block, 0, s, a);
  else                  hipLaunchKernelGGL((conv_wgrad_kernel<false, false>), grid, block, 0, s, a);
  U2_CHECK_LAUNCH();
  return reduce();
}
