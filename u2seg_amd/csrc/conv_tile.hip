// Persistent tile kernels of the implicit-GEMM convolution on CDNA4 MFMA (gfx950), bf16 in / fp32 accumulate.
//
// Same operation and argument block as conv_igemm.hip (F.conv2d / F.linear forward and data gradient behind
// detectron2/layers/wrappers.py:127-134, roi_heads/box_head.py:94-97, roi_heads/mask_head.py:287-290), restructured
// around three facts measured on the round-1 kernels:
//   * a work-group that stops at the end of its tile leaves the matrix pipe idle for the tile's prologue (first operands
//     come from HBM) and epilogue; for the 1x1 layers (4-16 K steps) that is most of the tile's life.  Here a work-group
//     is persistent: it walks a strided list of tiles and its LDS-DMA ring never drains - the operands of the next
//     tile are already in flight while the last K steps of the current one are multiplied and while it is stored;
//   * the LDS-staged output transpose cost two barriers, 32-48 KB of LDS and made the ring alias the output tile.  Here
//     the weight rows of a wave's 64-channel slice are staged in a permuted order, so that MFMA leaves every lane with
//     two runs of 8 consecutive channels of one pixel: the accumulators go to HBM as 16-byte stores straight from
//     registers (4 lanes = one 64-byte segment), bias / ReLU / accumulate / BN column statistics applied on the way;
//   * with the whole LDS free for operands the ring is up to five half-K tiles deep (160 KB): global_load_lds runs
//     three to four half tiles ahead of the MFMAs under counted vmcnt, one raw barrier per half tile.
//
// Wave tile 64 channels x 128 pixels (4 x 8 v_mfma_f32_16x16x32_bf16 accumulators, 12 ds_read_b128 per 32 MFMA);
// work-group = WCH x WPX waves: 4x2 (256 ch x 256 px, one group per CU), 2x2 and 4x1 (two groups per CU).
#include <stdlib.h>
#include <mutex>
#include <type_traits>
#include <vector>
#include <algorithm>
#include <stdio.h>

#include "common.h"
#include "conv_args.h"

namespace u2conv {
namespace {

#ifdef U2_TILE_TRACE
// Debug build only (tools/exp/tile_trace.sh): wave 0 of every work-group stamps s_memtime at the phase boundaries of its life
// into g_tile_trace[blockIdx.x][64]; slots: 0 entry, 1 first stage published, 2 + gh (gh < 44) the barrier of half K tile gh,
// 48 / 49 around sk_publish, 50 K loop done, 51 peer's flag seen, 52 partial added, 53 epilogue done, 54 statistics flushed.
__device__ unsigned long long* g_tile_trace_dev = nullptr;
#define U2_STAMP(SLOT)                                                                                                        \
  do {                                                                                                                         \
    if (trace_on) {                                                                                                            \
      unsigned long long t_;                                                                                                   \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                              \
      if (lane == 0) asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" ::"v"(trace_row + (SLOT)), "v"(t_) : "memory"); \
    }                                                                                                                          \
  } while (0)
// Sub-step timeline (-DU2_TILE_TRACE_POINT=k, k = 1 .. 6): waves 0 and 4 of a work-group - the two waves of one SIMD - stamp the
// barrier of every half K tile (kind 0) and ONE further point of the step (kind 1; one point per build, so that a build carries one extra
// blocking stamp per step): 1 = behind the pixel staging and the first eight MFMAs of phase B, 2 = end of the step, 3 = behind the
// first four MFMAs of phase A, 4 = behind phase A's fragment reads and weight staging, 5 = behind its twelve MFMAs, 6 = behind the
// counted vmcnt wait (in front of lgkmcnt(0) + barrier).  g_tile_sub_dev[blockIdx.x][wave 0 | 4][kind][48 steps].
#ifndef U2_TILE_TRACE_POINT
#define U2_TILE_TRACE_POINT 0
#endif
__device__ unsigned long long* g_tile_sub_dev = nullptr;
#define U2_SUB(KIND)                                                                                                           \
  do {                                                                                                                         \
    if (sub_on && gh < 48) {                                                                                                   \
      unsigned long long t_;                                                                                                   \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                              \
      if (lane == 0) asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" ::"v"(sub_row + (KIND) * 48 + gh), "v"(t_) : "memory"); \
    }                                                                                                                          \
  } while (0)
#define U2_SUBK(K) do { if constexpr (U2_TILE_TRACE_POINT == (K)) U2_SUB(1); } while (0)
#else
#define U2_STAMP(SLOT) do { } while (0)
#define U2_SUB(KIND) do { } while (0)
#define U2_SUBK(K) do { } while (0)
#endif

// measurement switch: -DU2_TILE_NOPRIO compiles the s_setprio brackets around the MFMA groups out (tools/exp/tile_noprio_ab.sh)
#ifdef U2_TILE_NOPRIO
#define U2_TILE_SETPRIO(P) do { } while (0)
#else
#define U2_TILE_SETPRIO(P) __builtin_amdgcn_s_setprio(P)
#endif

#ifdef U2_TILE_DMA_BY_HALF
template <int N> __device__ __forceinline__ void wait_vm() {   // (the issuing half has twice the loads per stage in flight)
  if ((threadIdx.x >> 6) < (blockDim.x >> 7)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * N > 63 ? 63 : 2 * N) : "memory");
}
#else
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#endif

template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// Transposing reduction over the 16 lanes of a DPP row (lanes sharing lane >> 4): every lane enters with 16 partial sums
// v[0..15]; lane fr leaves with the row total of v[fr].  Four exchange steps, each halving the values a lane carries
// (the lane keeps the half selected by one bit of fr and adds its partner's copy of that half): 15 adds instead of the 64
// of an all-reduce, and the totals end up one per lane, which is what a full-wave atomic wants.
__device__ __forceinline__ float row16_transpose_sum(float (&v)[16], int fr) {
  {
    const bool up = fr & 8;  // partner fr ^ 8 (row_ror:8)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float keep = up ? v[k + 8] : v[k], send = up ? v[k] : v[k + 8];
      v[k] = keep + dpp_mov<0x128>(send);
    }
  }
  {
    const bool up = fr & 4;  // partner fr ^ 7 (row_half_mirror): same bit 3, opposite bit 2
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float keep = up ? v[k + 4] : v[k], send = up ? v[k] : v[k + 4];
      v[k] = keep + dpp_mov<0x141>(send);
    }
  }
  {
    const bool up = fr & 2;  // partner fr ^ 2 (quad_perm [2,3,0,1])
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float keep = up ? v[k + 2] : v[k], send = up ? v[k] : v[k + 2];
      v[k] = keep + dpp_mov<0x4E>(send);
    }
  }
  const bool up = fr & 1;    // partner fr ^ 1 (quad_perm [1,0,3,2])
  const float keep = up ? v[1] : v[0], send = up ? v[0] : v[1];
  return keep + dpp_mov<0xB1>(send);
}

// m / d and m % d for 0 <= m < 2^24 through one float multiply (inv = 1.0f / d) and a +-1 correction
__device__ __forceinline__ void divmod_f(int m, int d, float inv, int& q, int& r) {
  q = (int)((float)m * inv);
  r = m - q * d;
  if (r < 0) { --q; r += d; }
  if (r >= d) { ++q; r -= d; }
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
// two floats -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return *reinterpret_cast<const uint32_t*>(&r);
}

// ACC: out = act(bf16(conv + bias) + out) - the epilogue reads what it overwrites (a residual block's tail at inference)
// SK ("stream-K"): instead of whole tiles, the work-groups of an XCD take EQUAL shares of the XCD's (tile, half K tile)
// sequence - for layers whose tile count fills the last round of tiles badly (263 or 525 tiles on 256 / 512 slots: half of
// the chip idle for the second round).  A share that begins inside a tile (k0 > 0) holds a non-leading part of it: its
// accumulators go, raw fp32, into the work-group's slot of a.sk_ws and its flag is raised - at the START of that work-group's
// life.  The share that holds the tile's first half K tile is its owner: at the END of its life it waits for the flags of
// the (consecutive) work-groups holding the rest, adds their slots and runs the ordinary epilogue.  Every work-group of the
// launch is resident (grid <= CUs x work-groups per CU), and nobody waits before having published, so the hand-over cannot
// deadlock; slot stores and loads are both `sc0 sc1` (coherent for any placement of the two work-groups, MI355X_MICROARCH.md
// "Workgroup dispatch ..."), the flag is a relaxed agent-scope atomic behind s_waitcnt vmcnt(0) + barrier.
// KT = 2 (round 5): a stage of the ring is a WHOLE K tile - rows of 64 channels, 128 bytes, so that an LDS-DMA instruction moves
// 8 rows x 128 contiguous bytes instead of 16 x 64 (tests/native/dma_bench: 64-byte pieces at a 1-4 KB pitch stream at 3.5 TB/s,
// 128-byte pieces at 6.3).  For the HBM-streaming 1x1 layers with long rows (K >= 512).  Two half-tile MFMA sequences per stage;
// staging, the counted wait and the barrier happen once per stage (weights at the first half, pixels behind the barrier of the
// second).  In the text below "half K tile" then reads "K tile" wherever it means the ring's unit.
#ifndef U2_SKB
#define U2_SKB 16
#endif
#ifndef U2_TILE_READS_FIRST
#define U2_TILE_READS_FIRST 0   // measured: not faster (profiles/r06_tile_loop_experiments.txt)
#endif
#ifndef U2_TILE_LATE_PIXELS
#define U2_TILE_LATE_PIXELS 0
#endif
#ifndef U2_TILE_W_FIRST
#define U2_TILE_W_FIRST 0
#endif
#ifndef U2_TILE_PINGPONG
#define U2_TILE_PINGPONG 0
#endif
#ifndef U2_TILE_ROLES
#define U2_TILE_ROLES 0
#endif
#ifndef U2_TILE_PEEL
#define U2_TILE_PEEL 0   // measured: not faster (profiles/r06_tile_loop_experiments.txt)
#endif
template <int WCH, int WPX, int RING, bool ACC = false, bool SK = false, int KT = 1>
__global__ __launch_bounds__(WCH * WPX * 64, 2) void conv_tile_kernel(const ConvArgs a) {
  constexpr int NW = WCH * WPX, NT = NW * 64;
  constexpr int TN = WCH * 64, TM = WPX * 128;
  constexpr int KU = 32 * KT;        // channels per stage
  constexpr int ROWB = 64 * KT;      // bytes per staged row
  constexpr int CPRW = ROWB / 16;    // 16-byte chunks per row
  constexpr int RPI = 1024 / ROWB;   // rows one LDS-DMA instruction moves
  // U2_TILE_ROLES (round 6, profiles/r06_tile_loop_experiments.txt item 8): in the eight-wave work-groups the WEIGHT operand is staged by
  // waves 0-3 alone and the PIXEL operand by waves 4-7 alone (twice the rows per issuing thread), so that on every SIMD one wave
  // issues a phase's four LDS-DMA instructions back to back while its partner goes on multiplying - an LDS-DMA instruction costs
  // the issuing wave ~100+ cycles when both waves of a SIMD issue two each in the same phase.  Loads per wave and stage stay 4.
  constexpr bool ROLES = U2_TILE_ROLES && NW == 8 && KT == 1;
  constexpr int SNW = ROLES ? NW / 2 : NW;          // waves that stage one operand
  constexpr int LP = TM * CPRW / (SNW * 64);        // 16-byte chunks an ISSUING thread moves per stage, pixel operand
  constexpr int LW = TN * CPRW / (SNW * 64);        //                                                   weight operand
  constexpr int LPT = ROLES ? LP : LP + LW;         // LDS-DMA instructions per wave and stage
  static_assert(!ROLES || LP == LW, "the role split needs equally many pixel and weight rows per stage");
  constexpr int PBYTES = TM * ROWB, WBYTES = TN * ROWB, BUF = PBYTES + WBYTES;  // one stage: rows of KU bf16
  constexpr int AHEAD = RING - 1;
  static_assert(LP >= 1 && LW >= 1 && RING >= (KT == 2 ? 2 : 3) && RING <= 5 && (KT == 1 || KT == 2), "unsupported configuration");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sw_ = ROLES ? (w & (SNW - 1)) : w;            // this wave's index among the waves that stage an operand
  const bool p_issuer = !ROLES || w >= SNW, w_issuer = !ROLES || w < SNW;
#ifdef U2_TILE_TRACE
  const bool trace_on = w == 0 && g_tile_trace_dev != nullptr;
  unsigned long long* trace_row = g_tile_trace_dev + (size_t)blockIdx.x * 64;
  const bool sub_on = U2_TILE_TRACE_POINT != 0 && (w == 0 || w == 4) && g_tile_sub_dev != nullptr;
  unsigned long long* sub_row = g_tile_sub_dev + ((size_t)blockIdx.x * 2 + (w >> 2)) * 96;
  U2_STAMP(0);
#endif

  // ---- the tiles of this work-group: XCD x owns a contiguous range of the (tile_m, tile_n) list (tile_n fastest), its
  // work-groups walk that range with stride gridDim/8, so at any time an XCD's L2 serves one window of neighbouring tiles
  const int T = a.tiles_m * a.tiles_n;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, stride = gridDim.x >> 3;
  const int q8 = T >> 3, r8 = T & 7;
  const int xbase = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int xcnt = q8 + (xcd < r8 ? 1 : 0);
  const int kh_per_tap = a.C / KU;           // stages (KU channels) per filter tap
  const int nkh = a.ntaps * kh_per_tap;      // >= RING (host)
  int my_tiles, first_tile, H;
  int sk_k0 = 0, sk_k1 = nkh;                // SK: first half K tile of the first tile / end of the last tile of this share
  long long sk_s0 = 0;
  const long long sk_S = (long long)xcnt * nkh;   // SK: half K tiles of this XCD's tile range
  if constexpr (SK) {
    sk_s0 = sk_S * idx / stride;
    const long long s1 = sk_S * (idx + 1) / stride;
    if (s1 <= sk_s0) return;
    first_tile = xbase + (int)(sk_s0 / nkh);
    sk_k0 = (int)(sk_s0 % nkh);
    const int last = (int)((s1 - 1) / nkh);
    my_tiles = last - (int)(sk_s0 / nkh) + 1;
    sk_k1 = (int)(s1 - (long long)last * nkh);
    H = (int)(s1 - sk_s0);
  } else {
    if (idx >= xcnt) return;
    my_tiles = (xcnt - idx + stride - 1) / stride;
    first_tile = xbase + idx;
    H = my_tiles * nkh;                      // half K tiles this work-group multiplies, across all its tiles
  }
  const int tstep = SK ? 1 : stride;         // distance of consecutive tiles of this work-group in the tile list

  // ---- staging cursors.  The pixel cursor runs AHEAD + 1 half tiles in front of the MFMAs, the weight cursor AHEAD; both
  // cross tile boundaries on their own.  Per row a thread keeps the address of the tap-(0,0) source pixel and its (y, x):
  // a tap only adds a wave-uniform offset and a bounds test (rows that fall into the padding read the zero page).
  const int row_in = lane / CPRW;
  const int cc = (lane % CPRW) ^ swz<KU>(row_in);
  const int hw = a.Hout * a.Wout;
  const float inv_hw = 1.0f / (float)hw, inv_w = 1.0f / (float)a.Wout;
  const bool linear = a.ntaps == 1 && a.mul == 1 && a.Hin == a.Hout && a.Win == a.Wout &&
                      (a.remap_out ? ((a.tap_pk[0] & 0xffff) == 0) : (a.pad_h == 0 && a.pad_w == 0));

  unsigned p_center[LP];  // byte offset of the tap-(0,0) source chunk from a.in (host: the input is < 4 GB)
  int p_yx[LP];
  const bf16_t* p_src[LP];
  unsigned p_okmask = 0;  // bit i: row i of this thread reads real data in the current tap (advance by 32 channels)
  int pt = -1, p_tap = a.ntaps, p_kc = 0, p_kh = 0, p_kw = 0;
  auto setup_pixels = [&](int tile) {
    const int m0 = (tile / a.tiles_n) * TM;
#pragma unroll
    for (int i = 0; i < LP; ++i) {
      const int m = m0 + (i * SNW + sw_) * RPI + row_in;
      if (m < a.M) {
        if (linear) {
          p_center[i] = (unsigned)(((size_t)m * a.in_ld + cc * 8) * 2);
          p_yx[i] = 0;
        } else {
          int img, rem, oy, ox;
          divmod_f(m, hw, inv_hw, img, rem);
          divmod_f(rem, a.Wout, inv_w, oy, ox);
          const int y = oy * a.mul, x = ox * a.mul;
          p_center[i] = (unsigned)((((size_t)(img * a.Hin + y) * a.Win + x) * a.in_ld + cc * 8) * 2);
          p_yx[i] = (y & 0xffff) | (x << 16);
        }
      } else {
        p_center[i] = 0;
        p_yx[i] = 0x8000;  // y = -32768: every tap is out of range
      }
    }
  };
  // enters filter tap p_tap (of the tile setup_pixels() prepared) at its 32-channel slab `skip` (0 except for a stream-K
  // share that begins inside a tile)
  auto pixel_tap = [&](int skip) {
      int dy, dx;
      if (a.remap_out) {
        const int pk = a.tap_pk[__builtin_amdgcn_readfirstlane(p_tap)];
        dy = (int)(signed char)(pk & 0xff); dx = (int)(signed char)((pk >> 8) & 0xff);
      } else {
        dy = p_kh - a.pad_h; dx = p_kw - a.pad_w;
        if (++p_kw == a.KW) { p_kw = 0; ++p_kh; }
      }
      const unsigned char* tap_base = reinterpret_cast<const unsigned char*>(a.in) + ((long long)dy * a.Win + dx) * a.in_ld * 2;
      p_okmask = 0;
#pragma unroll
      for (int i = 0; i < LP; ++i) {
        const int sy = (int)(short)(p_yx[i] & 0xffff) + dy, sx = (p_yx[i] >> 16) + dx;
        const bool ok = (unsigned)sy < (unsigned)a.Hin && (unsigned)sx < (unsigned)a.Win;
        p_src[i] = ok ? reinterpret_cast<const bf16_t*>(tap_base + p_center[i]) + skip * KU : a.zero;
        p_okmask |= ok ? (1u << i) : 0u;
      }
      ++p_tap;
      p_kc = kh_per_tap - skip;
  };
  auto stage_pixels = [&](int buf) {
    if (p_kc == 0) {
      if (p_tap == a.ntaps) {
        p_tap = 0; p_kh = 0; p_kw = 0;
        ++pt;
        setup_pixels(first_tile + pt * tstep);
      }
      pixel_tap(0);
    }
    --p_kc;
    unsigned char* base = smem + buf * BUF;
    // measurement switches (timing only, results are wrong): abl bit 4 = the pixel operand is staged by waves 0 .. NW/2-1 only and
    // the weights by the others (half of the LDS-DMA instructions, the two waves of a SIMD issue theirs in different phases);
    // bit 5 = no wave stages weights (half of the instructions, every wave in the same phase); both = no staging at all
    // (compiled in with -DU2_TILE_DMA_ABLATION only - tools/exp/dma_phase_ablation.sh: the test puts a branch around every staging
    //  instruction of the production loop otherwise)
#ifdef U2_TILE_DMA_ABLATION
    const bool skip_p = ((a.abl & 48) == 16 && w >= NW / 2) || (a.abl & 48) == 48;
#else
    constexpr bool skip_p = false;
#endif
#ifdef U2_TILE_DMA_BY_HALF
    // TIMING ONLY (wrong results): waves 0 .. NW/2-1 issue the staging instructions of the whole work-group - their own and, with
    // their own source rows, their SIMD partner's - and the other waves issue none: does an LDS-DMA instruction cost less when one
    // wave of a SIMD issues them back to back than when both waves issue two each?
    if (w < NW / 2) {
#pragma unroll
      for (int i = 0; i < LP; ++i) {
        glds16(p_src[i], base + (i * NW + w) * 1024);
        glds16(p_src[i], base + (i * NW + w + NW / 2) * 1024);   // (ablation build: not combined with U2_TILE_ROLES)
      }
    }
#pragma unroll
    for (int i = 0; i < LP; ++i) p_src[i] += (p_okmask >> i & 1u) * KU;
#else
#pragma unroll
    for (int i = 0; i < LP; ++i) {
      if (!skip_p && p_issuer) glds16(p_src[i], base + (i * SNW + sw_) * 1024);
      p_src[i] += (p_okmask >> i & 1u) * KU;
    }
#endif
  };

  // Weight rows: LDS row rho = blk * 16 + q of a wave's 64-channel slice holds channel
  //   (blk >> 1) * 32 + (q >> 2) * 8 + (blk & 1) * 4 + (q & 3),
  // so that the MFMA output rows a lane owns (q = fg * 4 + r in each of the 4 blocks) are channels fg*8 .. fg*8+7 and
  // 32 + fg*8 .. 32 + fg*8+7 of the slice: two 16-byte runs per pixel.
  unsigned w_base[LW];  // byte offset of the row's first chunk from a.wt; 0xffffffff = row beyond N (reads the zero page)
  const bf16_t* w_src[LW];
  int wt_i = -1, w_tap = a.ntaps, w_kc = 0;
  auto setup_weights = [&](int tile) {
    const int n0 = (tile % a.tiles_n) * TN;
#pragma unroll
    for (int i = 0; i < LW; ++i) {
      const int R = (i * SNW + sw_) * RPI + row_in;
      const int blk = (R >> 4) & 3, q = R & 15;
      const int n = n0 + (R & ~63) + (blk >> 1) * 32 + (q >> 2) * 8 + (blk & 1) * 4 + (q & 3);
      w_base[i] = n < a.N ? (unsigned)(((size_t)n * ((size_t)a.wt_taps * a.C) + cc * 8) * 2) : 0xffffffffu;
    }
  };
  auto weight_tap = [&](int skip) {
      const int wtap = a.remap_out ? (a.tap_pk[__builtin_amdgcn_readfirstlane(w_tap)] >> 16) : w_tap;
      const unsigned char* tap_base = reinterpret_cast<const unsigned char*>(a.wt) + ((size_t)wtap * a.C + (size_t)skip * KU) * 2;
#pragma unroll
      for (int i = 0; i < LW; ++i)
        w_src[i] = w_base[i] != 0xffffffffu ? reinterpret_cast<const bf16_t*>(tap_base + w_base[i]) : a.zero;
      ++w_tap;
      w_kc = kh_per_tap - skip;
  };
  auto stage_weights = [&](int buf) {
    if (w_kc == 0) {
      if (w_tap == a.ntaps) {
        w_tap = 0;
        ++wt_i;
        setup_weights(first_tile + wt_i * tstep);
      }
      weight_tap(0);
    }
    --w_kc;
    unsigned char* base = smem + buf * BUF + PBYTES;
#ifdef U2_TILE_DMA_ABLATION
    const bool skip_w = ((a.abl & 48) == 16 && w < NW / 2) || (a.abl & 32);
#else
    constexpr bool skip_w = false;
#endif
#ifdef U2_TILE_DMA_BY_HALF
    if (w < NW / 2) {
#pragma unroll
      for (int i = 0; i < LW; ++i) {
        glds16(w_src[i], base + (i * NW + w) * 1024);
        glds16(w_src[i], base + (i * NW + w + NW / 2) * 1024);
      }
    }
#pragma unroll
    for (int i = 0; i < LW; ++i) w_src[i] += w_base[i] != 0xffffffffu ? KU : 0;
#else
#pragma unroll
    for (int i = 0; i < LW; ++i) {
      if (!skip_w && w_issuer) glds16(w_src[i], base + (i * SNW + sw_) * 1024);
      w_src[i] += w_base[i] != 0xffffffffu ? KU : 0;
    }
#endif
  };

  const int wr = w / WPX;  // 64-channel slice of the tile
  const int wc = w % WPX;  // 128-pixel slice of the tile
  const int fr = lane & 15;
  const int fg = lane >> 4;
  // the swizzle has a period of 16 rows: fragment t of a wave is fragment 0 + t * 1024 bytes (an immediate offset)
  // (KT = 2: the second half's chunks 4-7 of a row sit at the first half's position ^ 64 bytes)
  const int wfrag0 = PBYTES + (wr * 64 + fr) * ROWB + ((fg ^ swz<KU>(fr)) << 4);
  const int pfrag0 = (wc * 128 + fr) * ROWB + ((fg ^ swz<KU>(fr)) << 4);
  auto ldw = [&](int hb, int t, int half = 0) { return *reinterpret_cast<const s16x8*>(smem + hb * BUF + (wfrag0 ^ (half << 6)) + t * 16 * ROWB); };
  auto ldp = [&](int hb, int t, int half = 0) { return *reinterpret_cast<const s16x8*>(smem + hb * BUF + (pfrag0 ^ (half << 6)) + t * 16 * ROWB); };
#if U2_TILE_READS_FIRST || U2_TILE_PINGPONG
  // Fragment reads as inline asm with counted waits (KT = 1 only): the compiler answers a plain LDS load it must wait for with
  // s_waitcnt lgkmcnt(0) whenever LDS-DMA is in flight (it will not count across it), which forbids having a SECOND batch of reads in
  // flight while the first is consumed.  LDS returns in order, so `lgkmcnt(6)` with twelve reads outstanding releases the older six.
  const unsigned lds_base = (unsigned)(size_t)U2_LDS_PTR(smem);
#define U2_T_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#define U2_T_WAIT6(CNT, R0, R1, R2, R3, R4, R5) \
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(R0), "+v"(R1), "+v"(R2), "+v"(R3), "+v"(R4), "+v"(R5) : "n"(CNT) : "memory")
#endif

  f32x4 acc[4][8];
  s16x8 wfA[2], wfB[2], pf[8];

  // BN column statistics: a lane accumulates the sums of ONE channel (st_n) over all tiles the work-group walks with the
  // same tile_n and sends them with two full-wave atomics when the channel changes or the work-group is done.
  float st_s = 0.f, st_ss = 0.f;
  int st_n = -1;
  auto stats_flush = [&]() {
    if (st_n >= 0 && st_n < a.N) {
      asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(a.stats + st_n), "v"(st_s) : "memory");
      asm volatile("global_atomic_add_f32 %0, %1, off\n\ts_nop 1" ::"v"(a.stats + a.N + st_n), "v"(st_ss) : "memory");
    }
    st_s = 0.f; st_ss = 0.f;
  };

  // ---- epilogue: registers -> HBM.  acc[i][j][r] = channel (i >> 1) * 32 + fg * 8 + (i & 1) * 4 + r of the wave's slice,
  // pixel j * 16 + fr of the wave's 128 pixels.  Every memory operation in here is inline asm on purpose: a global load,
  // store or atomic the compiler can see inside the tile loop makes its wait-count pass put `s_waitcnt vmcnt(0)` in front
  // of the first fragment read of every half tile (it cannot tell the LDS-DMA in flight from these), which turns the ring
  // into a synchronous load.  Stores carry their own `s_nop 1` (the data registers may be reused right after the
  // statement), the bias loads wait inside their statement.
  auto epilogue = [&](int tile) {
    const int tile_m = tile / a.tiles_n, tile_n = tile - tile_m * a.tiles_n;
    const int m0 = tile_m * TM + wc * 128;
    const int nb = tile_n * TN + wr * 64 + fg * 8;
    const bool okA = nb < a.N, okB = nb + 32 < a.N;  // N % 8 == 0: an 8-channel run is valid or absent as a whole
    f32x4 bia[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (a.bias) {  // drains the LDS-DMA ring once per tile (the layers with a bias have long reductions)
      if (okA)
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(bia[0]), "=&v"(bia[1]) : "v"(a.bias + nb) : "memory");
      if (okB)
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(bia[2]), "=&v"(bia[3]) : "v"(a.bias + nb + 32) : "memory");
    }
    float s[16], ss[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { s[e] = 0.f; ss[e] = 0.f; }
    const bool nt_out = (size_t)a.M * a.out_ld * 2 > ((size_t)160 << 20) && (a.abl & 8);
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    u32x4_t olda[4], oldb[4];  // ACC: the values under rows j & 3 of the current batch of four rows
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (ACC) {
        if ((j & 3) == 0) {
          // four rows' worth of loads, then ONE wait for everything in flight (a counted wait would have to know which of
          // the masked loads / stores below were issued at all): two exposed round trips per tile instead of eight
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int mm = m0 + (j + jj) * 16 + fr;
            olda[jj] = u32x4_t{0u, 0u, 0u, 0u};
            oldb[jj] = u32x4_t{0u, 0u, 0u, 0u};
            if (mm < a.M) {
              const bf16_t* src = a.out + (size_t)mm * a.out_ld + nb;
              if (okA) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(olda[jj]) : "v"(src) : "memory");
              if (okB) asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=&v"(oldb[jj]) : "v"(src) : "memory");
            }
          }
          asm volatile("s_waitcnt vmcnt(0)"
                       : "+v"(olda[0]), "+v"(olda[1]), "+v"(olda[2]), "+v"(olda[3]), "+v"(oldb[0]), "+v"(oldb[1]), "+v"(oldb[2]),
                         "+v"(oldb[3])::"memory");
        }
      }
      const int m = m0 + j * 16 + fr;
      if (m < a.M) {
        size_t orow = (size_t)m;
        if (a.remap_out) {
          int img, rem, qy, qx;
          divmod_f(m, hw, inv_hw, img, rem);
          divmod_f(rem, a.Wout, inv_w, qy, qx);
          orow = ((size_t)img * a.Hfull + qy * a.out_sy + a.out_y0) * a.Wfull + qx * a.out_sx + a.out_x0;
        }
        bf16_t* dst = a.out + orow * a.out_ld + nb;
        uint32_t pk[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // block i holds channels e0 .. e0 + 3 of this lane's 16, e0 = (i >> 1) * 8 + (i & 1) * 4
          float v0 = acc[i][j][0] + bia[i][0], v1 = acc[i][j][1] + bia[i][1];
          float v2 = acc[i][j][2] + bia[i][2], v3 = acc[i][j][3] + bia[i][3];
          if constexpr (ACC) {
            // the conv result is rounded to bf16 first, then the stored value is added (conv_igemm_kernel's accumulate rule)
            const uint32_t c01 = pack_bf16(v0, v1), c23 = pack_bf16(v2, v3);
            const u32x4_t o = (i < 2) ? olda[j & 3] : oldb[j & 3];
            const uint32_t o01 = o[(i & 1) * 2], o23 = o[(i & 1) * 2 + 1];
            v0 = __uint_as_float(c01 << 16) + __uint_as_float(o01 << 16);
            v1 = __uint_as_float(c01 & 0xffff0000u) + __uint_as_float(o01 & 0xffff0000u);
            v2 = __uint_as_float(c23 << 16) + __uint_as_float(o23 << 16);
            v3 = __uint_as_float(c23 & 0xffff0000u) + __uint_as_float(o23 & 0xffff0000u);
          }
          if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
          pk[2 * i] = pack_bf16(v0, v1);
          pk[2 * i + 1] = pack_bf16(v2, v3);
        }
        if (!ACC && a.stats) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float lo = __uint_as_float(pk[e] << 16), hi = __uint_as_float(pk[e] & 0xffff0000u);
            s[2 * e] += lo; ss[2 * e] += lo * lo;
            s[2 * e + 1] += hi; ss[2 * e + 1] += hi * hi;
          }
        }
        const u32x4_t va = {pk[0], pk[1], pk[2], pk[3]}, vb = {pk[4], pk[5], pk[6], pk[7]};
        if (nt_out) {  // outputs far beyond the MALL size: do not let them evict what the next layer can still reuse
          if (okA) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(va) : "memory");
          if (okB) asm volatile("global_store_dwordx4 %0, %1, off offset:64 nt\n\ts_nop 1" ::"v"(dst), "v"(vb) : "memory");
        } else {
          if (okA) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(va) : "memory");
          if (okB) asm volatile("global_store_dwordx4 %0, %1, off offset:64\n\ts_nop 1" ::"v"(dst), "v"(vb) : "memory");
        }
      }
    }
    if (!ACC && a.stats) {
      // after the transposing reduction lane (fg, fr) holds the tile's column sums of the channel below
      const int n_here = tile_n * TN + wr * 64 + fg * 8 + (fr >> 3) * 32 + (fr & 7);
      if (n_here != st_n) { stats_flush(); st_n = n_here; }
      st_s += row16_transpose_sum(s, fr);
      st_ss += row16_transpose_sum(ss, fr);
    }
  };

#define U2_T_MFMA(I, WF, J) acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF, pf[J], acc[I][J], 0, 0, 0)

  if constexpr (SK) {  // both cursors enter the first tile at half K tile sk_k0
    const int tap0 = sk_k0 / kh_per_tap, skip0 = sk_k0 - tap0 * kh_per_tap;
    pt = 0; wt_i = 0;
    setup_pixels(first_tile);
    setup_weights(first_tile);
    p_tap = tap0; w_tap = tap0;
    p_kh = tap0 / a.KW; p_kw = tap0 - p_kh * a.KW;
    pixel_tap(skip0);
    weight_tap(skip0);
  }
  // ---- prologue: P0 W0 ... P(AHEAD-1) W(AHEAD-1) P(AHEAD) in flight, publish half tile 0 ----
#pragma unroll
  for (int i = 0; i < AHEAD; ++i) { stage_pixels(i); stage_weights(i); }
  stage_pixels(AHEAD);
  wait_vm<ROLES ? LPT * (AHEAD - 1) : LPT * (AHEAD - 1) + LP>();   // (ROLES: a weight wave has stages 0 .. AHEAD - 1 in flight, a pixel wave one more)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  U2_STAMP(1);
#if U2_TILE_PINGPONG
  if constexpr (KT == 1 && !ACC) {
    // ping-pong form: both wave groups enter their first step "cold" (they read its fragments themselves)
  } else
#endif
#if U2_TILE_READS_FIRST
  if constexpr (KT == 1) {
    // (asm like the reads of the loop: a compiler-visible LDS load pending at the loop header would make the wait-count pass put
    //  s_waitcnt lgkmcnt(0) in front of the loop's first counted wait in EVERY iteration)
    const unsigned pa = lds_base + (unsigned)pfrag0, wa = lds_base + (unsigned)wfrag0;
    U2_T_RD(wfA[0], wa, 0); U2_T_RD(wfA[1], wa, 16 * ROWB);
    U2_T_RD(pf[0], pa, 0); U2_T_RD(pf[1], pa, 16 * ROWB); U2_T_RD(pf[2], pa, 2 * 16 * ROWB); U2_T_RD(pf[3], pa, 3 * 16 * ROWB);
  } else
#endif
  {
  wfA[0] = ldw(0, 0); wfA[1] = ldw(0, 1);
#pragma unroll
  for (int j = 0; j < 4; ++j) pf[j] = ldp(0, j);
  }

  // One half K tile gh (buffer hb = gh % RING), two phases:
  //   A: 4 MFMA | read pixel fragments 4-7 and the second weight pair of gh, stage weights(gh + AHEAD) | 12 MFMA
  //   B: vmcnt (half tile gh + 1 landed), barrier [publishes gh + 1, frees buffer hb] | stage pixels(gh + AHEAD + 1) into hb |
  //      8 MFMA | read the first weight pair and pixel fragments 0-3 of gh + 1 | 8 MFMA
  // so no MFMA waits on an LDS read issued less than ~8 MFMAs earlier, and the LDS-DMA of a half tile has AHEAD - 1 full
  // half tiles of MFMAs to land.  The sequence runs straight through tile boundaries; only the accumulators are stored
  // and cleared there.
  // ---- SK hand-over of a partial tile (see the kernel's head comment).  Slot layout: 16 bytes per thread and accumulator
  // block, [i * 8 + j][thread] - every wave instruction moves 1 KB contiguous.
  typedef __attribute__((ext_vector_type(4))) unsigned int sk_u32x4;
  auto sk_publish = [&]() {
    // one running pointer, opaque to the optimiser: 32 precomputed 64-bit addresses would not fit beside the accumulators
    unsigned char* slot = reinterpret_cast<unsigned char*>(a.sk_ws) + (size_t)blockIdx.x * (TM * TN * 4) + (size_t)tid * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(slot), "v"(acc[i][j]) : "memory");
        slot += NT * 16;
        asm volatile("" : "+v"(slot));
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's part of the slot has left (no LDS hazard: the ring is only read / DMA-written)
    // (inline asm like every other memory operation inside the tile loop: a compiler-visible global access here would make
    //  the wait-count pass drain the LDS-DMA queue in front of every half tile's first fragment read)
    if (tid == 0) {
      const int one = 1;
      asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(a.sk_flags + blockIdx.x), "v"(one) : "memory");
    }
  };
  constexpr int SKB = U2_SKB;   // loads in flight per thread in sk_combine: 8, 12 (no), 16
  auto sk_combine = [&](int tile) {
    // the work-groups idx + 1, idx + 2, ... of this XCD hold the rest of the tile: all whose share begins before the tile ends
    const long long tile_end = (long long)(tile - xbase + 1) * nkh;
    for (int j = idx + 1; j < stride && sk_S * j / stride < tile_end; ++j) {
      const int peer = j * 8 + xcd;  // blockIdx.x of that work-group
      if (tid == 0) {
        int f;
        do {
          asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(f) : "v"(a.sk_flags + peer) : "memory");
          if (f == 0) __builtin_amdgcn_s_sleep(8);
        } while (f == 0);
        const int zero = 0;  // consumed: clean for the next launch
        asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(a.sk_flags + peer), "v"(zero) : "memory");
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      U2_STAMP(51);
      const unsigned char* slot = reinterpret_cast<const unsigned char*>(a.sk_ws) + (size_t)peer * (TM * TN * 4) + (size_t)tid * 16;
      {
        // Round 6: the slot comes from the other XCD's side of the fabric (sc0 sc1: ~2-3 us per dependent round trip under load), and
        // eight round trips of four loads were most of the ~33 us every stream-K launch costs beyond its K steps
        // (profiles/r06_1x1_k512.txt).  The K loop is over here - fragment and staging registers are dead - so sixteen loads
        // (64 registers, 128 KB per CU) are in flight at once: two round trips per peer, counted waits release them in fours (SKB loads per round trip).
#pragma unroll
        for (int q = 0; q < 32 / SKB; ++q) {
          f32x4 t[SKB];
#pragma unroll
          for (int jj = 0; jj < SKB; ++jj) {
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(t[jj]) : "v"(slot) : "memory");
            slot += NT * 16;
            asm volatile("" : "+v"(slot));
          }
#pragma unroll
          for (int g = 0; g < SKB / 4; ++g) {
            if (g == SKB / 4 - 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[g * 4]), "+v"(t[g * 4 + 1]), "+v"(t[g * 4 + 2]), "+v"(t[g * 4 + 3])::"memory");
            else if (g == SKB / 4 - 2) asm volatile("s_waitcnt vmcnt(4)" : "+v"(t[g * 4]), "+v"(t[g * 4 + 1]), "+v"(t[g * 4 + 2]), "+v"(t[g * 4 + 3])::"memory");
            else if (g == SKB / 4 - 3) asm volatile("s_waitcnt vmcnt(8)" : "+v"(t[g * 4]), "+v"(t[g * 4 + 1]), "+v"(t[g * 4 + 2]), "+v"(t[g * 4 + 3])::"memory");
            else asm volatile("s_waitcnt vmcnt(12)" : "+v"(t[g * 4]), "+v"(t[g * 4 + 1]), "+v"(t[g * 4 + 2]), "+v"(t[g * 4 + 3])::"memory");
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const int blk = q * SKB + g * 4 + jj;   // accumulator block i * 8 + j in slot order
              acc[blk >> 3][blk & 7] += t[g * 4 + jj];
            }
          }
        }
      }
    }
  };

  int gh = 0, hb = 0;
  for (int ti = 0; ti < my_tiles; ++ti) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int h_begin = (SK && ti == 0) ? sk_k0 : 0, h_end = (SK && ti == my_tiles - 1) ? sk_k1 : nkh;
    // One half K tile (the sequence documented above).  FAST: the step lies in the interior of the work-group's life
    // (gh + AHEAD + 1 < H), where every staging / read-ahead condition holds and the counted wait is the steady-state one - the
    // interior steps run without the five compare-and-branch pairs and the wait ladder of the general form (round 6: the waves of
    // a SIMD sit in the same phase behind the step's barrier, so every cycle of this bookkeeping is a cycle the matrix pipe idles:
    // profiles/r06_1x1_k512.txt).  MEASURED NOT FASTER (26 instructions and 7 branches fewer per step, 0 ... 4 % slower on the 256 x 256
    // configurations: profiles/r06_tile_loop_experiments.txt), so the default is the single general loop; -DU2_TILE_PEEL=1 selects this form.
    auto step = [&](auto fast_tag) {
      constexpr bool FAST = decltype(fast_tag)::value;
      const int nb = (hb + 1 == RING) ? 0 : hb + 1;
      const int sb = (hb == 0) ? RING - 1 : hb - 1;  // buffer of stage gh + AHEAD (= gh - 1 mod RING)
      if constexpr (KT == 2) {
        // ---- first half of the stage: no staging of pixels, no wait, no barrier; the second half's fragments are in the same buffer
        U2_TILE_SETPRIO(1);
        U2_T_MFMA(0, wfA[0], 0); U2_T_MFMA(1, wfA[1], 0); U2_T_MFMA(0, wfA[0], 1); U2_T_MFMA(1, wfA[1], 1);
        U2_TILE_SETPRIO(0);
        __builtin_amdgcn_sched_barrier(0);
        pf[4] = ldp(hb, 4); pf[5] = ldp(hb, 5); pf[6] = ldp(hb, 6); pf[7] = ldp(hb, 7);
        wfB[0] = ldw(hb, 2); wfB[1] = ldw(hb, 3);
        if (FAST || gh + AHEAD < H) stage_weights(sb);   // the buffer of stage gh - 1: every wave left it at the last barrier
        __builtin_amdgcn_sched_barrier(0);
        U2_TILE_SETPRIO(1);
        U2_T_MFMA(0, wfA[0], 2); U2_T_MFMA(1, wfA[1], 2); U2_T_MFMA(0, wfA[0], 3); U2_T_MFMA(1, wfA[1], 3);
#pragma unroll
        for (int j = 4; j < 8; ++j) { U2_T_MFMA(0, wfA[0], j); U2_T_MFMA(1, wfA[1], j); }
#pragma unroll
        for (int j = 0; j < 4; ++j) { U2_T_MFMA(2, wfB[0], j); U2_T_MFMA(3, wfB[1], j); }
        U2_TILE_SETPRIO(0);
        __builtin_amdgcn_sched_barrier(0);
        wfA[0] = ldw(hb, 0, 1); wfA[1] = ldw(hb, 1, 1);
        pf[0] = ldp(hb, 0, 1); pf[1] = ldp(hb, 1, 1); pf[2] = ldp(hb, 2, 1); pf[3] = ldp(hb, 3, 1);
        __builtin_amdgcn_sched_barrier(0);
        U2_TILE_SETPRIO(1);
#pragma unroll
        for (int j = 4; j < 8; ++j) { U2_T_MFMA(2, wfB[0], j); U2_T_MFMA(3, wfB[1], j); }
        U2_TILE_SETPRIO(0);
      }
      constexpr int LH = KT - 1;   // the half of the stage the sequence below multiplies (its fragments 0-3 / first weight pair are in registers)
      // phase A
#if U2_TILE_READS_FIRST
      // round 6 experiment (MEASURED NOT FASTER, default off): the fragments the second half of phase A needs are requested at its very start (their registers are
      // free: the last MFMAs of the previous step read them), so that eight MFMAs and the weight staging cover their LDS round trip
      // instead of four MFMAs - all eight waves request 6 KB each at the same moment, ~190 cycles of LDS bandwidth alone
      if constexpr (KT == 1) {
        {
          const unsigned pa = lds_base + (unsigned)(hb * BUF + pfrag0), wa = lds_base + (unsigned)(hb * BUF + wfrag0);
          U2_T_RD(pf[4], pa, 4 * 16 * ROWB); U2_T_RD(pf[5], pa, 5 * 16 * ROWB); U2_T_RD(pf[6], pa, 6 * 16 * ROWB);
          U2_T_RD(pf[7], pa, 7 * 16 * ROWB); U2_T_RD(wfB[0], wa, 2 * 16 * ROWB); U2_T_RD(wfB[1], wa, 3 * 16 * ROWB);
        }
        // the six reads of the previous step's phase B (or the prologue's) have returned once at most six are outstanding
        U2_T_WAIT6(6, wfA[0], wfA[1], pf[0], pf[1], pf[2], pf[3]);
        __builtin_amdgcn_sched_barrier(0);
        U2_TILE_SETPRIO(1);
#pragma unroll
        for (int j = 0; j < 4; ++j) { U2_T_MFMA(0, wfA[0], j); U2_T_MFMA(1, wfA[1], j); }
        U2_TILE_SETPRIO(0);
        __builtin_amdgcn_sched_barrier(0);
        if (FAST || gh + AHEAD < H) stage_weights(sb);
        U2_T_WAIT6(0, pf[4], pf[5], pf[6], pf[7], wfB[0], wfB[1]);
        __builtin_amdgcn_sched_barrier(0);
        U2_TILE_SETPRIO(1);
#pragma unroll
        for (int j = 4; j < 8; ++j) { U2_T_MFMA(0, wfA[0], j); U2_T_MFMA(1, wfA[1], j); }
        U2_TILE_SETPRIO(0);
      } else
#endif
      {
      U2_TILE_SETPRIO(1);
      U2_T_MFMA(0, wfA[0], 0); U2_T_MFMA(1, wfA[1], 0); U2_T_MFMA(0, wfA[0], 1); U2_T_MFMA(1, wfA[1], 1);
      U2_TILE_SETPRIO(0);
      __builtin_amdgcn_sched_barrier(0);
      U2_SUBK(3);
#if U2_TILE_W_FIRST
      // round 6 experiment: the weight staging in FRONT of phase A's fragment reads (an LDS-DMA instruction issued behind six
      // outstanding ds_read_b128 is the expensive case of MI355X_MICROARCH.md's price list)
      if constexpr (KT == 1) {
        if (FAST || gh + AHEAD < H) stage_weights(sb);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
      pf[4] = ldp(hb, 4, LH); pf[5] = ldp(hb, 5, LH); pf[6] = ldp(hb, 6, LH); pf[7] = ldp(hb, 7, LH);
      wfB[0] = ldw(hb, 2, LH); wfB[1] = ldw(hb, 3, LH);
#if !U2_TILE_W_FIRST
      if constexpr (KT == 1) {
        if (FAST || gh + AHEAD < H) stage_weights(sb);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
      U2_SUBK(4);
      U2_TILE_SETPRIO(1);
      U2_T_MFMA(0, wfA[0], 2); U2_T_MFMA(1, wfA[1], 2); U2_T_MFMA(0, wfA[0], 3); U2_T_MFMA(1, wfA[1], 3);
#pragma unroll
      for (int j = 4; j < 8; ++j) { U2_T_MFMA(0, wfA[0], j); U2_T_MFMA(1, wfA[1], j); }
      U2_TILE_SETPRIO(0);
      U2_SUBK(5);
      }
      // phase B
      if constexpr (FAST) {
        wait_vm<LPT * (AHEAD - 1)>();
      } else {
        const int rem = H - 2 - gh;  // stages staged behind gh + 1
        if (rem >= AHEAD - 1) wait_vm<LPT * (AHEAD - 1)>();
        else if (AHEAD > 3 && rem == 2) wait_vm<LPT * 2>();
        else if (rem == 1) wait_vm<LPT>();
        else wait_vm<0>();
      }
      U2_SUBK(6);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#ifdef U2_TILE_TRACE
      if (gh < 44) U2_STAMP(2 + gh);
      U2_SUB(0);
#endif
      // U2_TILE_LATE_PIXELS (round 6 experiment): the pixel staging of the step behind the first eight MFMAs of phase B instead of in
      // front of them - all eight waves leave the barrier together and queue on the CU's one vector-memory path (16 instructions
      // of 16 cycles each) before any of them issues an MFMA; the order of the VMEM operations of a wave is unchanged
#if !U2_TILE_LATE_PIXELS
      if (FAST || gh + AHEAD + 1 < H) stage_pixels(hb);
#endif
      __builtin_amdgcn_sched_barrier(0);
      U2_TILE_SETPRIO(1);
#pragma unroll
      for (int j = 0; j < 4; ++j) { U2_T_MFMA(2, wfB[0], j); U2_T_MFMA(3, wfB[1], j); }
      U2_TILE_SETPRIO(0);
      __builtin_amdgcn_sched_barrier(0);
      U2_SUBK(1);
#if U2_TILE_LATE_PIXELS
      if (FAST || gh + AHEAD + 1 < H) stage_pixels(hb);
      __builtin_amdgcn_sched_barrier(0);
#endif
      if (FAST || gh + 1 < H) {
#if U2_TILE_READS_FIRST
        if constexpr (KT == 1) {
          const unsigned pa = lds_base + (unsigned)(nb * BUF + pfrag0), wa = lds_base + (unsigned)(nb * BUF + wfrag0);
          U2_T_RD(wfA[0], wa, 0); U2_T_RD(wfA[1], wa, 16 * ROWB);
          U2_T_RD(pf[0], pa, 0); U2_T_RD(pf[1], pa, 16 * ROWB); U2_T_RD(pf[2], pa, 2 * 16 * ROWB); U2_T_RD(pf[3], pa, 3 * 16 * ROWB);
        } else
#endif
        {
        wfA[0] = ldw(nb, 0); wfA[1] = ldw(nb, 1);
        pf[0] = ldp(nb, 0); pf[1] = ldp(nb, 1); pf[2] = ldp(nb, 2); pf[3] = ldp(nb, 3);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      U2_TILE_SETPRIO(1);
#pragma unroll
      for (int j = 4; j < 8; ++j) { U2_T_MFMA(2, wfB[0], j); U2_T_MFMA(3, wfB[1], j); }
      U2_TILE_SETPRIO(0);
      U2_SUBK(2);
      ++gh;
      hb = nb;
    };
#if U2_TILE_PINGPONG
    // Round 6 experiment (profiles/r06_tile_substep.txt): the two wave groups of the work-group - waves 0 .. NW/2-1 and the rest, one wave
    // of each on every SIMD - run each HALF step in opposite order: group X stages and reads first and multiplies then, group Y
    // multiplies first (on fragments it read at the end of the previous half) and stages / reads for the next half then, so that
    // on every SIMD one wave issues LDS-DMA and LDS reads while the other owns the matrix pipe.  Half A = weight rows 0-1 x pixel
    // fragments 0-7 (ten fragment reads), half B = weight rows 2-3 x the same pixel fragments (two reads).  One barrier per half:
    // #1 (behind half A, with the counted vmcnt wait in front of it) publishes stage gh + 1 and frees the pixel region of this
    // stage's buffer; #2 (end of the step) frees its weight region for the next step's weight staging.  Group Y does not read ahead
    // across a tile boundary (the first step of a tile is "cold" for both groups): no fragment register is live in the epilogue.
    // Same products into the same accumulators in the same K order as the lock-step form: bit-identical results.
    auto step_pp = [&](bool first_of_tile, bool last_of_tile) {
      constexpr bool FAST = false;
      const int nb = (hb + 1 == RING) ? 0 : hb + 1;
      const int sb = (hb == 0) ? RING - 1 : hb - 1;
      const bool grp_x = w < NW / 2;
      const bool cold = grp_x || first_of_tile;
      const unsigned pa = lds_base + (unsigned)(hb * BUF + pfrag0), wa = lds_base + (unsigned)(hb * BUF + wfrag0);
#define U2_PP_WAIT10()                                                                                                          \
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wfA[0]), "+v"(wfA[1]), "+v"(pf[0]), "+v"(pf[1]), "+v"(pf[2]), "+v"(pf[3])::"memory");  \
      asm volatile("" : "+v"(pf[4]), "+v"(pf[5]), "+v"(pf[6]), "+v"(pf[7])::"memory")
#define U2_PP_MA()                                                                                                               \
      U2_TILE_SETPRIO(1);                                                                                                        \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) { U2_T_MFMA(0, wfA[0], j); U2_T_MFMA(1, wfA[1], j); }                        \
      U2_TILE_SETPRIO(0)
#define U2_PP_MB()                                                                                                               \
      U2_TILE_SETPRIO(1);                                                                                                        \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) { U2_T_MFMA(2, wfB[0], j); U2_T_MFMA(3, wfB[1], j); }                        \
      U2_TILE_SETPRIO(0)
      // ---------------- half A  (ONE copy of every MFMA group: the group-dependent pieces around them are reads and staging only)
      if (cold) {
        U2_T_RD(wfA[0], wa, 0); U2_T_RD(wfA[1], wa, 16 * ROWB);
        U2_T_RD(pf[0], pa, 0); U2_T_RD(pf[1], pa, 16 * ROWB); U2_T_RD(pf[2], pa, 2 * 16 * ROWB); U2_T_RD(pf[3], pa, 3 * 16 * ROWB);
        U2_T_RD(pf[4], pa, 4 * 16 * ROWB); U2_T_RD(pf[5], pa, 5 * 16 * ROWB); U2_T_RD(pf[6], pa, 6 * 16 * ROWB); U2_T_RD(pf[7], pa, 7 * 16 * ROWB);
        if (gh + AHEAD < H) stage_weights(sb);
      }
      __builtin_amdgcn_sched_barrier(0);
      U2_SUBK(11);
      U2_PP_WAIT10();   // cold: the reads just issued; otherwise the ones this wave issued at the end of the previous half B
      U2_SUBK(12);
      U2_PP_MA();
      __builtin_amdgcn_sched_barrier(0);
      U2_SUBK(13);
      if (!cold) {
        if (gh + AHEAD < H) stage_weights(sb);
      }
      if (!grp_x) { U2_T_RD(wfB[0], wa, 2 * 16 * ROWB); U2_T_RD(wfB[1], wa, 3 * 16 * ROWB); }   // group Y: half B's weights now
      __builtin_amdgcn_sched_barrier(0);
      U2_SUBK(14);
      {
        const int rem = H - 2 - gh;  // stages staged behind gh + 1
        if (rem >= AHEAD - 1) wait_vm<LPT * (AHEAD - 1)>();
        else if (AHEAD > 3 && rem == 2) wait_vm<LPT * 2>();
        else if (rem == 1) wait_vm<LPT>();
        else wait_vm<0>();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#ifdef U2_TILE_TRACE
      if (gh < 44) U2_STAMP(2 + gh);
      U2_SUB(0);
#endif
      // ---------------- half B
      if (grp_x) {
        U2_T_RD(wfB[0], wa, 2 * 16 * ROWB); U2_T_RD(wfB[1], wa, 3 * 16 * ROWB);
        if (gh + AHEAD + 1 < H) stage_pixels(hb);
      }
      __builtin_amdgcn_sched_barrier(0);
      U2_SUBK(15);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wfB[0]), "+v"(wfB[1])::"memory");
      U2_SUBK(16);
      U2_PP_MB();
      __builtin_amdgcn_sched_barrier(0);
      U2_SUBK(17);
      if (!grp_x) {
        if (gh + AHEAD + 1 < H) stage_pixels(hb);
        if (!last_of_tile && gh + 1 < H) {   // next step's half A fragments (stage gh + 1: published by barrier #1)
          const unsigned pn = lds_base + (unsigned)(nb * BUF + pfrag0), wn = lds_base + (unsigned)(nb * BUF + wfrag0);
          U2_T_RD(wfA[0], wn, 0); U2_T_RD(wfA[1], wn, 16 * ROWB);
          U2_T_RD(pf[0], pn, 0); U2_T_RD(pf[1], pn, 16 * ROWB); U2_T_RD(pf[2], pn, 2 * 16 * ROWB); U2_T_RD(pf[3], pn, 3 * 16 * ROWB);
          U2_T_RD(pf[4], pn, 4 * 16 * ROWB); U2_T_RD(pf[5], pn, 5 * 16 * ROWB); U2_T_RD(pf[6], pn, 6 * 16 * ROWB); U2_T_RD(pf[7], pn, 7 * 16 * ROWB);
        }
      }
      U2_SUBK(18);
      __builtin_amdgcn_s_barrier();   // #2: the weight region of this stage's buffer is free for the next step's weight staging
      asm volatile("" ::: "memory");
#undef U2_PP_WAIT10
#undef U2_PP_MA
#undef U2_PP_MB
      ++gh;
      hb = nb;
    };
#endif
    int h = h_begin;
#if U2_TILE_PINGPONG
    if constexpr (KT == 1 && !ACC) {
      for (; h < h_end; ++h) step_pp(h == h_begin, h + 1 == h_end);
    } else
#endif
    {
#if U2_TILE_PEEL
    while (h < h_end) {
      int nfast = min(h_end - h, H - (AHEAD + 1) - gh);
      for (; nfast > 0; --nfast, ++h) step(std::true_type{});
      if (h < h_end) { step(std::false_type{}); ++h; }
    }
#else
    for (; h < h_end; ++h) step(std::false_type{});
#endif
    }
    if constexpr (SK) {
      if (h_begin > 0) { U2_STAMP(48); sk_publish(); U2_STAMP(49); continue; }   // a non-leading part: handed to the tile's owner
      if (h_end < nkh) break;                                   // the owner collects the parts behind its own: below
    }
    epilogue(first_tile + ti * tstep);
  }
  if constexpr (SK) {
    // (outside the tile loop on purpose: no K loop follows, so the fragment registers and the staging state are dead here
    //  and the additions need no spill - a scratch access inside the tile loop would drain the LDS-DMA queue every half tile)
    if (sk_k1 < nkh && !(my_tiles == 1 && sk_k0 > 0)) {
      U2_STAMP(50);
      sk_combine(first_tile + my_tiles - 1);
      U2_STAMP(52);
      epilogue(first_tile + my_tiles - 1);
      U2_STAMP(53);
    }
  }
  if (!ACC && a.stats) wg_flush_column_sums<NW>(a.stats, a.N, st_n, st_s, st_ss, w, lane, smem);
  U2_STAMP(54);
#undef U2_T_MFMA
}

// stream-K scratch: one fp32 partial-tile slot and one flag per work-group, per (device, stream) - convolutions of independent
// branches run concurrently on several streams, and a process may drive several devices.  Allocated on first use under a mutex
// (the autograd backward thread and the forward thread both get here), flags zeroed once: the kernel leaves every flag it
// consumed at zero.  64 MB per entry: the largest launch is 256 work-groups x 256 x 256 fp32 (one group per CU) or 512 x
// 128 x 256 (two per CU), both 64 MB - a work-group's slot starts at blockIdx.x * TM * TN * 4.
//
// Progress: owners wait for peers at the END of their life, peers publish at the START of theirs, and nobody
// waits before having published.  With the whole grid resident that is trivially deadlock-free.  When another kernel holds some
// CUs (a side-stream weight gradient, a second stream-K convolution) only the LAST resident work-group of an XCD's chain can be
// waiting for a peer that has not been dispatched yet - every other owner's peers are resident and have published - so all
// but at most eight work-groups of the launch run to completion and free their CUs; the hardware dispatches work-groups of a
// grid in index order, so the missing peer is the next one to get a CU.  The scheme relies on that in-order dispatch (as
// every persistent-kernel hand-over does); it does NOT rely on co-residency of the whole grid.
//
// Round 6: MEASURED OTHERWISE.  With the semantic / mask head's kernels on their own streams beside it, the driver's bench command hung
// in 4 of 23 runs (tools/exp/hang_hunt.sh: the chip never finishes a step), in 0 of 24 with this form forbidden (variant bit 28), in 0
// of 40 with those streams off, and still in 3 of 30 with the stream-K launches of different streams ordered behind each other by
// events - one launch of this form beside kernels of another stream is enough.  What the argument above misses is not understood.
// The product path (layers/functional.py:streamk_region) asks for this form only where no branch stream has work in flight and
// passes bit 28 elsewhere; callers with several streams must do the same.
constexpr int SK_MAX_GROUPS = 512;
constexpr size_t SK_WS_LIMIT = (size_t)64 << 20;
bool sk_scratch(hipStream_t s, ConvArgs& a, size_t need_bytes) {
  float* ws = (float*)scratch_get(0, s, need_bytes, SK_WS_LIMIT, false);
  if (!ws) return false;
  int* flags = (int*)scratch_get(1, s, SK_MAX_GROUPS * sizeof(int), (size_t)2 << 20, true);
  if (!flags) return false;
  a.sk_ws = ws; a.sk_flags = flags;
  return true;
}

// sk: 0 = never, 1 = where the tile count fills its last round badly, 2 = always (tests)
#ifdef U2_TILE_TRACE
// Host side of the phase trace: a [1024][64] buffer of s_memtime stamps, cleared before and printed after every U2_TILE_TRACE_EVERY-th
// launch (default 4: the timed launches of selftest bench2 come in fours) - phase durations in microseconds at 100 MHz ticks
// (s_memtime on gfx950 counts the constant 100 MHz reference clock), for a handful of work-groups and the median over all.
unsigned long long* g_tile_trace_host_ptr = nullptr;
unsigned long long* g_tile_sub_host_ptr = nullptr;
int g_tile_trace_launch = 0;
void tile_trace_begin(int G, hipStream_t s) {
  (void)G;
  if (!g_tile_trace_host_ptr) {
    if (hipMalloc(&g_tile_trace_host_ptr, 1024 * 64 * 8) != hipSuccess) return;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_trace_dev), &g_tile_trace_host_ptr, sizeof(void*));
  }
  (void)hipMemsetAsync(g_tile_trace_host_ptr, 0, 1024 * 64 * 8, s);
#if U2_TILE_TRACE_POINT
  if (!g_tile_sub_host_ptr) {
    if (hipMalloc(&g_tile_sub_host_ptr, 1024 * 2 * 96 * 8) != hipSuccess) return;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_sub_dev), &g_tile_sub_host_ptr, sizeof(void*));
  }
  (void)hipMemsetAsync(g_tile_sub_host_ptr, 0, 1024 * 2 * 96 * 8, s);
#endif
}
void tile_trace_end(int G, hipStream_t s, const char* what) {
  static const int every = getenv("U2_TILE_TRACE_EVERY") ? atoi(getenv("U2_TILE_TRACE_EVERY")) : 4;
  if (!g_tile_trace_host_ptr || (++g_tile_trace_launch % every) != 0 || G > 1024) return;
  (void)hipStreamSynchronize(s);
  static unsigned long long h[1024 * 64];
  if (hipMemcpy(h, g_tile_trace_host_ptr, (size_t)G * 64 * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
#if U2_TILE_TRACE_POINT
  {
    // sub-step timeline: cycles from the step's barrier to point U2_TILE_TRACE_POINT, waves 0 and 4 of a few work-groups, mean over
    // the interior steps; points 3-6 lie in front of the barrier of their step and are counted from the PREVIOUS barrier
    static unsigned long long sb[1024 * 2 * 96];
    if (hipMemcpy(sb, g_tile_sub_host_ptr, (size_t)G * 2 * 96 * 8, hipMemcpyDeviceToHost) == hipSuccess) {
      const int picks[] = {1, 9, G / 2 + 1, G - 15};
      for (int pi = 0; pi < 4; ++pi) {
        const int g = picks[pi];
        if (g < 0 || g >= G) continue;
        for (int wv = 0; wv < 2; ++wv) {
          const unsigned long long* r = sb + ((size_t)g * 2 + wv) * 96;
          double step = 0, off = 0; int ns = 0, no = 0;
          for (int k = 6; k < 30; ++k) {
            if (r[k] && r[k + 1]) { step += (double)(r[k + 1] - r[k]); ++ns; }
            const int ref = ((U2_TILE_TRACE_POINT >= 3 && U2_TILE_TRACE_POINT <= 6) || (U2_TILE_TRACE_POINT >= 11 && U2_TILE_TRACE_POINT <= 14)) ? k - 1 : k;
            if (r[48 + k] && r[ref] && r[48 + k] > r[ref]) { off += (double)(r[48 + k] - r[ref]); ++no; }
          }
          const unsigned long long* r0 = sb + ((size_t)g * 2) * 96;
          fprintf(stderr, "SUB point %d wg %4d wave %d: step %.0f cycles (n=%d), point at +%.0f cycles behind the %s barrier (n=%d), barrier of wave 4 - wave 0 at step 10: %lld\n",
                  U2_TILE_TRACE_POINT, g, wv * 4, ns ? step / ns : -1.0, ns, no ? off / no : -1.0, ((U2_TILE_TRACE_POINT >= 3 && U2_TILE_TRACE_POINT <= 6) || (U2_TILE_TRACE_POINT >= 11 && U2_TILE_TRACE_POINT <= 14)) ? "previous" : "same", no,
                  (long long)(r0[96 + 10]) - (long long)(r0[10]));
        }
      }
    }
  }
#endif
  static const double tick_us = getenv("U2_TILE_TRACE_TICK_US") ? atof(getenv("U2_TILE_TRACE_TICK_US")) : 0.01;
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int g = 0; g < G; ++g) {
    if (h[g * 64] && h[g * 64] < t0) t0 = h[g * 64];
    for (int k = 0; k < 64; ++k) if (h[g * 64 + k] > t1) t1 = h[g * 64 + k];
  }
  fprintf(stderr, "TRACE launch %d (%s, %d work-groups): first entry -> last stamp %.2f us (%llu ticks of %.4f us)\n", g_tile_trace_launch, what, G, (t1 - t0) * tick_us, t1 - t0, tick_us);
  auto us = [&](int g, int k) { return h[g * 64 + k] ? (double)(h[g * 64 + k] - t0) * tick_us : -1.0; };
  const int picks[] = {0, 1, 8, 9, 64, 65, 128, G / 2 + 1, G - 16, G - 8, G - 1};
  for (int pi = 0; pi < (int)(sizeof(picks) / sizeof(int)); ++pi) {
    const int g = picks[pi];
    if (g < 0 || g >= G) continue;
    int steps = 0;
    double first_step = -1, last_step = -1;
    for (int k = 2; k < 46; ++k)
      if (h[g * 64 + k]) { ++steps; if (first_step < 0) first_step = us(g, k); last_step = us(g, k); }
    fprintf(stderr, "  wg %4d: entry %6.2f  first stage %6.2f  K barriers %2d (%6.2f .. %6.2f, %.3f us/step)  publish %6.2f .. %6.2f  loop done %6.2f  flag %6.2f  combined %6.2f  epilogue %6.2f  end %6.2f\n",
            g, us(g, 0), us(g, 1), steps, first_step, last_step, steps > 1 ? (last_step - first_step) / (steps - 1) : 0.0, us(g, 48), us(g, 49),
            us(g, 50), us(g, 51), us(g, 52), us(g, 53), us(g, 54));
  }
  // per-step gaps of one work-group, to see stalls inside the K loop
  const int g = G / 2 + 1 < G ? G / 2 + 1 : 0;
  fprintf(stderr, "  wg %4d step gaps (us):", g);
  for (int k = 3; k < 46; ++k)
    if (h[g * 64 + k] && h[g * 64 + k - 1]) fprintf(stderr, " %.2f", (double)(h[g * 64 + k] - h[g * 64 + k - 1]) * tick_us);
  fprintf(stderr, "\n");
  // medians over all work-groups
  std::vector<double> d_entry, d_first, d_step, d_pub, d_comb, d_epi, d_end;
  for (int w = 0; w < G; ++w) {
    if (!h[w * 64]) continue;
    d_entry.push_back(us(w, 0));
    if (h[w * 64 + 1]) d_first.push_back(us(w, 1) - us(w, 0));
    int steps = 0; double fs = -1, ls = -1;
    for (int k = 2; k < 46; ++k) if (h[w * 64 + k]) { ++steps; if (fs < 0) fs = us(w, k); ls = us(w, k); }
    if (steps > 1) d_step.push_back((ls - fs) / (steps - 1));
    if (h[w * 64 + 48] && h[w * 64 + 49]) d_pub.push_back(us(w, 49) - us(w, 48));
    if (h[w * 64 + 50] && h[w * 64 + 52]) d_comb.push_back(us(w, 52) - us(w, 50));
    if (h[w * 64 + 52] && h[w * 64 + 53]) d_epi.push_back(us(w, 53) - us(w, 52));
    if (h[w * 64 + 54]) d_end.push_back(us(w, 54));
  }
  auto med = [](std::vector<double>& v) { if (v.empty()) return -1.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  auto mx = [](std::vector<double>& v) { return v.empty() ? -1.0 : *std::max_element(v.begin(), v.end()); };
  const double e_max = mx(d_entry), end_med = med(d_end), end_max = mx(d_end);
  fprintf(stderr, "  medians: entry %.2f (last %.2f)  entry->first stage %.2f  us/step %.3f  publish %.2f (n=%zu)  combine %.2f (n=%zu, max %.2f)  epilogue %.2f  end %.2f (last %.2f)\n",
          med(d_entry), e_max, med(d_first), med(d_step), med(d_pub), d_pub.size(), med(d_comb), d_comb.size(), mx(d_comb), med(d_epi), end_med, end_max);
}
#endif

template <int WCH, int WPX, int RING, bool ACC = false, int KT = 1>
int launch_cfg(ConvArgs& a, int N, int per_cu, int tiny_grid, hipStream_t s, int sk = 0) {
  constexpr int TN = WCH * 64, TM = WPX * 128;
  constexpr int LDS = RING * (TM + TN) * 64 * KT;
  a.tiles_m = (a.M + TM - 1) / TM;
  a.tiles_n = (N + TN - 1) / TN;
  const long long T = (long long)a.tiles_m * a.tiles_n;
  if (T >= (1 << 30)) return 0;
  long long cap = tiny_grid ? (sk == 2 ? 24 : 8) : 256LL * per_cu;
  if constexpr (!ACC) {
    const long long nkh = (long long)a.ntaps * (a.C / (32 * KT));   // stages per tile
    const long long rounds = (T + cap - 1) / cap;
    // Measured (tests/native/selftest bench2, profiles/r03_conv_streamk.txt): equal shares pay for the 256 x 256 configurations
    // on deep reductions when whole tiles fill the chip badly - 263 tiles (3x3 256->256 at stride 16: 642 -> 866 TFLOP/s), 132
    // tiles (res5 3x3, fc1: +14 % / +48 %) - or fill the last of several rounds badly (3x3 at stride 8: +7.5 %, fc1 data
    // gradient +10 %).  They lose on the two-groups-per-CU configurations (1x1 layers, -5...-25 %: the hand-over costs more than
    // the second, independent work-group hides) and where nearly every tile would be split for a small gain (mask head 3x3,
    // 203 tiles: -13 %).  The stream-K instantiation also spills a dozen registers (reloaded once per filter tap).
    const double util = (double)T / (double)(rounds * cap);
    const bool want = sk == 2 || (sk == 1 && WCH == 4 && WPX == 2 && T >= 8 && nkh * KT >= 32 && (util < 0.7 || (util < 0.9 && T >= 512) || (nkh * KT >= 128 && T >= 512)));
    long long Gs = cap;
    while (Gs > 8 && (T / 8) * nkh / (Gs / 8) < RING + 1) Gs -= 8;
    if (want && Gs <= SK_MAX_GROUPS && (T / 8) * nkh / (Gs / 8) >= RING + 1 && sk_scratch(s, a, (size_t)Gs * TM * TN * 4)) {
      static PerDeviceOnce attr_set_sk;
      if (auto once_guard = attr_set_sk.first()) {
        (void)hipFuncSetAttribute((const void*)conv_tile_kernel<WCH, WPX, RING, false, true, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      }
      g_last_conv_kernel += 400;  // 5xx: the stream-K form of configuration xx
#ifdef U2_TILE_TRACE
      tile_trace_begin((int)Gs, s);
#endif
      hipLaunchKernelGGL((conv_tile_kernel<WCH, WPX, RING, false, true, KT>), dim3((unsigned)Gs), dim3(WCH * WPX * 64), LDS, s, a);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return -1000 - (int)e;
#ifdef U2_TILE_TRACE
      tile_trace_end((int)Gs, s, "stream-K");
#endif
      return 1;
    }
  }
  long long G = T < cap ? T : cap;
  G = (G + 7) & ~7LL;
  static PerDeviceOnce attr_set;
  if (auto once_guard = attr_set.first()) {
    (void)hipFuncSetAttribute((const void*)conv_tile_kernel<WCH, WPX, RING, ACC, false, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
#ifdef U2_TILE_TRACE
  tile_trace_begin((int)G, s);
#endif
  hipLaunchKernelGGL((conv_tile_kernel<WCH, WPX, RING, ACC, false, KT>), dim3((unsigned)G), dim3(WCH * WPX * 64), LDS, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return -1000 - (int)e;
#ifdef U2_TILE_TRACE
  tile_trace_end((int)G, s, "whole tiles");
#endif
  return 1;
}

}  // namespace

// variant bits 12-15 select the tile configuration: 0 = automatic, 1 = 256ch x 256px ring 4, 2 = 256 x 256 ring 5,
// 3 = 128ch x 256px (ring 3, two groups per CU), 4 = 256ch x 128px (ring 3, two groups per CU), 5 = 128ch x 256px ring 4,
// 6 = 64ch x 512px ring 4 (layers with <= 64 output channels: no MFMA spent on absent channels),
// 7 = 256 x 256 with whole K tiles (64 channels, 128-byte rows) in a ring of 2 (C % 64 == 0; the long-row 1x1 layers),
// 15 = never (conv_igemm.hip kernels only); bit 16: 8 work-groups only (tests: forces several tiles per work-group).
// Tried and removed (numbers in profiles/r02_conv_ablation.txt, r02_conv_pingpong.txt): a j-split schedule
// with every LDS read 12+ MFMAs ahead of its use (+0 %), and a ping-pong schedule with the two wave groups half a sequence
// apart (-8 %).
int launch_conv_tile(ConvArgs& a, int N, int C, int variant, hipStream_t s) {
  int sel = (variant >> 12) & 15;
  const int tiny = (variant >> 16) & 1;
  if (sel == 15) return 0;
  if ((C & 31) || (N & 7) || (a.out_ld & 7) || a.ntaps < 1 || a.M >= (1 << 24) || a.M < 1) return 0;
  // the accumulating epilogue exists for configuration 4 (256ch x 128px: the 1x1 conv3 of a residual block), no statistics
  if (a.accumulate && (a.stats || a.remap_out || (sel != 0 && sel != 4) || N < 128)) return 0;
  // 32-bit byte offsets inside the kernel
  if ((unsigned long long)a.B * a.Hin * a.Win * a.in_ld * 2ull >= 0xffffffffull || (unsigned long long)N * a.wt_taps * C * 2ull >= 0xfffffff0ull) return 0;
  const int nkh = a.ntaps * (C >> 5);
  if (sel == 0) {
    // Automatic choice, from the per-layer A/B of tests/native/selftest bench2 on the shapes of the u2seg_R50_800 step
    // (profiles/r02_conv_variants.txt): tile shape by output width and by how many whole rounds of tiles the chip gets.

    const long long K = (long long)a.ntaps * C;
    if (a.accumulate) sel = 4;
    const long long t256 = (long long)((a.M + 255) / 256) * ((N + 255) / 256);
    if (sel == 4) {
      // accumulate: fixed above
    } else if (N <= 128) {
      // 128-wide tiles (two groups per CU); the deep 3x3 layers with 128 outputs stay on the 128 x 128 BK-64 kernel
      // (configuration 6, 64ch x 512px with one group per CU, measured slower on the 64-channel layers: 364 vs 451 TFLOP/s)
      if (a.M >= 100000 && !(a.ntaps > 1 && K >= 2048)) sel = 3;
    } else if (a.ntaps > 1) {
      if (t256 >= 2048) sel = 2;                                   // stride-4 maps: >= 8 rounds of 256 x 256 tiles
      else if (t256 >= 768) sel = 1;                               // stride-8 maps
      else if (t256 >= 180 && t256 <= 256 && K >= 2048) sel = 1;   // one nearly full round (mask head 3x3)
      else if (t256 > 256 && t256 < 768 && K >= 2048) sel = 1;     // stride-16 maps: 263 tiles, as equal stream-K shares
      else if (t256 >= 96 && K >= 8192) sel = 1;                   // fc1 (7x7 on 256 channels, 128 tiles): stream-K, 1024 -> 1160
      else if (N >= 512 && t256 >= 128 && K >= 4096 && a.M >= 16000) sel = 1;  // res5 3x3 (not the 7x7 fc1: M = 8192)
    } else {
      if (N >= 4096) sel = 1;                                      // fc1 data gradient (N = 12544)
      else if (a.M < 10000 && N <= 1024) sel = 0;                  // small fully connected layers
      else if (K >= 2048) sel = 2;
      // round 5 (profiles/r05_conv_laggards.txt): the stride-16 1x1 layers with K = 1024 (res4 conv1 and the data gradient of
      // conv3, 525 tiles of 256 ch x 128 px on 512 slots = two rounds, the second one 13 tiles) run 11 % faster as 263 tiles of
      // 256 x 256 in equal stream-K shares: 0.065 -> 0.058 ms
      else if (K >= 1024 && N >= 256 && t256 > 256 && t256 < 768) sel = 1;
      else sel = 4;                                                // 1x1 layers: 256 ch x 128 px, two groups per CU
    }
    if (sel == 0) return 0;
  }
  if (sel > 7) return 0;
  if (sel == 7) {
    if ((C & 63) || a.accumulate) return 0;
    const int nks = a.ntaps * (C >> 6);
    const long long T7 = (long long)((a.M + 255) / 256) * ((N + 255) / 256);
    if (nks * (T7 < 256 ? 1 : T7 / 256) < 3) return 0;   // a work-group needs ring + 1 stages over all its tiles
    g_last_conv_kernel = 107;
    const int sk7 = ((variant >> 28) & 1) ? 0 : ((variant >> 27) & 1) ? 2 : 1;
    return launch_cfg<4, 2, 2, false, 2>(a, N, 1, tiny, s, sk7);
  }
  int ring = (sel == 2) ? 5 : (sel == 1 || sel == 5 || sel == 6) ? 4 : 3;
  if (nkh < ring) {
    // a work-group only needs RING half tiles over ALL the tiles it walks; in automatic mode fall back to the ring-3
    // configurations (K >= 64 with two or more tiles per work-group, K >= 96 otherwise)
    if ((variant >> 12 & 15) != 0) return 0;
    if (a.accumulate && N <= 128) return 0;  // the accumulating epilogue exists for configuration 4 only
    sel = (N <= 128) ? 3 : 4;
    ring = 3;
    const int TN = sel == 3 ? 128 : 256, TM = sel == 3 ? 256 : 128;
    const long long T = (long long)((a.M + TM - 1) / TM) * ((N + TN - 1) / TN);
    const long long min_tiles = (T >> 3) / 64;  // 512 work-groups: 64 per XCD
    if (nkh * min_tiles < ring) return 0;
  }
  g_last_conv_kernel = 100 + sel;
  // stream-K form (equal shares of the (tile, half K tile) sequence): automatic where whole tiles fill the last round badly;
  // variant bit 27 forces it, bit 28 forbids it.  g_last_conv_kernel: 500 + configuration.
  const int sk = ((variant >> 28) & 1) ? 0 : ((variant >> 27) & 1) ? 2 : 1;
  switch (sel) {
    case 1: return launch_cfg<4, 2, 4>(a, N, 1, tiny, s, sk);
    case 2: return launch_cfg<4, 2, 5>(a, N, 1, tiny, s, sk);
    case 3: return launch_cfg<2, 2, 3>(a, N, 2, tiny, s, sk);
    case 4: return a.accumulate ? launch_cfg<4, 1, 3, true>(a, N, 2, tiny, s) : launch_cfg<4, 1, 3>(a, N, 2, tiny, s, sk);
    case 5: return launch_cfg<2, 2, 4>(a, N, 1, tiny, s);
    case 6: return launch_cfg<1, 4, 4>(a, N, 1, tiny, s);
    default: return 0;
  }
}

}  // namespace u2conv
