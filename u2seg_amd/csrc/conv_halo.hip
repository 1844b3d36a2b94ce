// 3x3 / stride 1 / pad 1 convolution (forward and - with the flipped weight layout - data gradient) on CDNA4 MFMA with the
// INPUT HALO staged once per 32-channel slab: F.conv2d behind detectron2/layers/wrappers.py:127-134 for the 3x3 layers of
// backbone/resnet.py:194-210, backbone/fpn.py:126-167, proposal_generator/rpn.py:158-177, meta_arch/semantic_seg.py:246-253.
//
// Why (round-3 measurements, profiles/r03_conv_ablation.txt): the persistent tile kernel (conv_tile.hip) is bound by operand
// delivery, not by MFMA issue - its K loop alone runs at ~93 % of what the clock allows, its LDS-DMA alone needs 0.68 us per
// half K tile against 0.56 us of MFMA, and each of the nine filter taps re-stages the same pixels, shifted by one row or one
// column, as a fresh 16 KB operand.  Here a work-group owns a 16 x 32 patch of output pixels x 128 output channels and keeps
// the 18 x 34 input halo of one 32-channel slab in LDS: every tap reads its pixel fragments from that halo at an address offset
// (dy * pitch + dx), so per slab the work-group stages 39 KB of pixels once + 9 x 8 KB of weights = 111 KB where the tile
// kernel stages 288 KB for the same 512 x 128 x 288 MACs... 2.6 x fewer global_load_lds instructions and L2 -> LDS bytes.
//
//  * halo image: rows of 32 channels (64 B) in raster order with a pitch of 36 pixels (a multiple of 4: the bank of a row
//    then depends on its column only).  Chunk c of halo column hx is stored at chunk c ^ (2 * ((hx >> 2) & 1)): found by
//    exhaustive search, this XOR is conflict-free for ds_read_b128 fragment reads (16 consecutive columns x 4 chunks per
//    16-lane group) at EVERY column offset, which the dx = -1 / 0 / +1 taps need (conv_tile's swizzle is only conflict-free
//    at offsets that are multiples of 4).  Three per-lane base addresses (one per dx) + immediates cover all taps;
//  * double-buffered halo (2 x 48 KB) + a seven-deep ring of (slab, tap) weight slots (7 x 8 KB); the nine taps of a slab are
//    unrolled, each step = 32 MFMA per wave + one raw barrier, with a per-tap table of counted vmcnt values (a step issues one
//    weight piece per wave and, in taps 0-5, one halo piece of the next slab; vmcnt retires in order);
//  * the pipeline runs across slabs AND across tiles (persistent work-groups, XCD-contiguous tile ranges), wave tile
//    64 ch x 128 px and the register-direct epilogue (two 16-byte runs per lane, BN column statistics by a transposing DPP
//    reduction) are conv_tile.hip's.
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "conv_args.h"

namespace u2conv {
namespace {

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// see conv_tile.hip: transposing reduction over the 16 lanes of a DPP row; lane fr leaves with the row total of v[fr]
__device__ __forceinline__ float row16_transpose_sum(float (&v)[16], int fr) {
  {
    const bool up = fr & 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float keep = up ? v[k + 8] : v[k], send = up ? v[k] : v[k + 8];
      v[k] = keep + dpp_mov<0x128>(send);
    }
  }
  {
    const bool up = fr & 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float keep = up ? v[k + 4] : v[k], send = up ? v[k] : v[k + 4];
      v[k] = keep + dpp_mov<0x141>(send);
    }
  }
  {
    const bool up = fr & 2;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float keep = up ? v[k + 2] : v[k], send = up ? v[k] : v[k + 2];
      v[k] = keep + dpp_mov<0x4E>(send);
    }
  }
  const bool up = fr & 1;
  const float keep = up ? v[1] : v[0], send = up ? v[0] : v[1];
  return keep + dpp_mov<0xB1>(send);
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return *reinterpret_cast<const uint32_t*>(&r);
}

// output patch of a work-group: PH rows x PWD columns = 512 pixels, 16 x 32 (template default) or 8 x 64 (maps whose height
// fills 16-row patches badly); halo row pitch PWD + 4 pixels (a multiple of 4; PH + 2 rows x PWD + 2 columns used)
constexpr int HBUF = 49152;              // one halo buffer: 768 rows of 64 B (648 / 680 used)
constexpr int HPIECES = 6;               // 1 KB pieces per wave and slab (8 waves x 6 x 16 rows = 768 rows)
constexpr int WR = 7, WA = WR - 1;       // weight ring depth / how many steps ahead the weight cursor runs
constexpr int TN = 128, WSLOT = TN * 64; // output channels of a tile (CW = 2) / one (slab, tap) weight slot
constexpr int WBASE = 2 * HBUF;
constexpr int LDS_BYTES = WBASE + WR * WSLOT;  // 155648

// OPT bit 0: no s_setprio around the MFMA runs, bit 1: no sched_barrier fences between the MFMA runs and the LDS reads / staging.
// Measured (profiles/r03_conv_halo_sched.txt): without both the kernel is 4 % faster (fpn_output2 1208 -> 1260 TFLOP/s), either
// one alone +2 % / -1 %: the default is 3; variant bits 29-30 select OPT ^ 3 (so 0 there = the default).
// CW: waves along the output channels (64 each).  CW = 2: 128-channel tiles, wave tile 64 ch x 128 px (8 pixel fragments).  CW = 1
// (round 4): 64-channel tiles for the layers with <= 64 output channels (res2 conv2 and its data gradient) - all eight waves along
// the pixels, wave tile 64 ch x 64 px (4 fragments): no half of the MFMA and epilogue work spent on absent channels.
template <int PH, int PWD, int OPT = 3, int CW = 2>
__global__ __launch_bounds__(512, 2) void conv_halo_kernel(const ConvArgs a) {
  constexpr int HPITCH = PWD + 4;
  constexpr int FPR = PWD / 16;       // 16-pixel fragments per patch row
  constexpr int PXW = 8 / CW;         // waves along the pixels
  constexpr int RPW = PH / PXW;       // patch rows per wave
  constexpr int NPF = RPW * FPR;      // pixel fragments per wave (8 or 4)
  constexpr int TNC = CW * 64;        // output channels of a tile
  static_assert(PH * PWD == 512 && (PH + 2) * HPITCH <= 768 && RPW * PXW == PH && (NPF == 8 || NPF == 4), "unsupported patch");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = CW == 2 ? (w >> 2) : 0;  // 64-channel slice
  const int wc = CW == 2 ? (w & 3) : w;   // group of RPW patch rows
  const int fr = lane & 15, fg = lane >> 4;

  // ---- tiles of this work-group (as conv_tile.hip): XCD x owns a contiguous range of the (patch, channel block) list
  const int T = a.tiles_m * a.tiles_n;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, stride = gridDim.x >> 3;
  const int q8 = T >> 3, r8 = T & 7;
  const int xbase = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int xcnt = q8 + (xcd < r8 ? 1 : 0);
  if (idx >= xcnt) return;
  const int my_tiles = (xcnt - idx + stride - 1) / stride;
  const int first_tile = xbase + idx;
  const int nslab = a.C >> 5;
  const int px_cols = (a.Wout + PWD - 1) / PWD, px_rows = (a.Hout + PH - 1) / PH;
  const int patches_per_img = px_cols * px_rows;
  const int total_slabs = my_tiles * nslab;   // slabs this work-group multiplies, across all its tiles
  const int G = total_slabs * 9;              // (slab, tap) steps

  auto tile_origin = [&](int tile, int& img, int& y0, int& x0, int& n0) {
    const int tm = tile / a.tiles_n;
    n0 = (tile - tm * a.tiles_n) * TNC;
    img = tm / patches_per_img;
    const int p = tm - img * patches_per_img;
    const int py = p / px_cols;
    y0 = py * PH;
    x0 = (p - py * px_cols) * PWD;
  };

  // ---- halo staging: piece i of this wave covers halo rows ((i * 8 + w) * 16 ..+15); a lane moves chunk (lane & 3) of its row
  unsigned h_off[HPIECES];  // byte offset of the lane's 16 bytes (slab 0) from a.in; 0xffffffff: outside the image / the halo
  int h_tile = -1, h_slab = nslab - 1;  // the slab most recently staged (or being staged)
  auto setup_halo = [&](int tile) {
    int img, y0, x0, n0;
    tile_origin(tile, img, y0, x0, n0);
    // opaque to the optimiser: otherwise the per-piece (hy, hx, chunk) values are hoisted out of the tile loop as twelve
    // loop-invariant registers, the kernel spills, and one scratch access inside the loop makes the wait-count pass drain
    // the LDS-DMA queue in front of every slab's first fragment read
    int lrow = lane >> 2, lchunk = lane & 3;
    asm volatile("" : "+v"(lrow), "+v"(lchunk));
#pragma unroll
    for (int i = 0; i < HPIECES; ++i) {
      const int R = (i * 8 + w) * 16 + lrow;
      const int hy = R / HPITCH, hx = R - hy * HPITCH;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const int c = lchunk ^ (((hx >> 2) & 1) << 1);
      const bool ok = hy < PH + 2 && hx < PWD + 2 && (unsigned)y < (unsigned)a.Hin && (unsigned)x < (unsigned)a.Win;
      h_off[i] = ok ? (unsigned)((((size_t)(img * a.Hin + y) * a.Win + x) * a.in_ld + c * 8) * 2) : 0xffffffffu;
    }
  };
  // advances the halo cursor to the next slab (possibly the first slab of the next tile)
  auto halo_next_slab = [&]() {
    if (++h_slab == nslab) {
      h_slab = 0;
      ++h_tile;
      setup_halo(first_tile + h_tile * stride);
    }
  };
  auto stage_halo_piece = [&](int i, int buf) {
    const unsigned char* src = h_off[i] != 0xffffffffu ? reinterpret_cast<const unsigned char*>(a.in) + h_off[i] + (size_t)h_slab * 64
                                                       : reinterpret_cast<const unsigned char*>(a.zero);
    glds16(reinterpret_cast<const bf16_t*>(src), smem + buf * HBUF + (i * 8 + w) * 1024);
  };

  // ---- weight staging: one row per thread and slot.  LDS row rho = blk * 16 + q of a wave's 64-channel slice holds channel
  // (blk >> 1) * 32 + (q >> 2) * 8 + (blk & 1) * 4 + (q & 3) (conv_tile.hip: the MFMA D layout then leaves every lane with two
  // runs of 8 consecutive channels).
  const int row_in = lane >> 2;
  const int cc = (lane & 3) ^ swz<32>(row_in);
  unsigned w_base = 0xffffffffu;
  int w_tile = -1, w_slab = nslab - 1, w_tap = 9;  // cursor: the (tile, slab, tap) staged most recently
  auto setup_weights = [&](int tile) {
    const int n0 = (tile % a.tiles_n) * TNC;
    const int R = (CW == 2 ? w : (w & 3)) * 16 + row_in;   // CW = 1: waves 4-7 stage the rows of waves 0-3 again (same bytes)
    const int blk = (R >> 4) & 3, q = R & 15;
    const int n = n0 + (R & ~63) + (blk >> 1) * 32 + (q >> 2) * 8 + (blk & 1) * 4 + (q & 3);
    w_base = n < a.N ? (unsigned)(((size_t)n * (9 * (size_t)a.C) + cc * 8) * 2) : 0xffffffffu;
  };
  auto stage_weights = [&](int slot) {
    if (w_tap == 9) {
      w_tap = 0;
      if (++w_slab == nslab) {
        w_slab = 0;
        ++w_tile;
        setup_weights(first_tile + w_tile * stride);
      }
    }
    const unsigned char* src = w_base != 0xffffffffu
                                   ? reinterpret_cast<const unsigned char*>(a.wt) + w_base + ((size_t)w_tap * a.C + (size_t)w_slab * 32) * 2
                                   : reinterpret_cast<const unsigned char*>(a.zero);
    glds16(reinterpret_cast<const bf16_t*>(src), smem + WBASE + slot * WSLOT + (CW == 2 ? w : (w & 3)) * 1024);
    ++w_tap;
  };

  // ---- fragment addresses.  Weights: as conv_tile.hip.  Pixels: fragment jj (0-7) of the wave is patch row wc * 4 + (jj >> 1),
  // columns (jj & 1) * 16 + fr; tap (dy, dx) reads halo row (patch row + 1 + dy), halo column (column + 1 + dx).
  const int wfrag0 = WBASE + (wr * 64 + fr) * 64 + ((fg ^ swz<32>(fr)) << 4);
  int pbase[3];  // per dx: ((k + fr) * 64 + swizzled chunk) + this wave's four patch rows, in halo buffer 0
#pragma unroll
  for (int k = 0; k < 3; ++k)
    pbase[k] = (wc * RPW * HPITCH + k + fr) * 64 + ((fg ^ ((((k + fr) >> 2) & 1) << 1)) << 4);
  auto ldw = [&](int slot, int t) { return *reinterpret_cast<const s16x8*>(smem + wfrag0 + slot * WSLOT + t * 1024); };

  f32x4 acc[4][NPF];
  s16x8 wfA[2], wfB[2], pf[NPF];

  // BN column statistics: a lane accumulates the sums of ONE channel over all tiles the work-group walks with the same
  // channel block and sends them with two full-wave atomics when the channel block changes or the work-group is done
  float st_s = 0.f, st_ss = 0.f;
  int st_base = -1;  // first channel of the 64-channel slice the lane sums belong to (wave-uniform; -1: none yet)
  auto stats_flush = [&]() {
    const int st_n = st_base + fg * 8 + (fr >> 3) * 32 + (fr & 7);
    if (st_base >= 0 && st_n < a.N && !(a.abl & 16)) {
      asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(a.stats + st_n), "v"(st_s) : "memory");
      asm volatile("global_atomic_add_f32 %0, %1, off\n\ts_nop 1" ::"v"(a.stats + a.N + st_n), "v"(st_ss) : "memory");
    }
    st_s = 0.f; st_ss = 0.f;
  };

  // ---- epilogue (conv_tile.hip's, with patch coordinates): every memory operation is inline asm, see there
  auto epilogue_body = [&](int tile, auto has_bias_c) {
    constexpr bool HAS_BIAS = decltype(has_bias_c)::value;
    int img, y0, x0, n0;
    tile_origin(tile, img, y0, x0, n0);
    const int nslice = n0 + wr * 64;
    const int nb = nslice + fg * 8;
    const bool okA = nb < a.N, okB = nb + 32 < a.N;
    f32x4 bia[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if constexpr (HAS_BIAS) {
      if (okA)
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(bia[0]), "=&v"(bia[1]) : "v"(a.bias + nb) : "memory");
      if (okB)
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(bia[2]), "=&v"(bia[3]) : "v"(a.bias + nb + 32) : "memory");
    }
    f32x2_t s2[8], ss2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s2[e] = f32x2_t{0.f, 0.f}; ss2[e] = f32x2_t{0.f, 0.f}; }
    const bool nt_out = (size_t)a.M * a.out_ld * 2 > ((size_t)160 << 20) && (a.abl & 8);
    const bool do_stats = a.stats && !(a.abl & 2);
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      const int y = y0 + wc * RPW + j / FPR, x = x0 + (j % FPR) * 16 + fr;
      const bool row_ok = y < a.Hout && x < a.Wout;
      uint32_t pk[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
        if constexpr (HAS_BIAS) { v0 += bia[i][0]; v1 += bia[i][1]; v2 += bia[i][2]; v3 += bia[i][3]; }
        if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
        pk[2 * i] = pack_bf16(v0, v1);
        pk[2 * i + 1] = pack_bf16(v2, v3);
      }
      if (do_stats && row_ok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const f32x2_t v = {__uint_as_float(pk[e] << 16), __uint_as_float(pk[e] & 0xffff0000u)};
          s2[e] += v;
          ss2[e] = __builtin_elementwise_fma(v, v, ss2[e]);
        }
      }
      if (row_ok && !(a.abl & 1)) {
        bf16_t* dst = a.out + ((size_t)(img * a.Hout + y) * a.Wout + x) * a.out_ld + nb;
        const u32x4_t va = {pk[0], pk[1], pk[2], pk[3]}, vb = {pk[4], pk[5], pk[6], pk[7]};
        if (nt_out) {
          if (okA) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(va) : "memory");
          if (okB) asm volatile("global_store_dwordx4 %0, %1, off offset:64 nt\n\ts_nop 1" ::"v"(dst), "v"(vb) : "memory");
        } else {
          if (okA) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(va) : "memory");
          if (okB) asm volatile("global_store_dwordx4 %0, %1, off offset:64\n\ts_nop 1" ::"v"(dst), "v"(vb) : "memory");
        }
      }
    }
    if (do_stats) {
      if (nslice != st_base) { stats_flush(); st_base = nslice; }
      float s[16], ss[16];
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[2 * e] = s2[e][0]; s[2 * e + 1] = s2[e][1]; ss[2 * e] = ss2[e][0]; ss[2 * e + 1] = ss2[e][1]; }
      st_s += row16_transpose_sum(s, fr);
      st_ss += row16_transpose_sum(ss, fr);
    }
  };
  auto epilogue = [&](int tile) {
    if (a.abl & 4) return;
    if (a.bias) epilogue_body(tile, std::true_type{});
    else epilogue_body(tile, std::false_type{});
  };

#define U2_H_MFMA(I, WF, J) acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF, pf[J], acc[I][J], 0, 0, 0)

  // pixel fragment jj of tap T (compile time) from the halo buffer the bases currently point into
#define U2_H_LDP(T, JJ)                                                                                                  \
  (*reinterpret_cast<const s16x8*>(smem + pbase[(T) % 3] + (((JJ) / FPR + (T) / 3) * HPITCH * 64 + ((JJ) % FPR) * 1024)))

  // ---- prologue: halo of the first slab, weight slots 0 .. WA-1, everything landed; first fragments of step 0
  halo_next_slab();
#pragma unroll
  for (int i = 0; i < HPIECES; ++i) stage_halo_piece(i, 0);
#pragma unroll
  for (int i = 0; i < WA; ++i) stage_weights(i);
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  wfA[0] = ldw(0, 0); wfA[1] = ldw(0, 1);
#pragma unroll
  for (int j = 0; j < NPF / 2; ++j) pf[j] = U2_H_LDP(0, j);

  // One step = one (slab, tap): 32 MFMA per wave, in conv_tile.hip's two phases
  //   A: 4 MFMA | read pixel fragments 4-7 and the second weight pair of this step; issue the halo piece of the NEXT slab
  //      (taps 0-5) and the weight piece of step g + WA | 12 MFMA
  //   B: counted vmcnt (weights of step g + 1 landed; tap 8: the next slab's halo too), barrier | 8 MFMA | read the first
  //      weight pair and pixel fragments 0-3 of step g + 1 | 8 MFMA
  // vmcnt(T): operations issued after the weight piece of step g + 1 (step g - 5): 5 weight pieces + the halo pieces of the
  // taps t-4 .. t that fall into 0-5; tap 8 instead waits for the last halo piece (tap 5): the 4 weight pieces behind it.
  int g = 0, slot = 0;      // step index, its weight slot (g % WR)
  int slab_g = 0;           // slabs multiplied so far, over all tiles
  bool tail = false;        // the last two slabs of this work-group: staging thins out, waits drain everything
#define U2_H_STEP(T, VMCNT)                                                                                             \
  {                                                                                                                      \
    const int nslot = (slot + 1 == WR) ? 0 : slot + 1;                                                                   \
    const int sslot = (slot == 0) ? WR - 1 : slot - 1; /* slot of step g + WA = g - 1 (mod WR) */                        \
    if constexpr (!(OPT & 1)) __builtin_amdgcn_s_setprio(1);                                                                                       \
    _Pragma("unroll") for (int j = 0; j < NPF / 4; ++j) { U2_H_MFMA(0, wfA[0], j); U2_H_MFMA(1, wfA[1], j); }            \
    if constexpr (!(OPT & 1)) __builtin_amdgcn_s_setprio(0);                                                                                       \
    if constexpr (!(OPT & 2)) __builtin_amdgcn_sched_barrier(0);                                                                                   \
    _Pragma("unroll") for (int j = NPF / 2; j < NPF; ++j) pf[j] = U2_H_LDP(T, j);                                        \
    wfB[0] = ldw(slot, 2); wfB[1] = ldw(slot, 3);                                                                        \
    if ((T) < HPIECES && slab_g + 1 < total_slabs) {                                                                     \
      if ((T) == 0) halo_next_slab();                                                                                    \
      stage_halo_piece((T) < HPIECES ? (T) : 0, (slab_g + 1) & 1);                                                       \
    }                                                                                                                    \
    if (g + WA < G) stage_weights(sslot);                                                                                \
    if constexpr (!(OPT & 2)) __builtin_amdgcn_sched_barrier(0);                                                                                   \
    if constexpr (!(OPT & 1)) __builtin_amdgcn_s_setprio(1);                                                                                       \
    _Pragma("unroll") for (int j = NPF / 4; j < NPF; ++j) { U2_H_MFMA(0, wfA[0], j); U2_H_MFMA(1, wfA[1], j); }          \
    if constexpr (!(OPT & 1)) __builtin_amdgcn_s_setprio(0);                                                                                       \
    if (tail) wait_vm<0>(); else wait_vm<VMCNT>();                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                        \
    asm volatile("" ::: "memory");                                                                                       \
    if constexpr (!(OPT & 1)) __builtin_amdgcn_s_setprio(1);                                                                                       \
    _Pragma("unroll") for (int j = 0; j < NPF / 2; ++j) { U2_H_MFMA(2, wfB[0], j); U2_H_MFMA(3, wfB[1], j); }            \
    if constexpr (!(OPT & 1)) __builtin_amdgcn_s_setprio(0);                                                                                       \
    if constexpr (!(OPT & 2)) __builtin_amdgcn_sched_barrier(0);                                                                                   \
    if ((T) == 8) { /* the next step reads the other halo buffer */                                                      \
      const int hd = (slab_g & 1) ? -HBUF : HBUF;                                                                        \
      pbase[0] += hd; pbase[1] += hd; pbase[2] += hd;                                                                    \
    }                                                                                                                    \
    if (g + 1 < G) {                                                                                                     \
      wfA[0] = ldw(nslot, 0); wfA[1] = ldw(nslot, 1);                                                                    \
      _Pragma("unroll") for (int j = 0; j < NPF / 2; ++j) pf[j] = U2_H_LDP(((T) + 1) % 9, j);                            \
    }                                                                                                                    \
    if constexpr (!(OPT & 2)) __builtin_amdgcn_sched_barrier(0);                                                                                   \
    if constexpr (!(OPT & 1)) __builtin_amdgcn_s_setprio(1);                                                                                       \
    _Pragma("unroll") for (int j = NPF / 2; j < NPF; ++j) { U2_H_MFMA(2, wfB[0], j); U2_H_MFMA(3, wfB[1], j); }          \
    if constexpr (!(OPT & 1)) __builtin_amdgcn_s_setprio(0);                                                                                       \
    ++g;                                                                                                                 \
    slot = nslot;                                                                                                        \
  }

  for (int ti = 0; ti < my_tiles; ++ti) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NPF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nslab; ++s) {
      tail = slab_g + 2 >= total_slabs;
      U2_H_STEP(0, 7)
      U2_H_STEP(1, 7)
      U2_H_STEP(2, 8)
      U2_H_STEP(3, 9)
      U2_H_STEP(4, 10)
      U2_H_STEP(5, 10)
      U2_H_STEP(6, 9)
      U2_H_STEP(7, 8)
      U2_H_STEP(8, 4)
      ++slab_g;
    }
    epilogue(first_tile + ti * stride);
  }
  if (a.stats) wg_flush_column_sums<8>(a.stats, a.N, st_base < 0 ? -1 : st_base + fg * 8 + (fr >> 3) * 32 + (fr & 7), st_s, st_ss, w, lane, smem);
#undef U2_H_STEP
#undef U2_H_LDP
#undef U2_H_MFMA
}

template <int PH, int PWD, int OPT = 3, int CW = 2>
int launch_halo_cfg(ConvArgs& a, int N, int tiny, hipStream_t s) {
  const long long patches = (long long)a.B * ((a.Hout + PH - 1) / PH) * ((a.Wout + PWD - 1) / PWD);
  a.tiles_m = (int)patches;
  a.tiles_n = (N + CW * 64 - 1) / (CW * 64);
  const long long T = patches * a.tiles_n;
  if (T >= (1 << 30)) return 0;
  long long G = T < (tiny ? 8 : 256) ? T : (tiny ? 8 : 256);
  G = (G + 7) & ~7LL;
  static PerDeviceOnce attr_set;
  if (auto once_guard = attr_set.first()) {
    (void)hipFuncSetAttribute((const void*)conv_halo_kernel<PH, PWD, OPT, CW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL((conv_halo_kernel<PH, PWD, OPT, CW>), dim3((unsigned)G), dim3(512), LDS_BYTES, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return -1000 - (int)e;
  return 1;
}

double patch_fill(int H, int W, int ph, int pw) {
  return (double)H * W / ((double)((H + ph - 1) / ph * ph) * ((W + pw - 1) / pw * pw));
}

}  // namespace

// Serves 3x3 / stride 1 / pad 1 launches (C % 32 == 0, N % 8 == 0, not accumulating).  variant bit 24: always where it applies,
// bit 25: never; otherwise automatic, from tests/native/selftest bench2 (profiles/r03_conv_halo.txt):
// the halo kernel wins wherever its patches are reasonably full - the stride-4 maps (200 x 336: 92 % full, +7...31 %) and the
// layers with <= 128 output channels on the stride-8 maps (78-84 % full, +11 %) - and loses on small maps (50 x 84 and below,
// 14 x 14 ROI maps), which stay on the tile kernels.  g_last_conv_kernel code: 300 (128-channel tiles) / 301 (64-channel tiles).
int launch_conv_halo(ConvArgs& a, int N, int C, int variant, hipStream_t s) {
  if ((variant >> 25) & 1) return 0;
  if (((variant >> 12) & 15) != 0) return 0;  // a tile-kernel configuration was asked for explicitly
  const bool forced = (variant >> 24) & 1;
  if (a.remap_out || a.accumulate || a.ntaps != 9 || a.KW != 3 || a.wt_taps != 9 || a.pad_h != 1 || a.pad_w != 1 || a.mul != 1 ||
      a.Hin != a.Hout || a.Win != a.Wout)
    return 0;
  if ((C & 31) || (N & 7) || (a.out_ld & 7) || a.M < 1) return 0;
  if ((unsigned long long)a.B * a.Hin * a.Win * a.in_ld * 2ull >= 0xffffffffull || (unsigned long long)N * 9 * C * 2ull >= 0xfffffff0ull)
    return 0;
  // (an 8 x 64 patch instantiation exists as a template parameter set, but at 256 registers it spills six of them into the
  //  tile loop, and a scratch access there makes the compiler drain the LDS-DMA queue in every slab: not dispatched)
  const double fill = patch_fill(a.Hout, a.Wout, 16, 32);
  if (!forced) {
    const int tnc = N <= 64 ? 64 : TN;
    const long long tiles = (long long)a.B * ((a.Hout + 15) / 16) * ((a.Wout + 31) / 32) * ((N + tnc - 1) / tnc);
    if (tiles < 512) return 0;                          // less than two rounds of tiles: the tile kernels' finer grain wins
    if (!(fill >= 0.88 || (N <= 128 && fill >= 0.70))) return 0;
  }
  const int tiny = (variant >> 16) & 1;
  a.abl = (variant >> 18) & 63;
  if (N <= 64 && !((variant >> 17) & 1)) {  // 64-channel tiles (bit 17: the 128-channel tiles anyway, for A/B runs)
    g_last_conv_kernel = 301;
    return launch_halo_cfg<16, 32, 3, 1>(a, N, tiny, s);
  }
  g_last_conv_kernel = 300;
  switch ((variant >> 29) & 3) {
    case 1: return launch_halo_cfg<16, 32, 2>(a, N, tiny, s);
    case 2: return launch_halo_cfg<16, 32, 1>(a, N, tiny, s);
    case 3: return launch_halo_cfg<16, 32, 0>(a, N, tiny, s);
    default: return launch_halo_cfg<16, 32, 3>(a, N, tiny, s);
  }
}

}  // namespace u2conv
