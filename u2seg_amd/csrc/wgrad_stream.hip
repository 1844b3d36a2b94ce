// Streaming weight gradient of the 1x1 / stride-1 convolutions over large maps (gfx950, bf16 in / fp32 out):
//
//   dW[n][c] += sum over pixels m of dY[m][n] * X[m][c]
//
// autograd's weight gradient of F.conv2d behind detectron2/layers/wrappers.py:127-134 for conv1 / conv3 / shortcut of the res2 /
// res3 bottlenecks (backbone/resnet.py:194-203), the FPN laterals (backbone/fpn.py:141-158), the semantic-head 1x1s
// (meta_arch/semantic_seg.py:196-205) and the stem's im2col GEMM.
//
// Why (round-4 profile, profiles/r05_pmc_wgrad_sq.txt): these launches move 2 (N + C) bytes per pixel for 2 N C flop - 51 to 205
// flop/B, HBM-bound by construction - and conv_wgrad_kernel ran them at 2.7-4.4 TB/s algorithmic with its waves parked 56 % of
// the time: a work-group there is a (128 x 128 tile, pixel split) pair that lives for ~65 steps of 32 pixels with ONE step of
// LDS-DMA in flight behind a drained wait, a 64-channel operand fills a quarter of its tile, an operand is streamed once per tile
// of the other one, and 512-1024 work-groups each end in a 64 KB fp32 epilogue.  Here
//   * a persistent work-group owns one contiguous pixel range for its whole life and accumulates a WHOLE (up to 256 x 256)
//     block of dW in registers: every operand byte is read once by one work-group, one fp32 flush per work-group;
//   * the LDS is a ring of 32-pixel steps; a step holds the dY rows and the X rows of its pixels as whole rows of NT / CT
//     channels (64-1024 bytes, consecutive lanes of one LDS-DMA instruction), 16-byte chunks XOR-swizzled on the source side so
//     that the transposing fragment reads (ds_read_b64_tr_b16: the reduction runs over pixels, both operands are pixel-strided)
//     are conflict-free for every row length; 60-128 KB per CU in flight behind counted vmcnt waits, one raw barrier per step;
//   * the MFMA operands of a step are read with 2 (FN + FC) transposing reads per wave for FN x FC MFMAs.
// The flush adds the block into dW with fp32 atomics (other work-groups hold the other pixel ranges of the same block).
#include <stdlib.h>

#include "common.h"
#include "conv_args.h"

namespace u2conv {
namespace {

struct WsArgs {
  const bf16_t* x;    // [M][x_ld]
  const bf16_t* dy;   // [M][dy_ld]
  float* dw;          // dw[n * dw_sn + c * dw_sc] += ..., n < n_valid, c < c_valid
  long long dw_sn;
  int dw_sc, n_valid, c_valid;
  const bf16_t* zero;
  int M, N, C, x_ld, dy_ld;   // N, C: physical channel counts of the rows (multiples of 8)
  int tiles_n, tiles_c, ranges, steps;  // steps = ceil(M / 32); work-group = (pixel range, n-tile, c-tile)
  int ring;
  // fused data gradient (DG instantiations): dx[m][c] = sum_n dY[m][n] W[n][c], from the dY rows the step has in LDS anyway
  const bf16_t* wt;   // [C][wt_ld] bf16, n contiguous (the data-gradient weight layout of a 1x1 filter)
  bf16_t* dx;         // [M][dx_ld]
  int wt_ld, dx_ld;
  int c_cover;        // channels the c-tiles must cover (DG: the physical C, so that padded channels get their zero gradient; else 0)
  // batch-norm apply step in front (AP instantiations): dy[m][n] = k1[n] dy_in[m][n] + k2[n] yn[m][n] + k3[n], never stored
  const bf16_t* yn;   // [M][dy_ld]: the normalisation's input = this layer's forward output
  const float *k1, *k2, *k3;
};

// the expression of norm.hip's apply kernels, spelled the same way (same contraction by the compiler)
__device__ __forceinline__ float ws_apply(float a1, float dz, float a2, float xf, float a3) { return a1 * dz + a2 * xf + a3; }

// XOR applied to the 16-byte chunk index of pixel row `pix` of a step image with rows of RB bytes.  A half-wave of
// ds_read_b64_tr_b16 reads 32 bytes (one chunk pair) of each of the pixels {P .. P+3, P+8 .. P+11}; the eight pieces must cover
// the 64 banks once.  RB >= 256: every row starts a bank row, eight different pairs (wgrad_halo.hip y_swz); RB = 128: two rows
// per bank row, the four rows of equal parity need four different pairs (x_swz); RB = 64: four rows per bank row, rows P + k and
// P + 8 + k need different pairs.
template <int RB> __device__ __forceinline__ int ws_swz(int pix) {
  if constexpr (RB >= 256) return ((pix & 3) << 1) | (pix & 8);
  else if constexpr (RB == 128) return (((pix >> 1) & 1) | (((pix >> 3) & 1) << 1)) << 1;
  else return ((pix >> 3) & 1) << 1;
}

__device__ __forceinline__ unsigned long long ws_tr_read(unsigned addr) {
  unsigned long long r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
  return r;
}
union WsFrag {
  unsigned long long u[2];
  s16x8 v;
};
template <int N> __device__ __forceinline__ void ws_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// The MFMAs of column j (FN of them: every dY fragment against X fragment j) behind a counted wait: the X fragments were
// requested in order after the dY fragments and LDS returns in order, so column j only needs the 2 (FC - 1 - j) youngest reads to
// be outstanding still - the reads of the later columns land while the earlier columns multiply.
// EXTRA: LDS reads issued BEHIND the X fragments that may stay outstanding throughout (the fused data gradient's first fragments).
template <int FN, int FC, int EXTRA = 0, int J = 0>
__device__ __forceinline__ void ws_mfma_columns(f32x4 (&acc)[FN][FC], WsFrag (&yf)[FN], WsFrag (&xf)[FC]) {
  if constexpr (J < FC) {
    static_assert(2 * (FC - 1) + EXTRA <= 15, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (FC - 1 - J) + EXTRA) : "memory");
    asm volatile("" : "+v"(xf[J].u[0]), "+v"(xf[J].u[1])::"memory");   // ties the released registers to the wait
    if constexpr (J == 0) {
#pragma unroll
      for (int i = 0; i < FN; ++i) asm volatile("" : "+v"(yf[i].u[0]), "+v"(yf[i].u[1])::"memory");
    }
#pragma unroll
    for (int i = 0; i < FN; ++i) acc[i][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i].v, xf[J].v, acc[i][J], 0, 0, 0);
    ws_mfma_columns<FN, FC, EXTRA, J + 1>(acc, yf, xf);
  }
}

// WN x WC waves; a wave owns FN x FC blocks of 16 (n) x 16 (c): the work-group's block of dW is NT = WN FN 16 by CT = WC FC 16
// DG (WC = 1, FC = 4 only: the block is all of N by 64 input channels): the same launch also forms the data gradient of its
// pixels and channels.  The dY rows of a step are already in LDS for the weight gradient; as the MFMA B operand of
// dx^T[c][m] = sum_n W^T[c][n] dY[m][n] they are read once more with plain ds_read_b128 (a pixel's 8 consecutive channels n per lane -
// under the image's swizzle these reads are conflict-free as well), against W^T fragments that stay in registers for the whole launch
// (wave w: 16 channels c, all of N: N / 8 registers).  Saves the second HBM pass over dY that a separate data-gradient launch
// makes - dY is the large operand of the expanding 1x1 layers (res2 64 -> 256: 550 of the 825 MB both passes touch).
// AP (with DG): the incoming gradient is not stored anywhere - the layer's output went through a batch normalisation whose
// backward apply step dy = k1[n] dz + k2[n] y + k3[n] (norm.hip: norm_bwd_apply) is evaluated here, in the staged step image: the
// step stages dz and the normalisation's input y (the conv's own output) next to X, every thread rewrites its share of the dz
// image in place, a second barrier publishes it.  One read of dz and y replaces write(dy) + read(dy) of the separate pass.
template <int WN, int WC, int FN, int FC, bool DG = false, bool AP = false>
__global__ __launch_bounds__(WN * WC * 64, (WN * WC == 4 ? 2 : 1)) void wgrad_stream_kernel(const WsArgs a) {
  static_assert(!DG || (WC == 1 && FC == 4 && (WN == 4 || WN == 8)), "fused data gradient: N-major blocks only");
  static_assert(!AP || DG, "the normalisation's apply step rides on the fused form");
  constexpr int NWV = WN * WC, NT = WN * FN * 16, CT = WC * FC * 16;
  constexpr int RBY = NT * 2, RBX = CT * 2;          // row bytes of the two step images
  constexpr int PY = RBY / 32, PX = RBX / 32;        // 1 KB LDS-DMA pieces per step and image (32 rows x RB bytes)
  constexpr int PYS = AP ? 2 * PY : PY;              // AP: a second N-wide image (y) behind the first
  // every wave issues the same number of pieces per step (the counted waits are immediates): a stage is rounded up to a multiple of
  // NWV pieces, the filler pieces read the zero page into the unused tail of the stage
  constexpr int PT = PYS + PX, PPW = (PT + NWV - 1) / NWV, STAGE = PPW * NWV * 1024;
  static_assert(PPW >= 1 && PPW <= 10 && RBY <= 1024 && RBX <= 1024 && RBY >= 64 && RBX >= 64, "unsupported block");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = w / WC, wc = w % WC;
  const int fr = lane & 15, fg = lane >> 4;

  // work-group -> (pixel range, tile): the tiles of one range stream the same pixels and sit on one XCD (b % 8)
  const int tiles = a.tiles_n * a.tiles_c;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int range = (idx / tiles) * 8 + xcd;
  const int tile = idx % tiles;
  const int tile_n = tile / a.tiles_c, tile_c = tile - tile_n * a.tiles_c;
  const int n0 = tile_n * NT, c0 = tile_c * CT;
  const int s_beg = (int)((long long)a.steps * range / a.ranges), s_end = (int)((long long)a.steps * (range + 1) / a.ranges);
  const int G = s_end - s_beg;
  if (G <= 0) return;

  // ---- LDS-DMA staging: piece p of a step (p < PY: dY rows, else X rows) = 1024 / RB consecutive pixel rows; lane i writes chunk
  // position i % (RB / 16) of row i / (RB / 16), which holds source chunk position ^ swizzle(row).  Wave w moves pieces q NWV + w.
  unsigned d_off[PPW];      // byte offset of the lane's source chunk from the operand's first row of the step
  int d_row[PPW];           // pixel row of the step (bit 8: the chunk lies beyond the operand's channels -> zero page)
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int p = q * NWV + w;
    const bool is_y = p < PYS;
    const int rb = is_y ? RBY : RBX, cpr = rb / 16;
    const int piece = is_y ? (p < PY ? p : p - PY) : p - PYS;
    const int row = piece * (1024 / rb) + lane / cpr;
    const int pos = lane % cpr;
    const int chunk = is_y ? (pos ^ ws_swz<RBY>(row)) : (pos ^ ws_swz<RBX>(row));
    const int ch = (is_y ? n0 : c0) + chunk * 8;
    const bool ok = p < PT && ch < (is_y ? a.N : a.C);
    d_row[q] = (row & 255) | (ok ? 0 : 256);
    d_off[q] = (unsigned)(((size_t)row * (is_y ? a.dy_ld : a.x_ld) + ch) * 2);
  }
  const unsigned char* yb = reinterpret_cast<const unsigned char*>(a.dy);
  const unsigned char* y2b = reinterpret_cast<const unsigned char*>(AP ? a.yn : a.dy);
  const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x);
  const unsigned ypitch = (unsigned)a.dy_ld * 64u, xpitch = (unsigned)a.x_ld * 64u;   // bytes per 32-pixel step
  auto issue = [&](int g, int slot) {
    const int st = s_beg + g;
    const int m0 = st * 32;
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int p = q * NWV + w;             // wave-uniform
      const unsigned char* base = p < PY ? yb + (size_t)st * ypitch : p < PYS ? y2b + (size_t)st * ypitch : xb + (size_t)st * xpitch;
      const int row = d_row[q] & 255;
      const unsigned char* src = (!(d_row[q] & 256) && m0 + row < a.M) ? base + d_off[q] : reinterpret_cast<const unsigned char*>(a.zero);
      glds16(reinterpret_cast<const bf16_t*>(src), smem + slot * STAGE + (q * NWV + w) * 1024);
    }
  };

  // ---- fragment addresses inside a stage.  A operand (dY^T): lane (fr, fg) of fragment i needs channel n = (wn FN + i) 16 + fr of
  // pixels fg 8 .. + 7; the transposing read h (h = 0, 1) takes the address of 4 contiguous channels (fr & 3) 4 .. + 3 of pixel
  // fg 8 + h 4 + (fr >> 2) and returns channel fr of pixels fg 8 + h 4 .. + 3.  The 16-channel block i only occupies bits 1-3 of the
  // chunk index, disjoint from the wave's base and from the lane's bit 0: offset(i) = offset(0) ^ (i << 5).
  const unsigned lds0 = (unsigned)(size_t)U2_LDS_PTR(smem);
  unsigned yoff[2], xoff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int pix = fg * 8 + h * 4 + (fr >> 2);
    const int chy = wn * FN * 16 + (fr & 3) * 4, chx = wc * FC * 16 + (fr & 3) * 4;
    yoff[h] = (unsigned)(pix * RBY + (((chy >> 3) ^ ws_swz<RBY>(pix)) << 4) + (chy & 7) * 2);
    xoff[h] = (unsigned)(PYS * 1024 + pix * RBX + (((chx >> 3) ^ ws_swz<RBX>(pix)) << 4) + (chx & 7) * 2);
  }

  f32x4 acc[FN][FC];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FC; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // a wave whose whole sub-block lies outside the valid N x C block (a 28-channel operand in a 64-wide block) only stages
  const bool wave_active = (n0 + wn * FN * 16 < a.n_valid) && (c0 + wc * FC * 16 < a.c_valid);

  // ---- fused data gradient: resident W^T fragments, B-fragment addresses, the step's result held back one step
  constexpr int KSN = DG ? NT / 32 : 1;           // 32-channel steps of the reduction over n
  constexpr int DPX = DG ? (NWV == 4 ? 2 : 1) : 1; // 16-pixel blocks per wave: 4 waves x (1 c-block, 2 px-blocks), 8 x (1, 1)
  constexpr int KG = 4;                           // reduction steps whose fragments are requested together
  s16x8 wreg[KSN];
  unsigned dyoff[DPX];
  unsigned pend[DPX][2];
  long long pend_m[DPX];
  const int dcb = NWV == 4 ? w : (w & 3);
  const int dpb0 = NWV == 4 ? 0 : (w >> 2);
  const int dxc = c0 + dcb * 16 + fg * 4;         // first of the lane's 4 output channels
  // AP: the thread rewrites chunk `acg` (8 channels) of APR rows per step; its 24 coefficients stay in registers
  constexpr int CPR = RBY / 16, APR = AP ? 32 * CPR / (NWV * 64) : 1;
  float ak1[8], ak2[8], ak3[8];
  unsigned aoff[APR];
  if constexpr (AP) {
    static_assert(32 * CPR % (NWV * 64) == 0 && (NWV * 64) % CPR == 0, "rows per thread");
    const int acg = tid % CPR;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int n = n0 + acg * 8 + e;
      const bool ok = n < a.n_valid;
      ak1[e] = ok ? a.k1[n] : 0.f;
      ak2[e] = ok ? a.k2[n] : 0.f;
      ak3[e] = ok ? a.k3[n] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < APR; ++j) {
      const int row = tid / CPR + j * (NWV * 64 / CPR);
      aoff[j] = (unsigned)(row * RBY + ((acg ^ ws_swz<RBY>(row)) << 4));
    }
  }
  if constexpr (DG) {
    const int crow = c0 + dcb * 16 + fr;          // A operand row: channel c
    const bf16_t* wsrc = a.wt + (size_t)(crow < a.C ? crow : 0) * a.wt_ld + fg * 8;
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks) {
      s16x8 v = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (crow < a.C && ks * 32 + fg * 8 < a.wt_ld) v = *reinterpret_cast<const s16x8*>(wsrc + ks * 32);
      wreg[ks] = v;
    }
    // landed before the first LDS-DMA is issued: a later compiler-generated wait for them would have to drain the ring
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks) asm volatile("" : "+v"(wreg[ks]));
    if constexpr (AP) {
#pragma unroll
      for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(ak1[e]), "+v"(ak2[e]), "+v"(ak3[e]));
    }
#pragma unroll
    for (int p = 0; p < DPX; ++p) {
      const int pix = (dpb0 + p) * 16 + fr;
      dyoff[p] = (unsigned)(pix * RBY + ((fg ^ ws_swz<RBY>(pix)) << 4));   // reduction step ks: ^ (ks << 6)
      pend_m[p] = -1;
      pend[p][0] = pend[p][1] = 0u;
    }
  }
  auto store_pending = [&]() {
    if constexpr (DG) {
#pragma unroll
      for (int p = 0; p < DPX; ++p) {
        if (pend_m[p] >= 0 && dxc < a.dx_ld) {
          typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
          const u32x2_t v = {pend[p][0], pend[p][1]};
          bf16_t* dst = a.dx + (size_t)pend_m[p] * a.dx_ld + dxc;
          asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(v) : "memory");
        }
      }
    }
  };

  // ---- pipeline: ring - 1 steps in flight; step g waits for its own pieces (counted), the barrier publishes the step and frees
  // the slot of step g - 1 for step g + ring - 1
  const int ring = a.ring;
  for (int sl = 0; sl < ring - 1 && sl < G; ++sl) issue(sl, sl);
  int slot = 0;
  for (int g = 0; g < G; ++g) {
    int nd = G - 1 - g;
    if (nd > ring - 2) nd = ring - 2;
    switch (nd) {
      case 0: ws_wait_vm<0>(); break;
      case 1: ws_wait_vm<PPW>(); break;
      case 2: ws_wait_vm<2 * PPW>(); break;
      case 3: ws_wait_vm<3 * PPW>(); break;
      case 4: ws_wait_vm<4 * PPW>(); break;
      case 5: ws_wait_vm<5 * PPW>(); break;
      default: ws_wait_vm<6 * PPW>(); break;
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (g + ring - 1 < G) {
      int s2 = slot - 1;
      if (s2 < 0) s2 += ring;
      issue(g + ring - 1, s2);
    }
    // the data gradient of the PREVIOUS step leaves now: its stores have a whole step to retire before the next counted wait
    // (which, counting only the LDS-DMA pieces issued behind it, waits for them as well)
    store_pending();
    const unsigned sb = lds0 + (unsigned)(slot * STAGE);
    if constexpr (AP) {
      // dy = k1 dz + k2 y + k3, evaluated and rounded as norm_bwd_apply does, over the dz image in place
      s16x8 zv[APR], yv[APR];
#pragma unroll
      for (int j = 0; j < APR; ++j) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(zv[j]) : "v"(sb + aoff[j]) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(yv[j]) : "v"(sb + aoff[j] + (unsigned)(PY * 1024)) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < APR; ++j) {
        asm volatile("" : "+v"(zv[j]), "+v"(yv[j]));
        s16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          ov[e] = (short)f2bf(ws_apply(ak1[e], bf2f((unsigned short)zv[j][e]), ak2[e], bf2f((unsigned short)yv[j][e]), ak3[e]));
        asm volatile("ds_write_b128 %0, %1" ::"v"(sb + aoff[j]), "v"(ov) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    // fused data gradient: the B fragments (dY rows as they lie, 8 channels n per lane) of reduction steps k0 .. k0 + KG - 1
    constexpr int NGRP = DG ? KSN / KG : 1, NRD = KG * DPX;
    s16x8 bf[2][KG][DPX];
    auto dg_read = [&](int grp, int buf) {
#pragma unroll
      for (int kk = 0; kk < KG; ++kk)
#pragma unroll
        for (int p = 0; p < DPX; ++p)
          asm volatile("ds_read_b128 %0, %1" : "=v"(bf[buf][kk][p]) : "v"(sb + (dyoff[p] ^ (unsigned)((grp * KG + kk) << 6))) : "memory");
    };
    // (one place for every wave, in front of the transposing reads: LDS returns in order, so the counted waits of the weight
    // gradient's columns cover these reads as well; requesting them BEHIND the X fragments - so that they land during the weight
    // gradient's MFMAs - measured the same, 0.182 vs 0.189 ms on res2 64 -> 256, and needed two code paths)
    if constexpr (DG) dg_read(0, 0);
    if (wave_active) {
      WsFrag yf[FN], xf[FC];
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) yf[i].u[h] = ws_tr_read(sb + (yoff[h] ^ (unsigned)(i << 5)));
#pragma unroll
      for (int j = 0; j < FC; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) xf[j].u[h] = ws_tr_read(sb + (xoff[h] ^ (unsigned)(j << 5)));
      ws_mfma_columns<FN, FC>(acc, yf, xf);
    }
    if constexpr (DG) {
      f32x4 dacc[DPX];
#pragma unroll
      for (int p = 0; p < DPX; ++p) dacc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int grp = 0; grp < NGRP; ++grp) {
        const int cur = grp & 1;
        if (grp + 1 < NGRP) {
          dg_read(grp + 1, cur ^ 1);
          asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NRD) : "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int kk = 0; kk < KG; ++kk)
#pragma unroll
          for (int p = 0; p < DPX; ++p) asm volatile("" : "+v"(bf[cur][kk][p]));
#pragma unroll
        for (int kk = 0; kk < KG; ++kk)
#pragma unroll
          for (int p = 0; p < DPX; ++p)
            dacc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[grp * KG + kk], bf[cur][kk][p], dacc[p], 0, 0, 0);
      }
      // D[c][m]: the lane holds channels dxc .. dxc + 3 of pixel (dpb0 + p) 16 + fr: 8 bytes of that pixel's row
#pragma unroll
      for (int p = 0; p < DPX; ++p) {
        const long long m = (long long)(s_beg + g) * 32 + (dpb0 + p) * 16 + fr;
        pend_m[p] = m < a.M ? m : -1;
        pend[p][0] = (unsigned)f2bf(dacc[p][0]) | ((unsigned)f2bf(dacc[p][1]) << 16);
        pend[p][1] = (unsigned)f2bf(dacc[p][2]) | ((unsigned)f2bf(dacc[p][3]) << 16);
      }
    }
    slot = slot + 1 == ring ? 0 : slot + 1;
  }
  store_pending();

  // ---- flush.  D[i = n][j = c]: lane holds column c = fr, rows n = fg 4 + r of its 16 x 16 blocks
  if (!wave_active) return;
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FC; ++j) {
      const int c = c0 + (wc * FC + j) * 16 + fr;
      if (c >= a.c_valid) continue;
      float* dst = a.dw + (size_t)c * a.dw_sc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (wn * FN + i) * 16 + fg * 4 + r;
        if (n < a.n_valid) atomicAdd(dst + n * a.dw_sn, acc[i][j][r]);
      }
    }
}

template <int WN, int WC, int FN, int FC, bool DG = false, bool AP = false>
int launch_ws(WsArgs& a, int per_cu, int tiny, int code, hipStream_t s) {
  constexpr int NWV = WN * WC, NT = WN * FN * 16, CT = WC * FC * 16;
  constexpr int STAGE = (((AP ? 2 : 1) * NT + CT) / 16 + NWV - 1) / NWV * NWV * 1024;
  a.tiles_n = (a.n_valid + NT - 1) / NT;
  a.tiles_c = ((a.c_cover > a.c_valid ? a.c_cover : a.c_valid) + CT - 1) / CT;
  const int tiles = a.tiles_n * a.tiles_c;
  const int lds_budget = (per_cu == 2 ? 80 : 160) * 1024;
  int ring = lds_budget / STAGE;
  if (ring > 8) ring = 8;
  if (const char* e = getenv("U2_WSTREAM_RING")) { const int r = atoi(e); if (r >= 2 && r < ring) ring = r; }  // measurement knob
  if (ring < 2) return 0;
  a.ring = ring;
  // pixel ranges: one or two work-groups per CU in total, a multiple of 8 ranges (the tiles of a range share an XCD); never
  // fewer than 2 steps per range
  int ranges = tiny ? 8 : (256 * per_cu / tiles) & ~7;
  if (ranges < 8) ranges = 8;
  while (ranges > 8 && a.steps < 2 * ranges) ranges -= 8;
  a.ranges = ranges;
  static PerDeviceOnce attr_set;
  if (auto once_guard = attr_set.first()) {
    (void)hipFuncSetAttribute((const void*)wgrad_stream_kernel<WN, WC, FN, FC, DG, AP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  g_last_conv_kernel = code;
  hipLaunchKernelGGL((wgrad_stream_kernel<WN, WC, FN, FC, DG, AP>), dim3((unsigned)(ranges * tiles)), dim3(NWV * 64), (size_t)ring * STAGE, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return -1000 - (int)e;
  return 1;
}

}  // namespace

// 1x1 / stride 1 / unpadded weight gradients; returns 1 when it took the launch, 0 when the shape is not served, < 0 on a launch
// failure.  `force`: every shape it can serve (tests); otherwise the large maps only.  cfg (tests / A-B runs): 0 = automatic,
// 1 = 256(n) x 64(c), 2 = 64 x 256, 3 = 128 x 128, 4 = 256 x 256 (8 waves).  g_last_conv_kernel code: 2700 + configuration.
int launch_wgrad_stream(const bf16_t* x, const bf16_t* dy, float* dw, long long dw_sn, int dw_sc, int n_valid, int c_valid,
                        const bf16_t* zero, int M, int C, int x_ld, int N, int dy_ld, int force, int tiny, int cfg, int per_cu,
                        hipStream_t s) {
  if ((C & 7) || (N & 7) || (x_ld & 7) || (dy_ld & 7) || M < 1 || n_valid < 1 || c_valid < 1) return 0;
  if ((unsigned long long)M * (unsigned)x_ld * 2ull >= 0xffffffffull || (unsigned long long)M * (unsigned)dy_ld * 2ull >= 0xffffffffull) return 0;
  WsArgs a;
  a.x = x; a.dy = dy; a.dw = dw; a.dw_sn = dw_sn; a.dw_sc = dw_sc; a.n_valid = n_valid; a.c_valid = c_valid; a.zero = zero;
  a.M = M; a.N = N; a.C = C; a.x_ld = x_ld; a.dy_ld = dy_ld;
  a.steps = (M + 31) / 32;
  a.wt = nullptr; a.dx = nullptr; a.wt_ld = a.dx_ld = 0; a.c_cover = 0;
  a.yn = nullptr; a.k1 = a.k2 = a.k3 = nullptr;
  if (!force && M < 200000) return 0;   // >= ~25 steps per work-group: shorter launches stay on the tile kernels
  if (cfg == 0) {
    // the block shape follows the operands: the wider one along its 256, blocks beyond 256 x 256 are tiled (the narrower
    // operand is then re-read from L2 by the tiles of a range)
    // measured (tests/native/selftest bench2w, round 5): 128 x 128 blocks of two work-groups per CU beat the 256 x 256 block of
    // one 8-wave work-group wherever that block would be tiled or half empty (res3 128 <-> 512: 0.084 vs 0.113 ms, 512 -> 256:
    // 0.111 vs 0.129); on the full 256 x 256 layer the two are within 6 %
    if (n_valid <= 64 && c_valid <= 64) cfg = 3;
    else if (c_valid <= 64) cfg = 1;
    else if (n_valid <= 64) cfg = 2;
    else if (n_valid > 128 && n_valid <= 256 && c_valid > 128 && c_valid <= 256) cfg = 4;
    else cfg = 3;
  }
  switch (cfg) {
    case 1: return launch_ws<4, 1, 4, 4>(a, per_cu ? per_cu : 2, tiny, 2701, s);
    case 2: return launch_ws<1, 4, 4, 4>(a, per_cu ? per_cu : 2, tiny, 2702, s);
    case 3: return launch_ws<2, 2, 4, 4>(a, per_cu ? per_cu : 2, tiny, 2703, s);
    case 4: return launch_ws<4, 2, 4, 8>(a, 1, tiny, 2704, s);
    default: return 0;
  }
}

// The same launch with the data gradient fused in (wgrad_stream_kernel<.., DG>): dx[m][c] = sum_n dy[m][n] wt[c][n] for the 1x1 /
// stride-1 layers whose block is all of N (<= 512) by 64 input channels.  Returns 1 when it took the launch, 0 when the shape is
// not served (the caller runs the two separate launches), < 0 on a launch failure.  g_last_conv_kernel code: 2750 + N block / 256.
// yn / k1..k3 (optional): dy is the gradient in front of a batch normalisation's apply step, evaluated on the fly (AP above).
int launch_wdgrad_stream(const bf16_t* x, const bf16_t* dy, const bf16_t* wt, bf16_t* dx, float* dw, long long dw_sn, int dw_sc,
                         int n_valid, int c_valid, const bf16_t* zero, int M, int C, int x_ld, int N, int dy_ld, int wt_ld,
                         int dx_ld, int force, int tiny, hipStream_t s, const bf16_t* yn, const float* k1, const float* k2,
                         const float* k3) {
  if ((C & 7) || (N & 7) || (x_ld & 7) || (dy_ld & 7) || (wt_ld & 7) || (dx_ld & 7) || M < 1 || n_valid < 1 || c_valid < 1) return 0;
  if (dx_ld < C || wt_ld < N) return 0;
  if ((unsigned long long)M * (unsigned)x_ld * 2ull >= 0xffffffffull || (unsigned long long)M * (unsigned)dy_ld * 2ull >= 0xffffffffull) return 0;
  const bool small = N <= 256 && C <= 64, wide = N > 256 && N <= 512 && C <= 128;
  if (!small && !wide) return 0;
  if (!(force & 1) && (M < 200000 || n_valid <= 128)) return 0;
  WsArgs a;
  a.x = x; a.dy = dy; a.dw = dw; a.dw_sn = dw_sn; a.dw_sc = dw_sc; a.n_valid = n_valid; a.c_valid = c_valid; a.zero = zero;
  a.M = M; a.N = N; a.C = C; a.x_ld = x_ld; a.dy_ld = dy_ld;
  a.steps = (M + 31) / 32;
  a.wt = wt; a.dx = dx; a.wt_ld = wt_ld; a.dx_ld = dx_ld;
  a.c_cover = C;   // the c-tiles cover the PHYSICAL input channels: padded channels get their zero data gradient written too
  a.yn = yn; a.k1 = k1; a.k2 = k2; a.k3 = k3;
  if (yn) {
    // with the normalisation's apply step in front: two N-wide images per step, so only the 256-channel block (one 8-wave
    // work-group per CU, 40 KB stages, ring of 4); code 2761
    if (!k1 || !k2 || !k3) return 0;
    // N <= 512 (res3 tails, 128 input channels = two c-tiles that each stage and transform dz and y): 72 KB stages, ring of 2;
    // code 2763.  Only on request (force bit 2): see the measurement in DESIGN.md 5.0 item 8
    if (wide) return (force & 4) ? launch_ws<8, 1, 4, 4, true, true>(a, 1, tiny, 2763, s) : 0;
    if (!small) return 0;
    if (force & 2) return launch_ws<4, 1, 4, 4, true, true>(a, 2, tiny, 2762, s);   // two 4-wave work-groups per CU, ring of 2
    return launch_ws<8, 1, 2, 4, true, true>(a, 1, tiny, 2761, s);
  }
  if (small) return launch_ws<4, 1, 4, 4, true>(a, 2, tiny, 2751, s);
  return launch_ws<8, 1, 4, 4, true>(a, 1, tiny, 2752, s);
}

}  // namespace u2conv
