// Argument block and staging helpers shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_tile.hip).
#pragma once
#include "common.h"

namespace u2conv {

struct ConvArgs {
  const bf16_t* in;
  const bf16_t* wt;
  bf16_t* out;
  const float* bias;
  float* stats;
  const bf16_t* zero;
  int B, Hin, Win, C, in_ld;
  int Hout, Wout, N, out_ld;
  int mul;                 // source pixel = output pixel * mul + tap offset
  int relu, accumulate;
  int M, tiles_m, tiles_n;
  // tap table: tap t reads the source at (qy*mul + tap_dy[t], qx*mul + tap_dx[t]) and uses filter tap tap_w[t]
  int ntaps, wt_taps;      // taps of this launch / taps in the weight layout ([N][wt_taps][C])
  int KW, pad_h, pad_w;    // regular launches (remap_out == 0) derive tap t = (kh, kw) arithmetically, no table reads
  short tap_dy[64], tap_dx[64], tap_w[64];
  int tap_pk[64];          // the same table as one dword per tap (scalar loads): (dy & 0xff) | (dx & 0xff) << 8 | w << 16
  // output placement: pixel (img, qy, qx) of the Hout x Wout grid is written at
  // (img, qy*out_sy + out_y0, qx*out_sx + out_x0) of the Hfull x Wfull map (identity for ordinary launches)
  int remap_out, Hfull, Wfull, out_sy, out_sx, out_y0, out_x0;
  int stagger_by_parity;   // conv_igemm256<true>: wave groups = even / odd waves instead of waves 0-3 / 4-7
  // stream-K launches of conv_tile.hip (work-groups own equal shares of the (tile, half K tile) sequence): one fp32 partial-tile
  // slot and one flag per work-group, device memory owned by the library (per stream)
  float* sk_ws;
  int* sk_flags;
  int abl;                 // conv_halo.hip measurement switches (tests/native/selftest bench2 only; 0 in production):
                           // bit 0 no output stores, bit 1 no BN statistics, bit 2 no epilogue at all, bit 3 `nt` stores for outputs > 160 MB
                           // (rounds 1-2 default; round 3 measured it 29-38 % SLOWER on the write-heavy 1x1 layers, -0.55 ms per step);
                           // bits 4-5 (conv_tile.hip, round 6, TIMING ONLY - wrong results): which waves issue the LDS-DMA staging
};

// Tile shapes: (TM pixels x TN output channels) = 128x128 (default) or 256x64 (layers with <= 64 output channels,
// so no half of the MFMA work is spent on zero-padded channels).  4 waves, each a 64(n) x 64(m) sub-tile.

template <int BK> __device__ __forceinline__ int swz(int row);
template <> __device__ __forceinline__ int swz<64>(int row) { return row & 7; }
template <> __device__ __forceinline__ int swz<32>(int row) { return (-(row >> 2)) & 3; }

__device__ __forceinline__ void glds16(const bf16_t* src, void* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds(U2_GLB_PTR(src), U2_LDS_PTR(lds_dst_wave_base), 16, 0, 0);
}


// Which kernel the most recent u2_conv_igemm / u2_conv_wgrad launch selected (u2_conv_last_kernel, a test / debugging aid):
//   conv_tile_kernel configuration k -> 100 + k;  conv_igemm256_kernel -> 256 (+ 1024 staggered);
//   conv_igemm_kernel<BK, GLDS, TM, TN, NST> -> 1000000 + BK * 10000 + (TM / 64) * 1000 + (TN / 64) * 100 + NST * 10 + GLDS;
//   conv_wgrad_kernel<GLDS, TR> -> 2000 + GLDS * 2 + TR (+ 100 when XCD-grouped);  conv_wgrad256_kernel -> 2256 (+ 100).
extern int g_last_conv_kernel;

// conv_tile.hip: persistent 64(ch) x 128(px)-per-wave tile kernels; returns 1 when it took the launch, 0 when the shape is
// not served (caller falls back to conv_igemm_kernel), -1000 - hipError_t on a launch failure.
int launch_conv_tile(ConvArgs& a, int N, int C, int variant, hipStream_t s);

// conv_halo.hip: 3x3 / stride 1 / pad 1 with the input halo of a 16 x 32 pixel patch staged once per 32-channel slab; same
// return convention as launch_conv_tile.  g_last_conv_kernel code: 300.
int launch_conv_halo(ConvArgs& a, int N, int C, int variant, hipStream_t s);

// conv_stream.hip: 1x1 / stride 1 layers with <= 256 input channels - weights resident in registers, whole pixel rows streamed
// through an LDS ring; same return convention as launch_conv_tile.  g_last_conv_kernel code: 7xx.
int launch_conv_stream(ConvArgs& a, int N, int C, int variant, hipStream_t s);

// wgrad_halo.hip: 3x3 / stride 1 / pad 1 weight gradient, all nine taps per work-group with the input halo in LDS; same
// return convention as launch_conv_tile.  g_last_conv_kernel code: 2900.
int launch_wgrad_halo(const bf16_t* x, const bf16_t* dy, float* dw, long long dw_sn, int dw_st, int dw_sc, int n_valid,
                      int c_valid, const bf16_t* zero, int B, int H, int W, int C, int x_ld, int N, int dy_ld, int rounds,
                      int force, int part_mode, hipStream_t s);

// wgrad_stream.hip: 1x1 / stride 1 weight gradient over large maps - a persistent work-group per pixel range accumulates a whole
// block of dW in registers, operands streamed once through an LDS ring of whole pixel rows; same return convention as
// launch_conv_tile.  g_last_conv_kernel code: 2700 + block configuration.
int launch_wgrad_stream(const bf16_t* x, const bf16_t* dy, float* dw, long long dw_sn, int dw_sc, int n_valid, int c_valid,
                        const bf16_t* zero, int M, int C, int x_ld, int N, int dy_ld, int force, int tiny, int cfg, int per_cu,
                        hipStream_t s);

int launch_wdgrad_stream(const bf16_t* x, const bf16_t* dy, const bf16_t* wt, bf16_t* dx, float* dw, long long dw_sn, int dw_sc,
                         int n_valid, int c_valid, const bf16_t* zero, int M, int C, int x_ld, int N, int dy_ld, int wt_ld,
                         int dx_ld, int force, int tiny, hipStream_t s, const bf16_t* yn = nullptr, const float* k1 = nullptr,
                         const float* k2 = nullptr, const float* k3 = nullptr);

// Library-owned device scratch (conv_igemm.hip), one block per (device, stream, kind): the kernels of a stream run one after
// the other, so consecutive launches share it.  Sized to the largest launch seen so far - allocated on first need, grown on
// demand (the stream is drained before the old block is freed), never above `limit_bytes` (nullptr: the caller takes its
// scratch-free path) - and released by u2_release_scratch().  kind 0: stream-K hand-over slots, 1: stream-K flags (zero on
// allocation; their consumers leave them at zero), 2: weight-gradient partial tiles.
void* scratch_get(int kind, hipStream_t s, size_t need_bytes, size_t limit_bytes, bool zero_on_alloc);

}  // namespace u2conv
