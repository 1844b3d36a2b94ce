// Weight gradient of the 3x3 / stride-1 / pad-1 convolutions with the input halo resident in LDS (gfx950, bf16 in / fp32 out).
//
//   dW[n][tap][c] += sum over output pixels m of dY[m][n] * X[m + tap offset][c]
//
// conv_wgrad_kernel (conv_igemm.hip) gives every (n-tile, c-tile, tap) its own work-group, so a pixel range is streamed
// 9 x tiles times and each 32-pixel step of a wave is only 16 MFMAs between two barriers.  Here a work-group owns a
// 128(n) x 64(c) block of ALL NINE taps: per step it stages the 32 dY pixels once and the three input rows the taps touch
// (34 pixels each), and multiplies them into 9 x 128 x 64 accumulators: 8 waves of 64(n) x 16(c) x 9 taps, 144 accumulator
// registers per lane, two waves per SIMD.  Per step and work-group: 24 KB through the LDS-DMA for 288 MFMAs (the per-tap
// kernel: 16 KB for 64), 26 transposing LDS reads per wave for 36 MFMAs (the per-tap kernel: 16 for 16).
//
// Padded pixel coordinates.  The reduction runs over q = (img * H + y) * (W + 1) + x, x in [0, W]: one extra column per
// image row.  A tap (dy, dx) then reads position q + dy * (W + 1) + dx for EVERY output pixel, and both horizontal borders
// land in the extra column, which is staged as zeros (as are rows -1 / H and the extra column of dY itself): no per-tap
// masks anywhere, all border logic lives in the address computation of the LDS-DMA.  Cost: 1 / (W + 1) idle positions.
//
// Pipeline (four-stage LDS ring, counted vmcnt and lgkmcnt, one raw barrier per step): the taps run continuously across
// steps with every fragment requested four taps before its MFMAs; at tap 2 of step s the wave waits for stage s + 1,
// passes the barrier (which also proves every wave has left stage s - 1) and refills that buffer with stage s + 3.
//
// Replaces the autograd weight gradient of F.conv2d (detectron2/layers/wrappers.py:127-134) for the 3x3 layers of the
// backbone, FPN, RPN, semantic head and mask head.
#include "conv_args.h"

namespace {

struct WgradHaloArgs {
  const bf16_t* x;   // [B][H][W][x_ld] input of the forward conv
  const bf16_t* dy;  // [B][H][W][dy_ld] output gradient
  float* dw;         // dw[n * dw_sn + tap * dw_st + c * dw_sc] += ..., n < n_valid, c < c_valid, tap = kh * 3 + kw
  long long dw_sn;
  int dw_st, dw_sc, n_valid, c_valid;
  const bf16_t* zero;
  int B, H, W, C, x_ld, N, dy_ld;
  int Mp;  // B * H * (W + 1) padded positions
  int tiles_c, tiles, pix_per_wg;
  // Partial blocks instead of atomics (round 6): when `part` is set a work-group stores its 9 x 64 x 128 fp32 block, [tap][c][n]
  // with n fastest, at part + (split * tiles + tile) * PART_BLOCK and wgrad_halo_reduce_kernel sums the splits into dw.  All 256
  // work-groups of a launch reach their 288 KB epilogue at the same moment; as atomics that is 75 MB at the ~1 TB/s fp32
  // atomics retire at (tests/native/atomics_bench) - as long as the whole reduction of a 50 x 84 map - as plain 16-byte stores
  // plus one pass that reads them back it is 2 x 75 MB at 4-5 TB/s.
  float* part;
};
constexpr int PART_BLOCK = 9 * 64 * 128;

constexpr int HS = 32;                  // padded positions per step
constexpr int YB = HS * 256;            // dY image [32 px][128 n], 8 KB
constexpr int RUNB = 40 * 128;          // one input row run [40 px][64 c] (34 used), 5 KB
constexpr int STAGE = YB + 16 * 1024;   // 3 runs = 15 wave-wide loads + 1 idle KB so that every wave issues 3 loads per step
constexpr int RING = 4;
constexpr int LOADS = 3;                // LDS-DMA instructions per wave per step

// 16-byte chunk swizzles.  dY image (256-byte rows): a half-wave of ds_read_b64_tr_b16 touches pixels {P..P+3, P+8..P+11} x
// one 32-byte chunk pair; XOR-ing the chunk index with this makes the eight 32-byte pieces cover all 64 banks once.
__device__ __forceinline__ int y_swz(int pix) { return ((pix & 3) << 1) | (pix & 8); }
// input runs (128-byte rows, two pixels per bank row): pixels of equal parity must get four different chunk pairs, for
// every start P (the taps shift P by 0..2): bits 1 and 3 of the pixel index do that.
__device__ __forceinline__ int x_swz(int p) { return (((p >> 1) & 1) | (((p >> 3) & 1) << 1)) << 1; }

template <int IMM>
__device__ __forceinline__ unsigned long long tr_read(unsigned addr) {
  unsigned long long r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(IMM) : "memory");
  return r;
}

union Frag {
  unsigned long long u[2];
  s16x8 v;
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 512 threads: 8 waves of 64(n) x 16(c) x 9 taps (144 accumulator registers), two waves per SIMD
__global__ __launch_bounds__(512) void conv_wgrad_halo_kernel(const WgradHaloArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

  // work-group -> (pixel split, tile): the tiles of one split stream the same pixels, they sit on one XCD (b % 8)
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int grp = idx / a.tiles;
  const int split = grp * 8 + xcd;
  const int tile = idx - grp * a.tiles;
  const int tile_n = tile / a.tiles_c, tile_c = tile - tile_n * a.tiles_c;
  const int n0 = tile_n * 128, c0 = tile_c * 64;
  const int qbeg = split * a.pix_per_wg;
  const int qend = min(a.Mp, qbeg + a.pix_per_wg);
  if (qbeg >= qend) return;
  const int nsteps = (qend - qbeg + HS - 1) / HS;
  const int Wp = a.W + 1, P = a.H * Wp;

  // ---- staging cursors.  A thread moves the same chunk of the same step-relative pixel in every step, so it keeps the
  // byte offset of its source, the column x (0 .. W, W = the padding column) and - for the input rows - the row, and
  // advances them by 32 positions per step: the real pixel index is q minus the number of rows passed (one padding
  // position per row), so every row wrap takes one pixel pitch off the offset.
  const int nwrap = Wp >= 32 ? 1 : (Wp >= 16 ? 2 : 3);  // row wraps a 32-position step can contain (host: W >= 11)
  // dY: wave w covers pixels w * 4 .. + 3 of the step, 16 chunks (128 n) each
  const long long dy_pitch = (long long)a.dy_ld * 2, x_pitch = (long long)a.x_ld * 2;
  int d_x, d_left;
  long long d_off;
  bool d_on;
  {
    const int pix = w * 4 + (lane >> 4);
    const int cc = (lane & 15) ^ y_swz(pix);
    const int q = qbeg + pix;
    const int gy = q / Wp;
    d_x = q - gy * Wp;
    d_left = qend - q;  // > 0: inside this work-group's range
    d_on = n0 + cc * 8 < a.N;
    d_off = ((long long)gy * a.W + d_x) * dy_pitch + (long long)(n0 + cc * 8) * 2;
  }
  // input: load k = w + 8 * jj (k < 15) is pixels g * 8 .. + 7 (g = k % 5) of the run of row offset dy = k / 5 - 1; run
  // position p holds padded position q - 1 + p of that row
  int x_x[2], x_y[2], x_gy[2];
  long long x_off[2];
  bool x_on[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int k = w + 8 * jj;
    const int g = k % 5, dy = k / 5 - 1;
    const int p = g * 8 + (lane >> 3);
    const int cc = (lane & 7) ^ x_swz(p);
    x_on[jj] = k < 15 && p < 34 && c0 + cc * 8 < a.C;
    const int t = qbeg - 1 + p + Wp;  // >= 0; one row added so that the division never sees a negative number
    const int gy = t / Wp - 1;        // global row img * H + y of the dy = 0 position (-1 for the position before q = 0)
    x_x[jj] = t - (t / Wp) * Wp;
    x_gy[jj] = gy;
    x_y[jj] = (gy + a.H) % a.H;
    x_off[jj] = ((long long)(gy + dy) * a.W + x_x[jj]) * x_pitch + (long long)(c0 + cc * 8) * 2;
  }
  const int rows = a.B * a.H;
  auto issue = [&](int buf) {
    unsigned char* ybase = smem + buf * STAGE;
    unsigned char* xbase = ybase + YB;
    {
      const bool ok = d_on && d_left > 0 && d_x < a.W;
      const bf16_t* src = ok ? reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(a.dy) + d_off) : a.zero;
      u2conv::glds16(src, ybase + w * 1024);
      d_left -= HS;
      d_x += HS;
      d_off += HS * dy_pitch;
      for (int r = 0; r < nwrap; ++r) {
        const bool wrap = d_x >= Wp;
        d_x -= wrap ? Wp : 0;
        d_off -= wrap ? dy_pitch : 0;
      }
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int k = w + 8 * jj;
      const int yy = x_y[jj] + k / 5 - 1;
      const bool ok = x_on[jj] && x_x[jj] < a.W && (unsigned)x_gy[jj] < (unsigned)rows && (unsigned)yy < (unsigned)a.H;
      const bf16_t* src = ok ? reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(a.x) + x_off[jj]) : a.zero;
      u2conv::glds16(src, xbase + k * 1024);
      x_x[jj] += HS;
      x_off[jj] += HS * x_pitch;
      for (int r = 0; r < nwrap; ++r) {
        const bool wrap = x_x[jj] >= Wp;
        x_x[jj] -= wrap ? Wp : 0;
        x_off[jj] -= wrap ? x_pitch : 0;
        x_gy[jj] += wrap ? 1 : 0;
        x_y[jj] += wrap ? 1 : 0;
        x_y[jj] = x_y[jj] == a.H ? 0 : x_y[jj];
      }
    }
  };

  // ---- fragment addresses.  ds_read_b64_tr_b16: lane fr of a 16-lane group supplies the address of 4 contiguous bf16
  // (pixel p0 + fr / 4, channels ch0 + (fr % 4) * 4 .. + 3) and receives channel ch0 + fr of pixels p0 .. p0 + 3; two reads
  // (pixels fg * 8 .. + 3 and + 4 .. + 7) make the 8-pixel K slice of one MFMA operand.
  const int wr = w >> 2, wc = w & 3;  // wave tile: n = wr * 64 .. + 63, c = wc * 16 .. + 15
  const int fr = lane & 15, fg = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)U2_LDS_PTR(smem);
  unsigned yoff[2], xoff[3][2];  // dY block i (16 n): yoff ^ (i << 5) - the block index only touches chunk bits 1-2
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int pix = fg * 8 + h * 4 + (fr >> 2);
    {
      const int ch = wr * 64 + (fr & 3) * 4;
      yoff[h] = pix * 256 + (((ch >> 3) ^ y_swz(pix)) << 4) + (ch & 7) * 2;
    }
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int p = pix + dx;  // run position of tap column dx (offset dx - 1) for output pixel `pix`: pix + 1 + (dx - 1)
      const int ch = wc * 16 + (fr & 3) * 4;
      xoff[dx][h] = lds0 + YB + p * 128 + (((ch >> 3) ^ x_swz(p)) << 4) + (ch & 7) * 2;
    }
  }

  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // a wave whose 64 x 16 sub-tile lies outside the valid N x C block multiplies zeros / padding and writes nothing
  const bool wave_active = (n0 + wr * 64 < a.n_valid) && (c0 + wc * 16 < a.c_valid);

  // Fragment pipeline.  Taps run continuously across steps; the input fragment of a tap is requested FOUR taps before its
  // MFMAs (a transposing read takes several hundred cycles to come back with eight waves reading), the four dY fragments
  // of the next step at taps 2-5 of the current one.  Six input buffers: 9 taps per step and 6 buffers re-align every two
  // steps, so the step body exists in two phases (PH) that also swap the two dY fragment sets - no register copies.  All
  // waits are counted: after tap t only the reads requested at taps t-2, t-1, t may still be outstanding (W below), which
  // leaves tap t+1's fragment complete.
  Frag ya[4], yb[4], xb[6];
#define U2_H_XB(PH, T) xb[((PH) * 3 + (T)) % 6]
#define U2_H_READX(F, T, SB)                                       \
  do {                                                             \
    F.u[0] = tr_read<((T) / 3) * RUNB>((SB) + xoff[(T) % 3][0]);   \
    F.u[1] = tr_read<((T) / 3) * RUNB>((SB) + xoff[(T) % 3][1]);   \
  } while (0)
#define U2_H_READY(F, I, SB)                                       \
  do {                                                             \
    F[I].u[0] = tr_read<0>((SB) + (yoff[0] ^ ((I) << 5)));         \
    F[I].u[1] = tr_read<0>((SB) + (yoff[1] ^ ((I) << 5)));         \
  } while (0)
#define U2_H_MFMA(T, Y, F)                                                                        \
  do {                                                                                            \
    if (wave_active)                                                                              \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                               \
        acc[T][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Y[i].v, F.v, acc[T][i], 0, 0, 0);     \
  } while (0)
#define U2_H_LGKM(CNT, F)                                                                        \
  do {                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(F.u[0]), "+v"(F.u[1]) : "n"(CNT) : "memory");    \
    __builtin_amdgcn_sched_barrier(0);                                                           \
  } while (0)
#define U2_H_TIE_Y(Y)                                                                                              \
  asm volatile(""                                                                                                  \
               : "+v"(Y[0].u[0]), "+v"(Y[0].u[1]), "+v"(Y[1].u[0]), "+v"(Y[1].u[1]), "+v"(Y[2].u[0]), "+v"(Y[2].u[1]), \
                 "+v"(Y[3].u[0]), "+v"(Y[3].u[1])::"memory")
  // reads requested at tap t: the next step's dY fragment t - 2 (taps 2-5), then the input fragment of tap t + 4
  // outstanding-read budget after tap t = requests of taps t-2, t-1, t: {2,2,4,4,4,4,2,2,2} summed over a window of three
#define U2_H_W(T) ((T) == 0 ? 6 : (T) == 1 ? 6 : (T) == 2 ? 8 : (T) == 3 ? 10 : (T) == 4 ? 12 : (T) == 5 ? 12 : (T) == 6 ? 10 : (T) == 7 ? 8 : 6)
#define U2_H_TAP(PH, T, YC, YN)                                                          \
  do {                                                                                  \
    if ((T) >= 2 && (T) <= 5) U2_H_READY(YN, ((T) >= 2 && (T) <= 5) ? (T) - 2 : 0, sn); \
    if ((T) + 4 <= 8) U2_H_READX(U2_H_XB(PH, (T) + 4), ((T) + 4) % 9, sb);              \
    else U2_H_READX(U2_H_XB(PH, (T) + 4), ((T) + 4) % 9, sn);                           \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    U2_H_MFMA(T, YC, U2_H_XB(PH, T));                                                   \
    U2_H_LGKM(U2_H_W(T), U2_H_XB(PH, (T) + 1));                                         \
  } while (0)
  // The last step still requests "the next stage's" fragments (stale ring contents, never multiplied): the counted waits
  // stay exact and the step body has no special cases.
#define U2_H_STEP(PH, YC, YN)                                                                       \
  do {                                                                                              \
    const unsigned sb = (unsigned)(buf * STAGE);                                                    \
    const unsigned sn = (unsigned)(((buf + 1) & (RING - 1)) * STAGE);                               \
    U2_H_TAP(PH, 0, YC, YN);                                                                        \
    U2_H_TAP(PH, 1, YC, YN);                                                                        \
    if (st + 1 < nsteps) {                                                                          \
      /* stage st + 1 landed (stage st + 2 may be in flight); every wave has left stage st - 1 */   \
      if (st + 2 < nsteps) wait_vm<LOADS>();                                                        \
      else wait_vm<0>();                                                                            \
      __builtin_amdgcn_s_barrier();                                                                 \
      asm volatile("" ::: "memory");                                                                \
      if (st + 3 < nsteps) issue((buf + 3) & (RING - 1));                                           \
    }                                                                                               \
    U2_H_TAP(PH, 2, YC, YN);                                                                        \
    U2_H_TAP(PH, 3, YC, YN);                                                                        \
    U2_H_TAP(PH, 4, YC, YN);                                                                        \
    U2_H_TAP(PH, 5, YC, YN);                                                                        \
    U2_H_TAP(PH, 6, YC, YN);                                                                        \
    U2_H_TAP(PH, 7, YC, YN);                                                                        \
    U2_H_TAP(PH, 8, YC, YN);                                                                        \
    U2_H_TIE_Y(YN);                                                                                 \
  } while (0)

  // ---- prologue: stages 0 .. 2 in flight, stage 0 published, dY fragments and taps 0-3 of stage 0 read ----
  static_assert(RING == 4, "the ring indices below are written for four stages");
#pragma unroll
  for (int s0 = 0; s0 < 3; ++s0)
    if (s0 < nsteps) issue(s0);
  if (nsteps >= 3) wait_vm<2 * LOADS>();
  else if (nsteps == 2) wait_vm<LOADS>();
  else wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  {
    U2_H_READY(ya, 0, 0u); U2_H_READY(ya, 1, 0u); U2_H_READY(ya, 2, 0u); U2_H_READY(ya, 3, 0u);
    U2_H_READX(xb[0], 0, 0u); U2_H_READX(xb[1], 1, 0u); U2_H_READX(xb[2], 2, 0u); U2_H_READX(xb[3], 3, 0u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    U2_H_TIE_Y(ya);
    U2_H_LGKM(0, xb[0]); U2_H_LGKM(0, xb[1]); U2_H_LGKM(0, xb[2]); U2_H_LGKM(0, xb[3]);
  }

  int buf = 0, st = 0;
  for (;;) {
    U2_H_STEP(0, ya, yb);
    ++st; buf = (buf + 1) & (RING - 1);
    if (st >= nsteps) break;
    U2_H_STEP(1, yb, ya);
    ++st; buf = (buf + 1) & (RING - 1);
    if (st >= nsteps) break;
  }
#undef U2_H_STEP
#undef U2_H_TAP
#undef U2_H_W
#undef U2_H_TIE_Y
#undef U2_H_LGKM
#undef U2_H_MFMA
#undef U2_H_READY
#undef U2_H_READX
#undef U2_H_XB

  // D[i = n][c]: a lane holds column c = fr, rows n = fg * 4 + r of each 16 x 16 block
  if (!wave_active) return;
  if (a.part) {  // 16 B per lane: rows fg * 4 .. + 3 of column c are consecutive in the [tap][c][n] block
    float* pt = a.part + ((size_t)split * a.tiles + tile) * PART_BLOCK + (wc * 16 + fr) * 128 + wr * 64 + fg * 4;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(pt + t * (64 * 128) + i * 16) = acc[t][i];
    return;
  }
  const int c = c0 + wc * 16 + fr;
  if (c >= a.c_valid) return;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* dst = a.dw + (size_t)t * a.dw_st + (size_t)c * a.dw_sc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wr * 64 + i * 16 + fg * 4 + r;
        if (n < a.n_valid) atomicAdd(dst + n * a.dw_sn, acc[t][i][r]);
      }
    }
}

// Sums the partial blocks of a launch over its pixel splits and adds the result into dw.  A work-group takes 8 columns c of one
// (tile, tap) pair: 32 lanes x 16 B per column and split (coalesced), eight splits in flight per thread; the sums go through
// LDS so that the dw accesses run along c (the contiguous direction of both gradient layouts).  blockIdx.z = a group of
// `per_group` consecutive splits (launches with two tiles and 128 splits); several groups meet in dw through atomics.
__global__ __launch_bounds__(256) void wgrad_halo_reduce_kernel(const float* __restrict__ part, int splits, int per_group,
                                                                const WgradHaloArgs a) {
  __shared__ float sm[8][128 + 1];
  const int tlin = blockIdx.x, slab = blockIdx.y;   // tlin = tile * 9 + tap
  const int tile = tlin / 9, tap = tlin - tile * 9;
  const int tile_n = tile / a.tiles_c, tile_c = tile - tile_n * a.tiles_c;
  const int tid = threadIdx.x;
  const int nq = tid & 31, ci = tid >> 5;
  {
    const float* src = part + (size_t)tlin * (64 * 128) + (slab * 8 + ci) * 128 + nq * 4;
    const size_t sstride = (size_t)a.tiles * PART_BLOCK;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    int sp = blockIdx.z * per_group;
    const int sp_end = min(splits, sp + per_group);
    for (; sp + 7 < sp_end; sp += 8) {
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)(sp + k) * sstride));
      s0 += (v[0] + v[2]) + (v[4] + v[6]);
      s1 += (v[1] + v[3]) + (v[5] + v[7]);
    }
    for (; sp < sp_end; ++sp) s0 += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)sp * sstride));
    s0 += s1;
    float* row = sm[ci] + nq * 4;
    row[0] = s0[0]; row[1] = s0[1]; row[2] = s0[2]; row[3] = s0[3];
  }
  __syncthreads();
  const int c_rel = tid & 7;
  const int c = tile_c * 64 + slab * 8 + c_rel;
  if (c >= a.c_valid) return;
  float* dst = a.dw + (size_t)tap * a.dw_st + (size_t)c * a.dw_sc;
  for (int nl = tid >> 3; nl < 128; nl += 32) {
    const int n = tile_n * 128 + nl;
    if (n < a.n_valid) {
      if (gridDim.z == 1) dst[n * a.dw_sn] += sm[c_rel][nl];
      else atomicAdd(dst + n * a.dw_sn, sm[c_rel][nl]);
    }
  }
}

}  // namespace

namespace u2conv {

// returns 1 when the launch was taken, 0 when the shape is not served, -1000 - hipError_t on a launch failure
int launch_wgrad_halo(const bf16_t* x, const bf16_t* dy, float* dw, long long dw_sn, int dw_st, int dw_sc, int n_valid,
                      int c_valid, const bf16_t* zero, int B, int H, int W, int C, int x_ld, int N, int dy_ld, int rounds,
                      int force, int part_mode, hipStream_t s) {
  if (H < 2 || W < 11 || (C & 7) || (N & 7)) return 0;
  const long long Mp = (long long)B * H * (W + 1);
  if (Mp >= (1LL << 30)) return 0;
  WgradHaloArgs a;
  a.x = x; a.dy = dy; a.dw = dw; a.dw_sn = dw_sn; a.dw_st = dw_st; a.dw_sc = dw_sc; a.n_valid = n_valid; a.c_valid = c_valid;
  a.zero = zero;
  a.B = B; a.H = H; a.W = W; a.C = C; a.x_ld = x_ld; a.N = N; a.dy_ld = dy_ld;
  a.Mp = (int)Mp;
  const int tiles_n = (n_valid + 127) / 128;
  a.tiles_c = (c_valid + 63) / 64;
  a.tiles = tiles_n * a.tiles_c;
  // one work-group per CU and round; pixel splits in multiples of 8 (one per XCD), at least 16 steps each so that the
  // 288 KB atomic epilogue of a work-group stays small beside its reduction
  if (rounds < 1) rounds = 1;
  int nsplit = (256 * rounds + a.tiles - 1) / a.tiles;
  nsplit = (nsplit + 7) & ~7;
  if (nsplit < 8) nsplit = 8;
  const int by_pixels = (int)((Mp / (16 * HS) + 7) / 8 * 8);
  if (nsplit > by_pixels) nsplit = by_pixels < 8 ? 8 : by_pixels;
  // Measured against the per-tap kernel (tests/native/selftest bench2w, profiles/r02_wgrad_halo.txt): +35..50 % on the
  // stride-4 maps, +17 % on 100x168 x 256 ch, +65 % on the 64-channel res2 layers; below ~4000 positions per work-group the
  // 288 KB atomic epilogue that all work-groups reach at the same moment outweighs the faster reduction (-6..-20 %).
  // Round 6: the partial-block epilogue (part_mode 0 = automatic, 1 = atomics only, 2 = partial blocks wherever there is
  // something to reduce) costs ~2 x 15 us per launch where the atomics cost ~75, so the kernel now also takes the mid-size maps
  // from the per-tap kernel (tests/native/selftest bench2w, profiles/r06_wgrad_halo_part.txt; per-tap -> atomics -> partials):
  // 50 x 84 x 256 0.132 -> 0.142 -> 0.112 ms, 100 x 168 x 128 0.130 -> 0.153 -> 0.112, 25 x 42 x 512 0.135 -> 0.137 -> 0.114,
  // the 14 x 14 mask-head maps 0.142 -> 0.145 -> 0.114, 25 x 42 x 256 0.060 -> 0.086 -> 0.054; and where it ran with atomics:
  // 200 x 336 x 64 0.170 -> 0.145, 100 x 168 x 256 0.355 -> 0.335; equal on the 200 x 336 maps with >= 128 channels (33 000
  // positions per work-group), which keep the atomics.
  const long long per_wg = Mp / nsplit;
  const bool use_part = part_mode == 2 ? nsplit >= 2 : (part_mode == 0 && per_wg < 16000 && nsplit >= 2);
  if (!force && per_wg < (use_part ? 384 : 4000)) return 0;
  int ppw = (int)((Mp + nsplit - 1) / nsplit);
  ppw = (ppw + HS - 1) / HS * HS;
  a.pix_per_wg = ppw;
  nsplit = (int)((Mp + ppw - 1) / ppw);   // the ranges that are not empty (a multiple of 8 is only needed for the XCD grouping)
  const int nsplit_grid = (nsplit + 7) & ~7;
  a.part = nullptr;
  if (use_part) {
    a.part = (float*)scratch_get(2, s, (size_t)nsplit * a.tiles * PART_BLOCK * sizeof(float), (size_t)96 << 20, false);
    if (!a.part && !force && per_wg < 4000) return 0;   // no scratch: the per-tap kernel takes the small maps
  }
  static PerDeviceOnce attr_set;
  if (auto once_guard = attr_set.first()) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_halo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL(conv_wgrad_halo_kernel, dim3((unsigned)(nsplit_grid * a.tiles)), dim3(512), RING * STAGE, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return -1000 - (int)e;
  if (a.part) {
    int groups = (512 + a.tiles * 72 - 1) / (a.tiles * 72);   // >= ~512 work-groups in the reduction pass
    if (groups > nsplit / 4) groups = nsplit / 4 > 1 ? nsplit / 4 : 1;
    const int per_group = (nsplit + groups - 1) / groups;
    groups = (nsplit + per_group - 1) / per_group;
    hipLaunchKernelGGL(wgrad_halo_reduce_kernel, dim3((unsigned)(a.tiles * 9), 8, (unsigned)groups), dim3(256), 0, s, a.part, nsplit,
                       per_group, a);
    e = hipGetLastError();
    if (e != hipSuccess) return -1000 - (int)e;
  }
  return 1;
}

}  // namespace u2conv
