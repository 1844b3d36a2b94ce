// Weights-resident streaming kernel for the short-reduction 1x1 convolutions (C <= 256 input channels, stride 1, no padding):
// F.conv2d behind detectron2/layers/wrappers.py:127-134 for conv1 / conv3 / shortcut of the res2-res4 bottlenecks
// (backbone/resnet.py:194-203), the FPN laterals (backbone/fpn.py:141-158) and their data gradients.
//
// Why (round-3 measurements): these launches move 2 (C + N) bytes per pixel for 2 C N flop - 51 to 205 flop/B, left of the 312
// flop/B ridge - so they are HBM-bound by construction, and the tile kernel (conv_tile.hip) ran them at 4.0 TB/s algorithmic:
// its K loop alone (no epilogue) reads at 3.7 TB/s, because a pixel's channel row arrives as 64-byte pieces, one per half K tile
// and each in its own LDS-DMA instruction (profiles/r02_dma_bench.txt: 4.8 TB/s from HBM in 64-byte pieces, 7.0 in >= 128-byte
// pieces), because the same <= 128 KB of weights are staged again for every tile, and because only 32 KB of pixels per CU are in
// flight.  Here
//   * the weights of a wave's 64-channel slice live in REGISTERS for the whole life of the persistent work-group (the MFMA A
//     operand: C / 2 registers per lane), loaded once;
//   * the whole LDS is a ring of pixel tiles [TM pixels][C channels], each pixel row one contiguous run of 64-512 bytes moved
//     by consecutive lanes of one LDS-DMA instruction (16-byte chunks XOR-swizzled on the source side so that every
//     ds_read_b128 fragment read is conflict-free for rows of 64 / 128 / 256 / 512 bytes), 64-160 KB per CU in flight;
//   * one raw barrier per pixel tile, counted vmcnt (DMA pieces only: loads retire in order; the epilogue's stores may retire
//     out of order and only make the wait conservative);
//   * wave tile 64 channels x (16..128) pixels, conv_tile.hip's register-direct epilogue (two 16-byte runs per lane, bias / ReLU,
//     BN column statistics by a transposing DPP reduction, accumulated per lane over the work-group's life and flushed once).
#include <stdlib.h>

#include "common.h"
#include "conv_args.h"

namespace u2conv {
namespace {

template <int CTRL> __device__ __forceinline__ float dpp_mov_s(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// conv_tile.hip: transposing reduction over the 16 lanes of a DPP row; lane fr leaves with the row total of v[fr]
__device__ __forceinline__ float row16_transpose_sum_s(float (&v)[16], int fr) {
  {
    const bool up = fr & 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float keep = up ? v[k + 8] : v[k], send = up ? v[k] : v[k + 8];
      v[k] = keep + dpp_mov_s<0x128>(send);
    }
  }
  {
    const bool up = fr & 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float keep = up ? v[k + 4] : v[k], send = up ? v[k] : v[k + 4];
      v[k] = keep + dpp_mov_s<0x141>(send);
    }
  }
  {
    const bool up = fr & 2;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float keep = up ? v[k + 2] : v[k], send = up ? v[k] : v[k + 2];
      v[k] = keep + dpp_mov_s<0x4E>(send);
    }
  }
  const bool up = fr & 1;
  const float keep = up ? v[1] : v[0], send = up ? v[0] : v[1];
  return keep + dpp_mov_s<0xB1>(send);
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_s;
typedef __attribute__((ext_vector_type(2))) float f32x2_s;
__device__ __forceinline__ uint32_t pack_bf16_s(float lo, float hi) {
  const f32x2_s v = {lo, hi};
  const bf16x2_s r = __builtin_convertvector(v, bf16x2_s);
  return *reinterpret_cast<const uint32_t*>(&r);
}

// XOR applied to the 16-byte chunk index of LDS row `row` (rows of RB bytes): conflict-free ds_read_b128 fragment reads
// (16 rows x 4 chunks per instruction) - checked by brute force over the four lane groups of MI355X_MICROARCH.md's LDS table
template <int RB> __device__ __forceinline__ int row_swz(int row) {
  if constexpr (RB == 64) return (-(row >> 2)) & 3;
  else if constexpr (RB == 128) return (row >> 1) & 7;
  else return row & 15;
}

// NF fragment reads at immediate offsets J * 16 * RB from one address, as inline asm: the compiler's wait-count pass answers
// compiler-visible LDS reads beside an LDS-DMA ring with lgkmcnt(0) even where a counted wait would do (LDS returns in order),
// which exposes the round trip a register double buffer is there to hide
template <int NF, int RB, int J = 0> __device__ __forceinline__ void lds_read_frags(s16x8 (&pf)[NF], unsigned addr) {
  if constexpr (J < NF) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pf[J]) : "v"(addr), "n"(J * 16 * RB) : "memory");
    lds_read_frags<NF, RB, J + 1>(pf, addr);
  }
}
// waits until at most CNT LDS operations are outstanding; the empty statements tie the released registers to the wait, so no
// MFMA that reads them can be scheduled in front of it
template <int CNT, int NF> __device__ __forceinline__ void lds_wait_frags(s16x8 (&pf)[NF]) {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CNT) : "memory");
#pragma unroll
  for (int j = 0; j < NF; ++j) asm volatile("" : "+v"(pf[j])::"memory");
}

template <int N> __device__ __forceinline__ void wait_vm_s() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// KS: 32-channel slabs of the reduction (C = 32 KS); TM: pixels per tile; PSW: waves along the pixels (NWV / PSW along the
// channels, 64 each); NWV: waves per work-group (4: two groups per CU, 8: one); KSS: slabs per ring step - a tile arrives in
// KS / KSS steps of TM x (64 KSS)-byte rows (KSS < KS keeps five steps of a 256-channel tile in flight in an 80 KB ring where
// whole tiles would leave room for two: the read-heavy reducing layers 256 -> 64 / 128)
template <int KS, int TM, int PSW, int NWV, int KSS = KS>
__global__ __launch_bounds__(NWV * 64, 2) void conv_stream_kernel(const ConvArgs a, int ring) {
  constexpr int RB = KSS * 64, CPR = KSS * 4, STAGE = TM * RB;
  constexpr int RPP = 16 / KSS;             // pixel rows per 1 KB LDS-DMA piece
  constexpr int PT = TM * KSS / 16;         // pieces per step
  constexpr int PPW = PT / NWV;             // pieces per wave and step
  constexpr int S = KS / KSS;               // steps per tile
  constexpr int CBW = NWV / PSW, WPX = TM / PSW, NF = WPX / 16;
  static_assert(PT % NWV == 0 && PPW >= 1 && PPW <= 8 && NF >= 1 && (KSS == 1 || KSS == 2 || KSS == 4 || KSS == 8) && KS % KSS == 0 && (S & (S - 1)) == 0, "unsupported configuration");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cbw = w % CBW, psw = w / CBW;
  const int fr = lane & 15, fg = lane >> 4;

  // ---- tiles of this work-group: XCD x owns a contiguous range of pixel tiles; inside the XCD the work-groups are
  // (walker, 256-channel block) pairs, the walkers of one channel block take the range's tiles round robin
  const int TP = a.tiles_m, tiles_n = a.tiles_n;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, gx = gridDim.x >> 3;
  const int q8 = TP >> 3, r8 = TP & 7;
  const int xbase = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int xcnt = q8 + (xcd < r8 ? 1 : 0);
  const int nwalk = gx / tiles_n;
  const int nt = idx % tiles_n, walker = idx / tiles_n;
  if (walker >= nwalk || walker >= xcnt) return;
  const int my_tiles = (xcnt - walker + nwalk - 1) / nwalk;
  const int first_tile = xbase + walker;
  const int n0 = nt * (CBW * 64) + cbw * 64;   // first channel of this wave's slice

  // ---- resident weights: block i (16 rows of the MFMA A operand), slab ks.  Row rho = i * 16 + fr of the slice holds channel
  // (i >> 1) * 32 + (fr >> 2) * 8 + (i & 1) * 4 + (fr & 3) (conv_tile.hip: the D layout then leaves a lane with two runs of 8
  // consecutive channels); lane (fr, fg) holds reduction channels ks * 32 + fg * 8 .. + 7 of that row
  s16x8 wr[KS][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + (i >> 1) * 32 + (fr >> 2) * 8 + (i & 1) * 4 + (fr & 3);
    const bf16_t* src = a.wt + (size_t)(n < a.N ? n : 0) * a.C + fg * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      s16x8 v = *reinterpret_cast<const s16x8*>(src + ks * 32);
      if (n >= a.N) v = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
      wr[ks][i] = v;
    }
  }
  // the loads above must have landed before the first LDS-DMA is issued: a later compiler-generated wait for them would
  // otherwise have to drain the ring
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(wr[ks][i]));

  // ---- LDS-DMA staging: piece q of this wave = piece q * NWV + w of the tile = RPP consecutive pixel rows; lane i moves
  // chunk position (i % CPR) of row (i / CPR), which holds source chunk position ^ swizzle(row)
  unsigned d_off[PPW];   // byte offset of the lane's source chunk inside a tile (first row of the tile = 0)
  int d_row[PPW];
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int row = (q * NWV + w) * RPP + lane / CPR;
    const int pos = lane % CPR;
    const int chunk = (pos & ~15) | ((pos ^ row_swz<RB>(row)) & (CPR < 16 ? CPR - 1 : 15));
    d_row[q] = row;
    d_off[q] = (unsigned)(((size_t)row * a.in_ld + chunk * 8) * 2);
  }
  const unsigned char* in_bytes = reinterpret_cast<const unsigned char*>(a.in);
  auto issue = [&](int g, int slot) {   // step g of this work-group = step g % S of its tile g / S
    const int ti = g / S, st = g % S;
    const int m0 = (first_tile + ti * nwalk) * TM;
    const unsigned char* tbase = in_bytes + (size_t)m0 * a.in_ld * 2 + st * RB;
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const unsigned char* src = (m0 + d_row[q] < a.M) ? tbase + d_off[q] : reinterpret_cast<const unsigned char*>(a.zero);
      glds16(reinterpret_cast<const bf16_t*>(src), smem + slot * STAGE + (q * NWV + w) * 1024);
    }
  };

  // ---- fragment address of pixel fragment 0, slab 0 in slot 0; fragment j adds j * 16 * RB, slab ks XORs (ks * 4) into the
  // chunk index (the swizzle only touches the low four chunk bits and ks * 4 + fg < CPR)
  const unsigned lds0 = (unsigned)(size_t)U2_LDS_PTR(smem);
  const int pfrag0 = (psw * WPX + fr) * RB;
  const int pchunk0 = fg ^ row_swz<RB>(fr);

  f32x4 acc[4][NF];
  // BN column statistics of one channel per lane, over all tiles of the work-group
  float st_s = 0.f, st_ss = 0.f;
  const int st_n = n0 + fg * 8 + (fr >> 3) * 32 + (fr & 7);

  const int nslice = n0;
  const int nb = nslice + fg * 8;
  const bool okA = nb < a.N, okB = nb + 32 < a.N;
  f32x4 bia[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  if (a.bias) {
    if (okA)
      asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(bia[0]), "=&v"(bia[1]) : "v"(a.bias + nb) : "memory");
    if (okB)
      asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(bia[2]), "=&v"(bia[3]) : "v"(a.bias + nb + 32) : "memory");
  }
  const bool do_stats = a.stats && !(a.abl & 2);
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

  // every memory operation of the epilogue is inline asm (conv_tile.hip: a compiler-visible global access inside the tile loop
  // makes the wait-count pass drain the LDS-DMA queue in front of the next fragment read)
  auto epilogue = [&](int ti) {
    const int m0 = (first_tile + ti * nwalk) * TM + psw * WPX;
    f32x2_s s2[8], ss2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s2[e] = f32x2_s{0.f, 0.f}; ss2[e] = f32x2_s{0.f, 0.f}; }
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const int m = m0 + j * 16 + fr;
      const bool row_ok = m < a.M;
      uint32_t pk[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v0 = acc[i][j][0] + bia[i][0], v1 = acc[i][j][1] + bia[i][1];
        float v2 = acc[i][j][2] + bia[i][2], v3 = acc[i][j][3] + bia[i][3];
        if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
        pk[2 * i] = pack_bf16_s(v0, v1);
        pk[2 * i + 1] = pack_bf16_s(v2, v3);
      }
      if (do_stats && row_ok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const f32x2_s v = {__uint_as_float(pk[e] << 16), __uint_as_float(pk[e] & 0xffff0000u)};
          s2[e] += v;
          ss2[e] = __builtin_elementwise_fma(v, v, ss2[e]);
        }
      }
      if (row_ok && !(a.abl & 1)) {
        bf16_t* dst = a.out + (size_t)m * a.out_ld + nb;
        const u32x4_t va = {pk[0], pk[1], pk[2], pk[3]}, vb = {pk[4], pk[5], pk[6], pk[7]};
        if (okA) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(va) : "memory");
        if (okB) asm volatile("global_store_dwordx4 %0, %1, off offset:64\n\ts_nop 1" ::"v"(dst), "v"(vb) : "memory");
      }
    }
    if (do_stats) {
      float s[16], ss[16];
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[2 * e] = s2[e][0]; s[2 * e + 1] = s2[e][1]; ss[2 * e] = ss2[e][0]; ss[2 * e + 1] = ss2[e][1]; }
      st_s += row16_transpose_sum_s(s, fr);
      st_ss += row16_transpose_sum_s(ss, fr);
    }
  };

  // ---- pipeline: ring - 1 steps in flight; step g waits for its own pieces (counted: the pieces of the up to ring - 2 younger
  // steps stay in flight), the barrier publishes the step and frees the slot of step g - 1 for step g + ring - 1
  const int G = my_tiles * S;
  for (int sl = 0; sl < ring - 1 && sl < G; ++sl) issue(sl, sl);
  int slot = 0, g = 0;
  for (int ti = 0; ti < my_tiles; ++ti) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
   for (int st = 0; st < S; ++st, ++g) {
    {
      int nd = G - 1 - g;
      if (nd > ring - 2) nd = ring - 2;
      switch (nd) {
        case 0: wait_vm_s<0>(); break;
        case 1: wait_vm_s<PPW>(); break;
        case 2: wait_vm_s<2 * PPW>(); break;
        case 3: wait_vm_s<3 * PPW>(); break;
        case 4: wait_vm_s<4 * PPW>(); break;
        case 5: wait_vm_s<5 * PPW>(); break;
        default: wait_vm_s<6 * PPW>(); break;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (g + ring - 1 < G) {
      int s2 = slot - 1;
      if (s2 < 0) s2 += ring;
      issue(g + ring - 1, s2);
    }
    const unsigned char* sb = smem + slot * STAGE + pfrag0;
    // the fragments of slab ks + 1 are requested in front of the MFMAs of slab ks (two register sets) where a slab is short:
    // with <= 4 pixel fragments a slab is <= 16 MFMAs per wave and an exposed LDS round trip per slab was most of the step
    constexpr bool DB = NF <= 4 && KSS > 1;
    s16x8 pf[DB ? 2 : 1][NF];
    if constexpr (DB) {
      const unsigned sba = lds0 + (unsigned)(slot * STAGE + pfrag0);
      lds_read_frags<NF, RB>(pf[0], sba + (unsigned)(pchunk0 << 4));
#pragma unroll
      for (int ks = 0; ks < KSS; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < KSS) {
          lds_read_frags<NF, RB>(pf[cur ^ 1], sba + (unsigned)((((ks + 1) * 4) ^ pchunk0) << 4));
          lds_wait_frags<NF, NF>(pf[cur]);
        } else {
          lds_wait_frags<0, NF>(pf[cur]);
        }
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[st * KSS + ks][i], pf[cur][j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < KSS; ++ks) {
#pragma unroll
        for (int j = 0; j < NF; ++j)
          pf[0][j] = *reinterpret_cast<const s16x8*>(sb + j * 16 * RB + (((ks * 4) ^ pchunk0) << 4));
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[st * KSS + ks][i], pf[0][j], acc[i][j], 0, 0, 0);
      }
    }
    slot = slot + 1 == ring ? 0 : slot + 1;
   }
    if (!(a.abl & 4)) epilogue(ti);
  }
  if (do_stats) wg_flush_column_sums<NWV>(a.stats, a.N, st_n, st_s, st_ss, w, lane, smem);
}

template <int KS, int TM, int PSW, int NWV, int KSS = KS>
int launch_stream_cfg(ConvArgs& a, int N, int tiny, int forced, int code, hipStream_t s) {
  constexpr int CBW = NWV / PSW;
  constexpr int STAGE = TM * KSS * 64;
  const int per_cu = NWV == 4 ? 2 : 1;
  const int lds_budget = (per_cu == 2 ? 80 : 160) * 1024;
  int ring = lds_budget / STAGE;
  if (ring > 8) ring = 8;
  if (const char* e = getenv("U2_STREAM_RING")) { const int r = atoi(e); if (r >= 2 && r < ring) ring = r; }  // measurement knob
  if (ring < 2) return 0;
  a.tiles_m = (a.M + TM - 1) / TM;
  a.tiles_n = (N + CBW * 64 - 1) / (CBW * 64);
  long long G = tiny ? 8LL * a.tiles_n : 256LL * per_cu;
  if ((G >> 3) < a.tiles_n) return 0;   // (work-groups of an XCD beyond walkers x channel blocks stay idle)
  // at least ~6 pixel tiles per work-group: shorter-lived launches (stride-32 maps, ROI heads) stay on the tile kernels, whose
  // finer (tile, K) grain fills the chip better there
  if (!forced && !tiny && (long long)a.tiles_m * a.tiles_n < 6 * G) return 0;
  static PerDeviceOnce attr_set;
  if (auto once_guard = attr_set.first()) {
    (void)hipFuncSetAttribute((const void*)conv_stream_kernel<KS, TM, PSW, NWV, KSS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  g_last_conv_kernel = code;
  hipLaunchKernelGGL((conv_stream_kernel<KS, TM, PSW, NWV, KSS>), dim3((unsigned)G), dim3(NWV * 64), (size_t)ring * STAGE, s, a, ring);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return -1000 - (int)e;
  return 1;
}

}  // namespace

// Serves 1x1 / stride 1 / unpadded launches with C in {32, 64, 128, 256}, N % 8 == 0, not accumulating.  variant bit 17:
// always where it applies, bit 26: never; any explicit kernel selection (bits 0-15) keeps the launch on the older kernels.
// g_last_conv_kernel code: 700 + C / 32 * 10 + (1, 2, 4 waves along the pixels -> 0, 1, 2).
int launch_conv_stream(ConvArgs& a, int N, int C, int variant, hipStream_t s) {
  if ((variant >> 26) & 1) return 0;
  if (variant & 0xffff) return 0;
  const bool forced = (variant >> 17) & 1;
  if (a.remap_out || a.accumulate || a.ntaps != 1 || a.wt_taps != 1 || a.pad_h != 0 || a.pad_w != 0 || a.mul != 1 ||
      a.Hin != a.Hout || a.Win != a.Wout)
    return 0;
  if (!(C == 32 || C == 64 || C == 128 || C == 256) || (N & 7) || (a.out_ld & 7) || a.in_ld < C || a.M < 1) return 0;
  if ((unsigned long long)a.M * a.in_ld * 2ull >= 0xffffffffull) return 0;
  const int tiny = (variant >> 16) & 1;
  const int code = 700 + (C / 32) * 10;
  if (C == 64) {
    if (N > 128) return launch_stream_cfg<2, 128, 1, 4>(a, N, tiny, forced, code, s);
    if (N > 64) return launch_stream_cfg<2, 128, 2, 4>(a, N, tiny, forced, code + 1, s);
    return launch_stream_cfg<2, 128, 4, 4>(a, N, tiny, forced, code + 2, s);
  }
  if (C == 256) {
    if (N > 128) return launch_stream_cfg<8, 64, 2, 8>(a, N, tiny, forced, code, s);
    if (N > 64) return launch_stream_cfg<8, 64, 2, 4, 4>(a, N, tiny, forced, code + 1, s);
    return launch_stream_cfg<8, 64, 4, 4, 4>(a, N, tiny, forced, code + 2, s);
  }
  if (C == 128) {
    if (N > 128) return launch_stream_cfg<4, 64, 1, 4>(a, N, tiny, forced, code, s);
    if (N > 64) return launch_stream_cfg<4, 64, 2, 4>(a, N, tiny, forced, code + 1, s);
    return launch_stream_cfg<4, 64, 4, 4>(a, N, tiny, forced, code + 2, s);
  }
  if (N > 128) return launch_stream_cfg<1, 128, 1, 4>(a, N, tiny, forced, code, s);
  if (N > 64) return launch_stream_cfg<1, 128, 2, 4>(a, N, tiny, forced, code + 1, s);
  return launch_stream_cfg<1, 128, 4, 4>(a, N, tiny, forced, code + 2, s);
}

}  // namespace u2conv
