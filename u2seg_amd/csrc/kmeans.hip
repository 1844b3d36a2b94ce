// Lloyd k-means over fp32 embeddings: assignment (pairwise distance + argmin) and update (segmented sums).
//
// Replaces KMeans in u2seg/Instance_Clustering/shared/utils/nn_utils.py:304-379:
//   E-step  cl = argmin_j sum_d (x_id - c_jd)^2      (:353-355, a pykeops LazyTensor reduction)
//   M-step  c.zero_(); c.scatter_add_(0, cl.repeat(1,D), x); Ncl = bincount(cl); c /= Ncl   (:358-364)
// E-step here: dist_j = |c_j|^2 - 2 x.c_j (the |x|^2 term does not change the argmin) with the dot products on
// the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, an fma chain in k order), so labels agree with the reference
// except where two centroids are equidistant to within fp32 rounding of the two formulations.
// Empty clusters give 0/0 = NaN centroids exactly like the reference (noted at usl-imagenet.py:135).
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "u2seg_hip.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int KM_PTS = 128;    // points per workgroup (32 per wave)
constexpr int KM_BD = 16;      // dims per staged chunk
constexpr int KM_PITCH = 17;   // padded LDS row pitch (floats)
constexpr int KM_TILES = 10;   // 32-centroid tiles per pass (320 centroids)
constexpr int KM_STAGE = (KM_PTS + KM_TILES * 32) * KM_PITCH;  // floats of one staging buffer

__global__ __launch_bounds__(256) void cnorm_kernel(const float* __restrict__ c, float* __restrict__ cn, int D, int K) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= K) return;
  float s = 0.f;
  for (int d = threadIdx.x & 63; d < D; d += 64) { const float v = c[(size_t)j * D + d]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) cn[j] = s;
}

__device__ __forceinline__ bool km_less(float v, int j, float bv, int bj) {
  // torch.argmin order: NaN beats numbers, first index wins ties
  const bool vn = v != v, bn = bv != bv;
  if (vn != bn) return vn;
  if (vn) return j < bj;
  return v < bv || (v == bv && j < bj);
}

// rows / nrows (optional): the kernel labels the points rows[0 .. *nrows) instead of 0 .. N - 1 (the re-check list of the
// screening kernel below; the grid is sized for the worst case and work-groups past the list leave at once)
__global__ __launch_bounds__(256, 1) void kmeans_assign_kernel(const float* __restrict__ x, const float* __restrict__ c,
                                                               const float* __restrict__ cn, long long* __restrict__ labels,
                                                               int N, int D, int K, const int* __restrict__ rows,
                                                               const int* __restrict__ nrows, const unsigned* __restrict__ scal,
                                                               unsigned* __restrict__ state, unsigned limit) {
  __shared__ float stage[2 * KM_STAGE];  // [2][xs: KM_PTS x KM_PITCH | cs: 320 x KM_PITCH]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int p0 = blockIdx.x * KM_PTS;
  // the last launch of a two-level E step also keeps the screening state (see km_state_begin_kernel): a coarse pass that left more than
  // `limit` points undecided switches itself off, every 64th call tries again
  if (state && blockIdx.x == 0 && tid == 0) {
    if (state[1] == 0u) {
      if (scal[2] > limit) { state[1] = 1u; state[2] = 0u; }
    } else if (++state[2] >= 64u) {
      state[1] = 0u; state[2] = 0u;
    }
  }
  if (rows) {
    N = min(N, *nrows);
    if (p0 >= N) return;
  }
  const int li = lane & 31, lk = lane >> 5;

  float best_v[16];
  int best_j[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { best_v[r] = INFINITY; best_j[r] = 0x7fffffff; }

  for (int k0 = 0; k0 < K; k0 += KM_TILES * 32) {
    const int ntile = min(KM_TILES, (K - k0 + 31) / 32);
    f32x16 acc[KM_TILES];
#pragma unroll
    for (int t = 0; t < KM_TILES; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // Staging registers for one 16-dim chunk: 128 points + 320 centroids x 4 float4 per row over 256 threads.  Rows past
    // the end are clamped to the last row instead of predicated (the epilogue ignores them): the loads stay straight-line
    // code, so the compiler can count them (s_waitcnt vmcnt(N)) instead of draining everything at a join.
    float4 xr[2], cr[5];
    const int c4 = tid & 3;  // every float4 index f = q * 256 + tid below has f % 4 == tid % 4
    const float* xrow[2];
    const float* crow[5];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int pi = min(p0 + ((q * 256 + tid) >> 2), N - 1);
      xrow[q] = x + (size_t)(rows ? rows[pi] : pi) * D + c4 * 4;
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) crow[q] = c + (size_t)min(k0 + ((q * 256 + tid) >> 2), K - 1) * D + c4 * 4;
    auto gload = [&](int d0) {
#pragma unroll
      for (int q = 0; q < 2; ++q) xr[q] = *reinterpret_cast<const float4*>(xrow[q] + d0);
#pragma unroll
      for (int q = 0; q < 5; ++q) cr[q] = *reinterpret_cast<const float4*>(crow[q] + d0);
    };
    auto lstore = [&](int buf) {
      float* xs = stage + buf * KM_STAGE;
      float* cs = xs + KM_PTS * KM_PITCH;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int f = q * 256 + tid;
        float* dst = xs + (f >> 2) * KM_PITCH + (f & 3) * 4;
        dst[0] = xr[q].x; dst[1] = xr[q].y; dst[2] = xr[q].z; dst[3] = xr[q].w;
      }
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const int f = q * 256 + tid;
        float* dst = cs + (f >> 2) * KM_PITCH + (f & 3) * 4;
        dst[0] = cr[q].x; dst[1] = cr[q].y; dst[2] = cr[q].z; dst[3] = cr[q].w;
      }
    };

    // One barrier per 16-dim chunk and a global prefetch one full chunk ahead: late in the MFMA sequence of chunk c the
    // registers holding chunk c + 1 (requested at the same point of chunk c - 1) move into the other LDS buffer (its
    // readers finished before the previous barrier) and are immediately re-used to request chunk c + 2.  The LDS operands
    // of step ks + 1 are read while the 10 MFMAs of step ks run; full passes (all 10 centroid tiles present) carry no
    // per-tile branch.
    auto dloop = [&](auto full) {
      constexpr bool FULL = decltype(full)::value;
      gload(0);
      lstore(0);
      if (KM_BD < D) gload(KM_BD);
      __syncthreads();
      int buf = 0;
      for (int d0 = 0; d0 < D; d0 += KM_BD, buf ^= 1) {
        const bool more = d0 + KM_BD < D;
        const float* xa = stage + buf * KM_STAGE + (w * 32 + li) * KM_PITCH + lk;
        const float* cb = stage + buf * KM_STAGE + KM_PTS * KM_PITCH + li * KM_PITCH + lk;
        float a = xa[0], b[KM_TILES];
#pragma unroll
        for (int t = 0; t < KM_TILES; ++t) b[t] = (FULL || t < ntile) ? cb[t * 32 * KM_PITCH] : 0.f;
#pragma unroll
        for (int ks = 0; ks < KM_BD / 2; ++ks) {
          float an = 0.f, bn[KM_TILES];
          if (ks + 1 < KM_BD / 2) {
            an = xa[(ks + 1) * 2];
#pragma unroll
            for (int t = 0; t < KM_TILES; ++t) bn[t] = (FULL || t < ntile) ? cb[t * 32 * KM_PITCH + (ks + 1) * 2] : 0.f;
          }
          if (ks == KM_BD / 2 - 3 && more) {
            lstore(buf ^ 1);
            if (d0 + 2 * KM_BD < D) gload(d0 + 2 * KM_BD);
          }
#pragma unroll
          for (int t = 0; t < KM_TILES; ++t)
            if (FULL || t < ntile) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
          if (ks + 1 < KM_BD / 2) {
            a = an;
#pragma unroll
            for (int t = 0; t < KM_TILES; ++t) b[t] = bn[t];
          }
        }
        __syncthreads();
      }
    };
    if (ntile == KM_TILES) dloop(std::true_type{}); else dloop(std::false_type{});
    // D[i = point][j = centroid]: lane holds column j = li of tile t, rows (r&3) + 8*(r>>2) + 4*lk
#pragma unroll
    for (int t = 0; t < KM_TILES; ++t) {
      if (t < ntile) {
        const int j = k0 + t * 32 + li;
        const float cj = (j < K) ? cn[j] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (j < K) {
            const float v = cj - 2.f * acc[t][r];
            if (km_less(v, j, best_v[r], best_j[r])) { best_v[r] = v; best_j[r] = j; }
          }
        }
      }
    }
    __syncthreads();
  }
  // reduce over the 32 lanes that share lk
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = best_v[r];
    int j = best_j[r];
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      const float ov = __shfl_xor(v, o, 64);
      const int oj = __shfl_xor(j, o, 64);
      if (km_less(ov, oj, v, j)) { v = ov; j = oj; }
    }
    if (li == 0) {
      const int p = p0 + w * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (p < N) labels[rows ? rows[p] : p] = (long long)j;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// E step, fast form: screen with split-bf16 MFMA, re-check the few undecided points exactly.
//
// x and c are split into two bf16 pieces each (v = hi + lo + r, |r| <= 2^-17 |v|) and the dot product is taken as
// hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_bf16 (fp32 accumulation): three MFMAs at the bf16 rate instead of one at
// the fp32 rate (16x the throughput), with |error| <= 3 * 2^-18 * sum_d |x_d c_d| <= 1.2e-5 |x| |c|.  A point whose two
// best distances differ by less than 1e-4 |x| max_j |c_j| (4x the worst case of both errors) is appended to a list and
// labelled again by the exact-fp32 kernel above - so the labels are those of the exact kernel for every point, while
// more than 99 % of the points never see it.
//
// Work-group = 256 points x 320 centroids (8 waves of 32 points; 160 accumulator registers): x goes from HBM straight
// into the MFMA A-operand registers (each element is needed by exactly one wave) and is split there; the split
// centroids (L2-resident, 2 x 320 x D bf16) stream through a three-stage LDS ring by LDS-DMA, 40 KB per 32-dimension step.
// ------------------------------------------------------------------------------------------------------------------
constexpr int KM_MAX_BLOCKS = 4;             // blocks of 320 centroids the screened E step serves (K <= 1280)
constexpr int KS_PTS = 256;
constexpr int KS_NB = 20;                    // 16-centroid blocks: 320 centroids
constexpr int KS_KMAX = KS_NB * 16;
constexpr int KS_STAGE = 2 * KS_KMAX * 64;   // [hi | lo][320 rows][32 bf16]
constexpr int KS_RING = 3;
#ifndef U2_KM_X3
#define U2_KM_X3 1
#endif
constexpr int KS_DMA = KS_STAGE / 1024 / 8;  // LDS-DMA instructions per wave per step (5)

// chl [2][320][D] bf16 (hi plane, lo plane; rows >= K zero), cn[j] = |c_j|^2 in fp32, *cmax2 = max_j cn[j]
typedef __attribute__((ext_vector_type(2))) float ks_hf32x2;
// fp16 (round to nearest even) for the shadow of the first pass: same two bytes and the same MFMA rate as bf16, eleven significant bits
// instead of eight - and the pass's margin is made of the exact norms of what the rounding dropped
typedef _Float16 ks_h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 ks_h16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t ks_pack_h(float lo, float hi) {
  const ks_hf32x2 v = {lo, hi};
  const ks_h16x2 r = __builtin_convertvector(v, ks_h16x2);
  return *reinterpret_cast<const uint32_t*>(&r);
}
__device__ __forceinline__ float ks_h_lo(uint32_t w) { const ks_h16x2 r = *reinterpret_cast<const ks_h16x2*>(&w); return (float)r[0]; }
__device__ __forceinline__ float ks_h_hi(uint32_t w) { const ks_h16x2 r = *reinterpret_cast<const ks_h16x2*>(&w); return (float)r[1]; }
__device__ __forceinline__ unsigned short ks_f2h(float v) { const _Float16 h = (_Float16)v; return *reinterpret_cast<const unsigned short*>(&h); }
__device__ __forceinline__ float ks_h2f(unsigned short b) { return (float)*reinterpret_cast<const _Float16*>(&b); }
// mu != nullptr (the first pass over the shadow works on x - mu, c - mu; see km_shadow_kernel): chc [320][D] = fp16(S (c - mu)) in the same
// dimension order (S = mu[-...]: the shadow's power-of-two scale, aux[0] behind mu), cnc[j] = |c_j - mu|^2, cs[0] = max_j of it,
// cs[1] = max_j |(c_j - mu) - chc_j / S|^2
__global__ __launch_bounds__(256) void csplit_kernel(const float* __restrict__ c, bf16_t* __restrict__ chl, float* __restrict__ cn,
                                                     unsigned* __restrict__ cmax2, int D, int K, const float* __restrict__ mu,
                                                     bf16_t* __restrict__ chc, float* __restrict__ cnc, unsigned* __restrict__ cs,
                                                     const float* __restrict__ aux) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= KS_KMAX) return;
  float s = 0.f, slo = 0.f, sc = 0.f, sclo = 0.f;
  const float S = mu ? aux[0] : 1.f, Sinv = mu ? aux[1] : 1.f;
  // Within every 32-dimension step the dimensions are stored in the order the screening kernel's x loads deliver them: position
  // fg * 8 + e holds dimension fg * 4 + e (e < 4) or 16 + fg * 4 + (e - 4) - a lane of the MFMA A operand then gets its eight
  // values from two 16-byte loads that are 64 bytes apart, and the four lanes of a row read 64 contiguous bytes per instruction.
  // loads of CH rows of 64 dimensions first, then the arithmetic and the stores (with the stores between them the compiler kept one row of
  // loads in flight: 18.5 us for a 0.9 MB job, every E step)
  constexpr int CH = 12;
  for (int d0 = threadIdx.x & 63; d0 < D; d0 += 64 * CH) {
    float vv[CH], uu[CH], mm[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int d = d0 + i * 64;
      const int pos = d & 31, fgp = pos >> 3, e = pos & 7;
      const int src = (d & ~31) + ((e < 4) ? fgp * 4 + e : 16 + fgp * 4 + (e - 4));
      const bool ok = d < D && j < K;
      vv[i] = ok ? c[(size_t)j * D + src] : 0.f;
      uu[i] = ok ? c[(size_t)j * D + d] : 0.f;
      mm[i] = (ok && mu) ? mu[src] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int d = d0 + i * 64;
      if (d >= D) break;
      const float v = vv[i];
      const bf16_t h = f2bf(v);
      chl[(size_t)j * D + d] = h;
      chl[((size_t)KS_KMAX + j) * D + d] = f2bf(v - bf2f(h));
      slo += (v - bf2f(h)) * (v - bf2f(h));   // |c_j - bf16(c_j)|^2 (the permutation does not change the sum's terms)
      if (mu) {
        const float vc = j < K ? v - mm[i] : 0.f;
        const unsigned short hc = ks_f2h(vc * S);
        chc[(size_t)j * D + d] = hc;
        sc += vc * vc;
        const float rc = vc - ks_h2f(hc) * Sinv;
        sclo += rc * rc;
      }
      // |c_j|^2 is summed over the UNPERMUTED dimensions, lane by lane exactly as cnorm_kernel does: the re-check must see the
      // same bits as a run of the exact kernel alone, or near-duplicate centroids (exact-fp32 ties) are decided differently
      s += uu[i] * uu[i];
    }
  }
  if (j >= K) return;
  s = wave_sum(s);
  slo = wave_sum(slo);
  if ((threadIdx.x & 63) == 0) {
    cn[j] = s;
    atomicMax(cmax2, __float_as_uint(s));  // s >= 0: the unsigned order of the bits is the order of the floats
    atomicMax(cmax2 + 3, __float_as_uint(slo));   // max_j |c_j - bf16(c_j)|^2
  }
  if (mu) {
    sc = wave_sum(sc);
    sclo = wave_sum(sclo);
    if ((threadIdx.x & 63) == 0) {
      cnc[j] = sc;
      atomicMax(cs, __float_as_uint(sc));
      atomicMax(cs + 1, __float_as_uint(sclo));
    }
  }
}

typedef __attribute__((ext_vector_type(2))) __bf16 ks_bf16x2;
typedef __attribute__((ext_vector_type(2))) float ks_f32x2;
__device__ __forceinline__ uint32_t ks_pack(float lo, float hi) {  // v_cvt_pk_bf16_f32, round to nearest even
  const ks_f32x2 v = {lo, hi};
  const ks_bf16x2 r = __builtin_convertvector(v, ks_bf16x2);
  return *reinterpret_cast<const uint32_t*>(&r);
}
__device__ __forceinline__ int ks_swz(int row) { return (-(row >> 2)) & 3; }  // 64-byte rows: conflict-free ds_read_b128
// v_min_f32 / v_max_f32 as they are: fminf / fmaxf on a value that went through integer instructions get a canonicalising v_max_f32 v, v, v
// in front (a fifth of the arg-min's instructions); a signalling NaN cannot come out of an fma, and every NaN case goes by the margin
__device__ __forceinline__ float kc_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float kc_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// ---- the 16-bit shadow of x (round 6) -----------------------------------------------------------------------------------------
// x does not change between Lloyd iterations, and the coarse pass only ever uses a 16-bit rounding of it (bf16(x) at first, fp16 of the
// translated and scaled x now - see km_shadow_kernel): u2_kmeans_prepare writes that once, in the
// order the coarse pass consumes it, and every later E step streams 2 instead of 4 bytes per element (the values are the ones the
// kernel packed on the fly before - same products, same labels).  Layout: [D / 32 steps][G groups of 16 points][64 lanes][8 bf16],
// G = 16 * ceil(N / 256); the 16 bytes of lane (fg, fr) = fg * 16 + fr are point fr's positions fg * 8 .. + 7 of the step in
// csplit_kernel's dimension order - a group is 1 KB, one LDS-DMA instruction, and lands in LDS as the MFMA A fragment of 16 points
// (lane L reads byte L * 16: conflict-free by construction).  Points >= N are zero.  Behind it: |x_p| in fp32, [G * 16].
// Column mean of x in a fixed order (deterministic: the shadow of a given x is the same in every run): partial sums of 1024-row slabs,
// then one thread per column over the slabs.
constexpr int KM_MU_ROWS = 1024;
__global__ __launch_bounds__(256) void km_colsum_kernel(const float* __restrict__ x, float* __restrict__ part, int N, int D) {
  const int r0 = blockIdx.x * KM_MU_ROWS, r1 = min(N, r0 + KM_MU_ROWS);
  for (int d = threadIdx.x; d < D; d += 256) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
      s0 += x[(size_t)r * D + d]; s1 += x[(size_t)(r + 1) * D + d]; s2 += x[(size_t)(r + 2) * D + d]; s3 += x[(size_t)(r + 3) * D + d];
    }
    for (; r < r1; ++r) s0 += x[(size_t)r * D + d];
    part[(size_t)blockIdx.x * D + d] = (s0 + s1) + (s2 + s3);
  }
}
__global__ __launch_bounds__(256) void km_mu_kernel(const float* __restrict__ part, int nslabs, float* __restrict__ mu, int N, int D) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  float s = 0.f;
  for (int b = 0; b < nslabs; ++b) s += part[(size_t)b * D + d];
  s /= (float)N;
  mu[d] = (s == s && fabsf(s) != INFINITY) ? s : 0.f;    // a NaN / Inf somewhere in the column: no translation of that column
}
// The shadow holds a rounding of x - mu: argmin_j |x - c_j|^2 does not change when x and every c_j are translated by the same vector, and the
// first pass's margin is proportional to |x - mu| |c - mu| instead of |x| |c| - on L2-normalised features with a common direction
// (F.normalize(DINO features): usl-imagenet.py:103; mean cosine between rows 0.3-0.8) that is what lets it decide anything at all.
// Any mu is valid as long as x and c use the same one; mu = the column mean of x, fixed when the shadow is made.
// Stored as fp16(S (x - mu)), S a power of two that puts the largest |x_p - mu| at 2^13 .. 2^14 (km_scale_kernel): the scaling is exact,
// nothing overflows, and elements down to 2^-28 of the largest norm keep their eleven bits.
__global__ __launch_bounds__(256) void km_shadow_kernel(const float* __restrict__ x, const float* __restrict__ mu, uint4* __restrict__ xh,
                                                        int N, int D, int G, const float* __restrict__ aux) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = (int)(t & 63), fr = lane & 15, fg = lane >> 4;
  const size_t gi = t >> 6;                     // step * G + group
  const int step = (int)(gi / (size_t)G), g = (int)(gi % (size_t)G);
  if (step >= (D >> 5)) return;
  const int p = g * 16 + fr;
  float4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
  if (p < N) {
    const float* src = x + (size_t)p * D + step * 32 + fg * 4;
    const float* ms = mu + step * 32 + fg * 4;
    a = *reinterpret_cast<const float4*>(src);
    b = *reinterpret_cast<const float4*>(src + 16);
    const float4 ma = *reinterpret_cast<const float4*>(ms), mb = *reinterpret_cast<const float4*>(ms + 16);
    const float S = aux[0];
    a.x = (a.x - ma.x) * S; a.y = (a.y - ma.y) * S; a.z = (a.z - ma.z) * S; a.w = (a.w - ma.w) * S;
    b.x = (b.x - mb.x) * S; b.y = (b.y - mb.y) * S; b.z = (b.z - mb.z) * S; b.w = (b.w - mb.w) * S;
  }
  uint4 o;
  o.x = ks_pack_h(a.x, a.y); o.y = ks_pack_h(a.z, a.w); o.z = ks_pack_h(b.x, b.y); o.w = ks_pack_h(b.z, b.w);
  xh[t] = o;
}
// with y = x_p - mu.  Phase 0: xn[p] = |y|, xn[2 NP16 + p] = |x_p|, *maxn2 = max_p |y|^2 (for the scale).  Phase 1 (S known):
// xn[NP16 + p] = |y - fp16(S y) / S| - what the first pass does not see of the point, exactly.
__global__ __launch_bounds__(256) void km_xnorm_kernel(const float* __restrict__ x, const float* __restrict__ mu, float* __restrict__ xn,
                                                       int N, int D, int NP16, int phase, float* __restrict__ aux) {
  const float S = phase ? aux[0] : 1.f, Sinv = phase ? aux[1] : 1.f;
  float wmax = 0.f;       // largest finite |y|^2 this wave has seen (one atomic per wave: a million of them on one address took 6 ms)
  for (int p = blockIdx.x * 4 + (threadIdx.x >> 6); p < NP16; p += gridDim.x * 4) {
  float s = 0.f, sl = 0.f, so = 0.f;
  if (p < N)
    for (int d = (threadIdx.x & 63) * 4; d < D; d += 256) {
      float4 v = *reinterpret_cast<const float4*>(x + (size_t)p * D + d);
      const float4 m = *reinterpret_cast<const float4*>(mu + d);
      so += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      v.x -= m.x; v.y -= m.y; v.z -= m.z; v.w -= m.w;
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      if (phase) {
        const uint32_t h0 = ks_pack_h(v.x * S, v.y * S), h1 = ks_pack_h(v.z * S, v.w * S);
        const float r0 = v.x - ks_h_lo(h0) * Sinv, r1 = v.y - ks_h_hi(h0) * Sinv;
        const float r2 = v.z - ks_h_lo(h1) * Sinv, r3 = v.w - ks_h_hi(h1) * Sinv;
        sl += r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3;
      }
    }
  s = wave_sum(s);
  sl = wave_sum(sl);
  so = wave_sum(so);
  if ((threadIdx.x & 63) == 0) {
    if (phase) {
      xn[NP16 + p] = sqrtf(sl);
    } else {
      xn[p] = sqrtf(s);
      xn[2 * (size_t)NP16 + p] = sqrtf(so);
      if (p < N && s < 3.0e38f) wmax = fmaxf(wmax, s);   // rows with a NaN / Inf do not set the scale (no pass decides them anyway)
    }
  }
  }
  if (!phase && (threadIdx.x & 63) == 0)   // >= 0: the unsigned order of the bits is the order of the floats
    atomicMax(reinterpret_cast<unsigned*>(aux + 2), __float_as_uint(wmax));
}
// aux[0] = S, aux[1] = 1 / S from aux[2] = the largest finite |x_p - mu|^2: that norm lands in [2^13, 2^14); S = 1 when there is none
__global__ void km_scale_kernel(float* __restrict__ aux) {
  if (threadIdx.x != 0) return;
  const float m = sqrtf(aux[2]);
  float S = 1.f;
  if (m > 0.f && m < 3.0e38f) {
    int e;
    (void)frexpf(m, &e);                         // m = f 2^e, 0.5 <= f < 1
    S = ldexpf(1.f, min(max(14 - e, -100), 100));
  }
  aux[0] = S;
  aux[1] = 1.f / S;
}

// NP = 3: the three products hi.hi + hi.lo + lo.hi (error <= 1.2e-5 |x| |c|, margin_rel 1e-4).  NP = 1 (round 4): hi.hi only - a
// third of the matrix work, error <= 2^-8 |x| |c| per product (both operands rounded to bf16), i.e. 2^-6 |x| max|c| between two
// distances; with margin_rel 0.02 (1.28 x that worst case) it decides every point of well-separated data (the mixture: own
// centroid vs the next differ by ~|c - c'|^2, two orders above the margin) and hands the rest to the NP = 3 pass.
// rows / nrows: the kernel labels the points rows[0 .. *nrows) (the undecided list of the coarser pass) instead of 0 .. N - 1.
// gate / gate_want: the launch is skipped (work-groups leave at once) unless (*gate == 1) == gate_want - the host queues the
// coarse pass and the fine pass, and a flag in the workspace says whether the coarse one runs; gate_want = 2: the fine pass takes
// rows / nrows while the flag is 0 and every point while it is 1.
template <int NP, bool XR3 = false>
__global__ __launch_bounds__(512) void kmeans_screen_kernel(const float* __restrict__ x, const bf16_t* __restrict__ chl,
                                                            const float* __restrict__ cn, const unsigned* __restrict__ cmax2,
                                                            long long* __restrict__ labels, int* __restrict__ list,
                                                            int* __restrict__ nlist, int N, int D, int K, float margin_rel,
                                                            const int* __restrict__ rows, const int* __restrict__ nrows,
                                                            const int* __restrict__ gate, int gate_want) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ks_smem[];
  if (gate) {
    if (gate_want == 2) {                      // one launch for both cases: coarse pass off -> every point, on -> its list
      if (*gate == 1) rows = nullptr;
    } else if ((*gate == 1) != (gate_want != 0)) {
      return;
    }
  }
  if (rows) {
    N = min(N, *nrows);
    if ((int)blockIdx.x * KS_PTS >= N) return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int p0 = blockIdx.x * KS_PTS + w * 32;
  const int nsteps = D >> 5;
  // Round 5, XR: the x rows come through a wave-private LDS-DMA ring of two 4 KB slots instead of registers: a slot is re-filled
  // with x(s + 2) as soon as the wave has read x(s) out of it, so two steps of x (64 KB per CU) are in flight instead of one - the
  // coarse pass streamed x at 2.9 TB/s with 32 KB per CU in flight, which is what a ~2.5 us round trip allows - and nothing that is
  // in flight lives in a register the compiler could move.  NP = 1 stages only the hi plane of the centroids (3 instead of 5
  // LDS-DMA instructions per wave and step: rows 320-383 of the stage are filler from the lo plane), ring of 3 as before; NP = 3
  // with XR keeps both planes in a ring of 2 (centroids(s + 1) requested at the start of step s, 144 KB in all).
  // Round 6, X3 (coarse pass only, -DU2_KM_X3=0 restores the round-5 form): THREE x slots.  The coarse pass is bound by how many bytes of x
  // a CU has in flight (3.8 TB/s with two steps = 64 KB); a third slot needs (i) the centroid stage without its filler rows - 320 rows
  // x 64 B = 20 KB, twenty LDS-DMA instructions: waves 0-5 issue three, wave 6 two, wave 7 none - so that 3 x 20 + 3 x 32 KB = 156 KB
  // fit, and (ii) the centroids of a step requested BEFORE the step's x: vmcnt retires in order, so with x(s + 2) queued in front of
  // centroids(s + 1) the wait for the centroids waited for that x as well and a third slot bought nothing (round 5's note).  Issue
  // order now: ... c(s + 1) x(s + 2) | c(s + 2) x(s + 3) | ...; the wait that closes step s leaves x(s + 2), c(s + 2), x(s + 3) in flight.
  constexpr bool XR = NP == 1 || XR3;
  constexpr bool X3 = NP == 1 && U2_KM_X3;
  constexpr int CDMA = NP == 1 ? 3 : KS_DMA;           // centroid LDS-DMA instructions per wave and step (X3: at most)
  constexpr int CSTAGE = X3 ? KS_KMAX * 64 : CDMA * 8 * 1024;   // bytes of a centroid stage
  constexpr int CRING = (NP == 3 && XR) ? 2 : KS_RING; // centroid stages
  constexpr int XDMA = 4;                              // x: 32 rows x 128 bytes per wave and step
  constexpr int XSLOT = KS_PTS * 128;                  // bytes of an x slot of the work-group
  constexpr int XSLOTS = X3 ? 3 : 2;
  unsigned char* const xring = ks_smem + CRING * CSTAGE;

  // x: lane (fr, fg) owns row m * 16 + fr of both 16-point blocks and, per 32-dimension step, dimensions fg * 4 .. + 3 and
  // 16 + fg * 4 .. + 3 (csplit_kernel stores the centroids' dimensions in the matching order): the four lanes of a row read 64
  // contiguous bytes per load instruction, every 64-byte sector is requested once (with fg * 8 .. + 7 per lane each sector was
  // requested by both instructions)
  const float* xp[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int pi = min(p0 + m * 16 + fr, N - 1);
    xp[m] = x + (size_t)(rows ? rows[pi] : pi) * D + fg * 4;
  }
  // XR: instruction q of the wave moves rows q * 8 .. + 7 of its 32 points, 8 lanes per 128-byte row; LDS position (lane & 7)
  // of row r holds the row's 16-byte chunk (lane & 7) ^ ((r >> 1) & 7): rows of equal parity share a bank half (128-byte pitch),
  // and the fragment reads below - 16 rows per quarter wave, one chunk each - then hit eight different positions per half
  const float* xq[XDMA];
  unsigned xrd[2];
  if constexpr (XR) {
#pragma unroll
    for (int q = 0; q < XDMA; ++q) {
      const int r = q * 8 + (lane >> 3);
      const int pi = min(p0 + r, N - 1);
      xq[q] = x + (size_t)(rows ? rows[pi] : pi) * D + (((lane & 7) ^ ((r >> 1) & 7)) << 2);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int r = m * 16 + fr;
      xrd[m] = (unsigned)(w * 4096 + r * 128 + ((fg ^ ((r >> 1) & 7)) << 4));   // chunk fg; chunk 4 + fg is at ^ 64
    }
  }
  auto stage_x = [&](int slot) {
#pragma unroll
    for (int q = 0; q < XDMA; ++q) {
      __builtin_amdgcn_global_load_lds(U2_GLB_PTR(xq[q]), U2_LDS_PTR(xring + slot * XSLOT + w * 4096 + q * 1024), 16, 0, 0);
      xq[q] += 32;
    }
  };
  // Inline-asm loads (the compiler would drain the LDS-DMA ring in front of the first use of a load it can see).  The
  // registers are loaded in step s and split in step s + 1, i.e. loop-carried, and the compiler - which does not know the data is
  // in flight - is free to move them (it did: v_mov at the back-edge, in front of a wait that used to sit at the top of the
  // next step; the ~120 MFMAs in between usually covered the latency, and when they did not a wave split stale registers:
  // 1 run in 6 of the GPU test-suite with thousands of wrong labels).  The wait for x(s + 1) therefore closes step s, so that
  // nothing is in flight when the back-edge is taken.
  f32x4 raw[2][2];
  auto load_x = [&]() {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:64"
                   : "=&v"(raw[m][0]), "=&v"(raw[m][1]) : "v"(xp[m]) : "memory");
      xp[m] += 32;
    }
  };
  // centroids: instruction i of wave w fills LDS rows (w * 5 + i) * 16 .. + 15 (row = plane * 320 + centroid)
  const bf16_t* cp[CDMA];
#pragma unroll
  for (int i = 0; i < CDMA; ++i) {
    const int row = (w * CDMA + i) * 16 + (lane >> 2);
    cp[i] = chl + (size_t)row * D + (((lane & 3) ^ ks_swz(lane >> 2)) << 3);
  }
  auto stage_c = [&](int buf) {
#pragma unroll
    for (int i = 0; i < CDMA; ++i) {
      if (!X3 || w * CDMA + i < KS_NB)
        __builtin_amdgcn_global_load_lds(U2_GLB_PTR(cp[i]), U2_LDS_PTR(ks_smem + buf * CSTAGE + (w * CDMA + i) * 1024), 16, 0, 0);
      cp[i] += 32;
    }
  };
  // X3: centroid LDS-DMA instructions THIS wave issues per stage (the counted waits below are per wave)
  const int cw = min(max(KS_NB - w * CDMA, 0), CDMA);
  // s_waitcnt vmcnt(XDMA * NX + cw): cw is wave-uniform but not a compile-time constant
#define U2_KS_WAIT_VM(NX)                                                                                                       \
  do {                                                                                                                           \
    if (cw == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XDMA * (NX) + 3) : "memory");                                          \
    else if (cw == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XDMA * (NX) + 2) : "memory");                                     \
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XDMA * (NX)) : "memory");                                                      \
  } while (0)
  const unsigned lds0 = (unsigned)(size_t)U2_LDS_PTR(ks_smem);
  const int boff = fr * 64 + ((fg ^ ks_swz(fr)) << 4);  // B fragment of block nb, plane pl: + (pl * 320 + nb * 16) * 64

  f32x4 acc[2][KS_NB];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int nb = 0; nb < KS_NB; ++nb) acc[m][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float n2[2] = {0.f, 0.f};

  // in flight when step s begins: centroids(s + 1) only - centroids(s) and x(s) were waited for at the end of step s - 1
  // (NP = 1: x(s + 1) and centroids(s + 1))
  if constexpr (X3) {
    // issue order of the steady state from the start: x(0) | c(0) x(1) | c(1) x(2); complete before step 0: x(0), c(0)
    stage_x(0);
    stage_c(0);
    if (nsteps > 1) { stage_x(1); stage_c(1); }
    if (nsteps > 2) stage_x(2);
    if (nsteps > 2) U2_KS_WAIT_VM(2);            // x(1), c(1), x(2) may be in flight
    else if (nsteps > 1) U2_KS_WAIT_VM(1);       // x(1), c(1)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if constexpr (XR && CRING == 3) {
    stage_x(0);
    stage_c(0);
    if (nsteps > 1) {
      stage_x(1);
      stage_c(1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XDMA + CDMA) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else if constexpr (XR) {   // ring of 2: in flight when step s begins: x(s + 1) only
    stage_c(0);
    stage_x(0);
    if (nsteps > 1) {
      stage_x(1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XDMA) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else {
    stage_c(0);
    load_x();
    if (nsteps > 1) {
      stage_c(1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CDMA) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("" : "+v"(raw[0][0]), "+v"(raw[0][1]), "+v"(raw[1][0]), "+v"(raw[1][1])::"memory");
  }
  // One step; LOAD: x(s + 1) is fetched, STAGE: centroids(s + 2) are staged.  The three forms (steady state, last but one,
  // last) are separate straight-line instances, so that the wait that closes a step is unconditional in the generated code
  // (tools/check_inflight_moves.py follows every path of the control-flow graph, also the infeasible one that skips two
  // complementary conditional waits).
  auto step = [&](int s, auto load_tag, auto stage_tag) {
    constexpr bool LOAD = decltype(load_tag)::value, STAGE = decltype(stage_tag)::value;
    __builtin_amdgcn_s_barrier();  // stage s is complete for every wave, and every wave is done with stage s - 1
    asm volatile("" ::: "memory");
    if constexpr (XR) {
      // this step's x out of the wave's slot; the slot is free for x(s + 2) once the reads have returned
      const int xsl = X3 ? s % 3 : (s & 1);
      const unsigned xs = (unsigned)(size_t)U2_LDS_PTR(xring) + (unsigned)(xsl * XSLOT);
#pragma unroll
      for (int m = 0; m < 2; ++m)
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=&v"(raw[m][0]), "=&v"(raw[m][1]) : "v"(xs + xrd[m]), "v"(xs + (xrd[m] ^ 64u)) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0][0]), "+v"(raw[0][1]), "+v"(raw[1][0]), "+v"(raw[1][1])::"memory");
      if constexpr (X3) {           // (LOAD: centroids(s + 2) exist, STAGE: x(s + 3) exists) centroids first, see the head comment
        if (LOAD) stage_c((s + 2) % 3);
        if (STAGE) stage_x(xsl);
      } else {
        if constexpr (CRING == 2) {   // centroids(s + 1) go first: the wait that closes the step leaves only x(s + 2) in flight
          if (LOAD) stage_c((s + 1) & 1);
        }
        if (STAGE) stage_x(s & 1);
      }
    }
    // split this step's x into its two bf16 pieces (MFMA A operands)
    s16x8 ah[2], al[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      uint32_t h[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v0 = raw[m][e >> 1][(e & 1) * 2], v1 = raw[m][e >> 1][(e & 1) * 2 + 1];
        n2[m] += v0 * v0 + v1 * v1;
        h[e] = ks_pack(v0, v1);
        if constexpr (NP == 3) l[e] = ks_pack(v0 - __uint_as_float(h[e] << 16), v1 - __uint_as_float(h[e] & 0xffff0000u));
        else l[e] = 0u;
      }
      ah[m] = *reinterpret_cast<const s16x8*>(h);
      al[m] = *reinterpret_cast<const s16x8*>(l);
    }
    if constexpr (!XR) {
      if (LOAD) load_x();
    }
    if constexpr (CRING == 3 && !X3) {
      if (STAGE) stage_c((s + 2) % KS_RING);
    }
    // Two centroid blocks at a time, piece by piece: consecutive MFMAs go to four different accumulators, so the three products
    // of one accumulator (hi.hi, hi.lo, lo.hi) are four issue slots apart instead of back to back (-4 % kernel time; a second
    // register set that keeps two steps of x in flight from HBM changed nothing: the waves' 43 % parked cycles - PMC, round 3 -
    // were not the x loads).  Round 4: they were the centroid fragments - the compiler had scheduled every pair as
    // `4 x ds_read_b128; s_waitcnt lgkmcnt(0); 12 x v_mfma`, ten exposed LDS round trips per step with both waves of a SIMD in
    // the same phase.  The fragments of pair g + 1 are now requested in front of the MFMAs of pair g (two register sets, the
    // fences keep the order), so a read has twelve MFMAs to land.
    // The reads and their waits are inline asm: with plain loads the compiler's wait-count pass answers every second pair with
    // s_waitcnt lgkmcnt(0) although LDS returns in order (it will not count across the LDS-DMA in flight), which exposes the
    // round trip it was supposed to hide.  The wait names the four registers it releases, so no MFMA can move in front of it.
    s16x8 bq[2][4];   // [set][hi block 0, hi block 1, lo block 0, lo block 1]
    const unsigned sba = lds0 + (unsigned)((s % CRING) * CSTAGE) + (unsigned)boff;
#define U2_KS_LDQ(SET, NB)                                                                                                    \
    if constexpr (NP == 3)                                                                                                     \
      asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\t"       \
                   "ds_read_b128 %3, %4 offset:%8"                                                                             \
                   : "=&v"(bq[SET][0]), "=&v"(bq[SET][1]), "=&v"(bq[SET][2]), "=&v"(bq[SET][3])                               \
                   : "v"(sba), "n"((NB) * 1024), "n"(((NB) + 1) * 1024), "n"(KS_KMAX * 64 + (NB) * 1024),                     \
                     "n"(KS_KMAX * 64 + ((NB) + 1) * 1024)                                                                    \
                   : "memory");                                                                                                \
    else                                                                                                                       \
      asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"                                           \
                   : "=&v"(bq[SET][0]), "=&v"(bq[SET][1]) : "v"(sba), "n"((NB) * 1024), "n"(((NB) + 1) * 1024) : "memory")
#define U2_KS_WAIT(SET, CNT)                                                                                                  \
    if constexpr (NP == 3)                                                                                                     \
      asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(bq[SET][0]), "+v"(bq[SET][1]), "+v"(bq[SET][2]), "+v"(bq[SET][3]) : "n"(CNT) : "memory"); \
    else                                                                                                                       \
      asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bq[SET][0]), "+v"(bq[SET][1]) : "n"((CNT) / 2) : "memory")
#define U2_KS_PAIR(SET, NB)                                                                                                   \
    {                                                                                                                          \
      const s16x8 bh0 = bq[SET][0], bh1 = bq[SET][1], bl0 = bq[SET][2], bl1 = bq[SET][3];                                     \
      acc[0][NB] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[0], bh0, acc[0][NB], 0, 0, 0);                                  \
      acc[1][NB] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[1], bh0, acc[1][NB], 0, 0, 0);                                  \
      acc[0][NB + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[0], bh1, acc[0][NB + 1], 0, 0, 0);                          \
      acc[1][NB + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[1], bh1, acc[1][NB + 1], 0, 0, 0);                          \
      if constexpr (NP == 3) {                                                                                                 \
        acc[0][NB] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[0], bl0, acc[0][NB], 0, 0, 0);                                \
        acc[1][NB] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[1], bl0, acc[1][NB], 0, 0, 0);                                \
        acc[0][NB + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[0], bl1, acc[0][NB + 1], 0, 0, 0);                        \
        acc[1][NB + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[1], bl1, acc[1][NB + 1], 0, 0, 0);                        \
        acc[0][NB] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[0], bh0, acc[0][NB], 0, 0, 0);                                \
        acc[1][NB] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[1], bh0, acc[1][NB], 0, 0, 0);                                \
        acc[0][NB + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[0], bh1, acc[0][NB + 1], 0, 0, 0);                        \
        acc[1][NB + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[1], bh1, acc[1][NB + 1], 0, 0, 0);                        \
      }                                                                                                                        \
    }
    // pair g + 1 is requested in front of the MFMAs of pair g: the wait of pair g leaves the 4 younger reads in flight
#define U2_KS_GROUP(SET, NB)                                                                                                  \
    U2_KS_LDQ((SET) ^ 1, (NB) + 2);                                                                                            \
    U2_KS_WAIT(SET, 4);                                                                                                        \
    U2_KS_PAIR(SET, NB)
    static_assert(KS_NB == 20, "the unrolled pair sequence below covers 20 centroid blocks");
    U2_KS_LDQ(0, 0);
    U2_KS_GROUP(0, 0) U2_KS_GROUP(1, 2) U2_KS_GROUP(0, 4) U2_KS_GROUP(1, 6) U2_KS_GROUP(0, 8)
    U2_KS_GROUP(1, 10) U2_KS_GROUP(0, 12) U2_KS_GROUP(1, 14) U2_KS_GROUP(0, 16)
    U2_KS_WAIT(1, 0);
    U2_KS_PAIR(1, 18)
#undef U2_KS_GROUP
#undef U2_KS_PAIR
#undef U2_KS_WAIT
#undef U2_KS_LDQ
    // close the step: centroids(s + 1) and x(s + 1) have landed, only centroids(s + 2) stays in flight over the back-edge
    if constexpr (X3) {
      // complete behind this wait: everything up to c(s + 1), i.e. x(s + 1) and c(s + 1); may stay in flight: x(s + 2), c(s + 2), x(s + 3)
      if (LOAD && STAGE) U2_KS_WAIT_VM(2);
      else if (LOAD) U2_KS_WAIT_VM(1);                        // s = nsteps - 3: x(s + 2), c(s + 2)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
    if (STAGE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(!XR ? CDMA : CRING == 3 ? XDMA + CDMA : XDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if constexpr (!XR) asm volatile("" : "+v"(raw[0][0]), "+v"(raw[0][1]), "+v"(raw[1][0]), "+v"(raw[1][1])::"memory");
  };
  int s = 0;
  if constexpr (X3) {
    for (; s + 3 < nsteps; ++s) step(s, std::true_type{}, std::true_type{});     // c(s + 2) and x(s + 3) exist
    if (s + 2 < nsteps) { step(s, std::true_type{}, std::false_type{}); ++s; }   // s = nsteps - 3
    if (s + 1 < nsteps) { step(s, std::false_type{}, std::false_type{}); ++s; }  // s = nsteps - 2
    step(s, std::false_type{}, std::false_type{});
  } else {
  for (; s + 2 < nsteps; ++s) step(s, std::true_type{}, std::true_type{});
  if (s + 1 < nsteps) { step(s, std::true_type{}, std::false_type{}); ++s; }
  step(s, std::false_type{}, std::false_type{});
  }
  __syncthreads();
  // |x| per point: the four k-chunk lanes of a row, then through LDS into the D layout (lane (fg, fr): points fg * 4 + r)
  float* norms = reinterpret_cast<float*>(ks_smem);
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    float v = n2[m];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (fg == 0) norms[w * 32 + m * 16 + fr] = sqrtf(v);
  }
  __syncthreads();
  const float cmax = sqrtf(__uint_as_float(*cmax2));
  // Branch-free arg-min (round 4: the first version - km_less per element, one global load of |c_j|^2 per use, ds_bpermute
  // shuffles - compiled to ~700 divergent branches and was 0.32 of the kernel's 1.8 ms, with the matrix pipe idle meanwhile).
  // Round 6 (from kmeans_coarse_kernel, where the stamps put a third of a work-group's life here): inside a lane the block index rides
  // in the five low mantissa bits of the distance ((v & ~31) | nb: v_and_or_b32), so best and second best are v_min_f32 + v_med3_f32
  // per value with no compares or selects; the 2^-18 of a distance those bits cost is added to the margin (twice, generously).  The
  // cross-lane steps carry the index beside the values - nine index bits would cost 2^-14, which is the fine pass's whole margin - and
  // the 32 results of a wave are moved into 32 lanes and written by one store.
  // NaN needs no ordering here: a NaN or Inf in x makes |x| and therefore the margin NaN / Inf, one in c makes max|c| NaN, and
  // `!(gap >= margin)` then sends the point to the exact kernel, which orders NaNs like torch.argmin; v_min skips a NaN operand and
  // v_med3 answers min3 when it sees one (second best = best: undecided).  Ties need no rule either: a gap of zero is below any margin.
  float cnr[KS_NB];
#pragma unroll
  for (int nb = 0; nb < KS_NB; ++nb) {
    const int j = nb * 16 + fr;
    cnr[nb] = cn[min(j, K - 1)];
    cnr[nb] = j < K ? cnr[nb] : 3.0e38f;   // not an infinity: the index bits would turn it into a NaN
  }
  float kb = INFINITY, ks = INFINITY;   // best / second best / centroid of the point this lane writes (lanes fr < 8: point block
  int kj = 0;                           // fr >> 2, row fg * 4 + (fr & 3))
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    float b[4], s2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { b[r] = INFINITY; s2[r] = INFINITY; }
#pragma unroll
    for (int nb = 0; nb < KS_NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = fmaf(acc[m][nb][r], -2.f, cnr[nb]);   // cn - 2 x.c
        v = __uint_as_float((__float_as_uint(v) & 0xffffffe0u) | (unsigned)nb);
        s2[r] = __builtin_amdgcn_fmed3f(b[r], s2[r], v);  // b <= s2: the middle one is the new second best
        b[r] = kc_min(b[r], v);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int bj = (int)(__float_as_uint(b[r]) & 31u) * 16 + fr;
      float bb = b[r], ss = s2[r];
      // the 16 lanes of a DPP row hold the 16 centroids of every block: xor 1, xor 2 (quad permutes), then the other quad of
      // the half row, then the other half row (all lanes of a quad / a half row agree by then)
#define U2_KS_MERGE(CTRL)                                                                                                   \
      {                                                                                                                      \
        const float ob = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bb), CTRL, 0xf, 0xf, true));           \
        const float os = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ss), CTRL, 0xf, 0xf, true));           \
        const int oj = __builtin_amdgcn_update_dpp(0, bj, CTRL, 0xf, 0xf, true);                                            \
        const bool lt = ob < bb || (ob == bb && oj < bj);                                                                    \
        ss = kc_min(kc_min(ss, os), lt ? bb : ob);                                                                           \
        bj = lt ? oj : bj;                                                                                                   \
        bb = kc_min(bb, ob);                                                                                                 \
      }
      U2_KS_MERGE(0xB1)    // quad_perm [1, 0, 3, 2]
      U2_KS_MERGE(0x4E)    // quad_perm [2, 3, 0, 1]
      U2_KS_MERGE(0x141)   // row_half_mirror
      U2_KS_MERGE(0x140)   // row_mirror
#undef U2_KS_MERGE
      const bool mine = fr == m * 4 + r;
      kb = mine ? bb : kb;
      ks = mine ? ss : ks;
      kj = mine ? bj : kj;
    }
  }
  const int pl = ((fr >> 2) & 1) * 16 + fg * 4 + (fr & 3);
  const int p = p0 + pl;
  if (fr < 8 && p < N) {
    const int row = rows ? rows[p] : p;
    labels[row] = (long long)kj;
    // screening margin + the mantissa bits the block index took: |distance| <= |c|^2 + 2 |x| |c|
    const float xn = norms[w * 32 + pl];
    const float margin = margin_rel * cmax * xn + 7.6293945e-6f * (cmax * cmax + 2.f * xn * cmax);
    if (!(ks - kb >= margin)) list[atomicAdd(nlist, 1)] = row;  // also: NaN anywhere, K == 1
  }
}


// ---- coarse pass over the shadow (round 6) ----------------------------------------------------------------------------------------
// One product per pair, all points, x from the fp16 shadow (km_shadow_kernel) - the first pass of the two-level screening when a shadow is given.
// Same products and wave tile (32 points x 320 centroids) as kmeans_screen_kernel<1>; what differs (steps and their measurements:
// profiles/r06_km_coarse.txt):
//  * Persistent work-groups (one per CU, tiles blockIdx.x, + gridDim.x, ...): the request streams run on across the tile boundary.
//  * fp16 operands (v_mfma_f32_16x16x32_f16: the rate of the bf16 instruction, three more significant bits) and a margin per point from
//    the exact norms of what their rounding dropped instead of a worst case (at the arg-min below).
//  * The LDS-DMA queues are split by wave: vmcnt retires in order, so a wave that requests both centroids and x cannot wait for "the
//    centroids of the next step" without also waiting for every x it requested before them - two steps of slack whatever the ring
//    depth.  Waves 0-3 request the centroid stages (five 1 KB instructions each, ring of three 20 KB stages), waves 4-7 the x slots
//    of the whole work-group (four 1 KB groups each, ring of six 16 KB slots) and wait for x(s + 1) only: x(s + 2 .. s + 5) stay
//    in flight, 64 KB of bf16 per CU.  The step-start barrier publishes both to all eight waves.
//  * The arg-min.  s_memtime stamps (tools/exp/km_trace.sh) put 31 % of a work-group's time into the epilogue of the first version of this
//    kernel - 3400 instructions per wave for 160 distances per lane, with the matrix pipe idle and nothing in flight.  Now the block
//    index rides in the low mantissa bits of the distance ((v & ~31) | nb, one v_and_or_b32), so "best and second best" is v_min_f32 +
//    v_med3_f32 per value with no compares or selects; the cross-lane steps carry (best | 9-bit centroid index, second) only; and the 32
//    results of a wave are moved into 32 lanes and written by one store.  The bits given up (2^-14 of the distance at most) are added to
//    the margin, so a point is still either decided correctly or handed to the finer passes; ties need no rule here - a gap of zero is
//    below any margin.
#ifdef U2_KM_TRACE
// measurement build (tools/exp/km_trace.sh): s_memtime of waves 0 (centroid requests) and 4 (x requests) at kernel entry, first barrier,
// behind the loop and behind the arg-min of the work-group's first tile, kernel exit; read back through u2_km_trace_dump
__device__ unsigned long long km_trace[8192 * 16];   // [work-group][role][8]
#define U2_KM_STAMP(I)                                                                                  \
  if ((w == 0 || w == 4) && lane == 0 && blockIdx.x < 8192) km_trace[blockIdx.x * 16 + (w >> 2) * 8 + (I)] = __builtin_amdgcn_s_memtime()
#else
#define U2_KM_STAMP(I)
#endif
// timing-only ablations of the loop (results are wrong): 1 no MFMAs, 2 no centroid-fragment reads behind the first group, 4 no LDS-DMA
// requests in the loop, 8 no barrier
#ifndef U2_KC_ABL
#define U2_KC_ABL 0
#endif
constexpr int KC_SLOTS = 6;
constexpr int KC_SLOT = KS_PTS * 64;           // 16 groups x 1 KB
constexpr int KC_STAGE = KS_KMAX * 64;         // hi plane: 320 rows x 64 B
constexpr int KC_RINGS = KS_RING * KC_STAGE + KC_SLOTS * KC_SLOT;
constexpr int KC_LDS = KC_RINGS + KS_KMAX * 4; // + |c_j|^2
// chl / cn / cs: the centred centroids of csplit_kernel (chc, cnc, {max |c - mu|^2, max |(c - mu) - chc / S|^2}); cmax2: max |c|^2
// of the centroids as they are (the exact kernel's frame); xnorm: km_xnorm_kernel's three arrays
__global__ __launch_bounds__(512) void kmeans_coarse_kernel(const unsigned char* __restrict__ xh, const float* __restrict__ xnorm,
                                                            const bf16_t* __restrict__ chl, const float* __restrict__ cn,
                                                            const unsigned* __restrict__ cs, const unsigned* __restrict__ cmax2,
                                                            long long* __restrict__ labels,
                                                            int* __restrict__ list, int* __restrict__ nlist, int N, int D, int K,
                                                            float margin_rel, const int* __restrict__ gate, int gate_want,
                                                            float2* __restrict__ part, const float* __restrict__ aux) {
  // part != nullptr (K > 320, one launch per block of 320 centroids - chl, cn, K are the block's): the point's (best | index inside the
  // block, second best) go to part[p] and km_chunk_merge_kernel decides over all blocks
  extern __shared__ __attribute__((aligned(16))) unsigned char ks_smem[];
  if (gate && (*gate == 1) != (gate_want != 0)) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int nsteps = D >> 5;
  // Persistent: the work-group takes the tiles blockIdx.x, blockIdx.x + gridDim.x, ... (one work-group per CU) and both request streams
  // run on across the tile boundary - the x of the next tile's first steps is in flight while this tile's last steps and its arg-min
  // run (as one launch per tile, 7-9 % of a work-group's life was the wait for its first stage).  g counts steps over all tiles of the
  // work-group: stage g % 3, slot g % KC_SLOTS.
  const int ntiles = (N + KS_PTS - 1) / KS_PTS;
  const int nmine = ((int)blockIdx.x < ntiles) ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int T = nmine * nsteps;
  if (T == 0) return;
  unsigned char* const xring = ks_smem + KS_RING * KC_STAGE;
  float* const cnl = reinterpret_cast<float*>(ks_smem + KC_RINGS);
  const bool xrole = w >= 4;
  U2_KM_STAMP(0);
  // |c_j|^2 for the arg-min, in LDS (a huge finite value for j >= K: an infinity would turn into a NaN when the block index is or-ed in)
  if (tid < KS_KMAX) cnl[tid] = tid < K ? cn[tid] : 3.0e38f;
  // one address set per wave: centroid rows (w * 5 + i) * 16 .. + 15 (64 bytes further per step, back to the start with every tile) or
  // x groups (w - 4) * 4 + i of the tile (ntiles * 16 KB further per step); i = 4 is unused by the x waves
  const size_t xstep = (size_t)ntiles * 16 * 1024;
  const unsigned char* gp[5];
  int rs = 0, rtile = blockIdx.x;                // step and tile of the next request
  auto point_requests = [&]() {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int row = (w * 5 + i) * 16 + (lane >> 2);
      gp[i] = xrole ? xh + ((size_t)rtile * 16 + (size_t)((w - 4) * 4 + (i & 3))) * 1024 + lane * 16
                    : reinterpret_cast<const unsigned char*>(chl + (size_t)min(row, KS_KMAX - 1) * D + (((lane & 3) ^ ks_swz(lane >> 2)) << 3));
    }
  };
  point_requests();
  auto advance_requests = [&]() {
    if (++rs == nsteps) {
      rs = 0;
      rtile += gridDim.x;
      point_requests();
    } else {
      const size_t gstep = xrole ? xstep : 64;
#pragma unroll
      for (int i = 0; i < 5; ++i) gp[i] += gstep;
    }
  };
  auto stage_c = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 5; ++i)
      __builtin_amdgcn_global_load_lds(U2_GLB_PTR(gp[i]), U2_LDS_PTR(ks_smem + buf * KC_STAGE + (w * 5 + i) * 1024), 16, 0, 0);
    advance_requests();
  };
  auto stage_x = [&](int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds(U2_GLB_PTR(gp[i]), U2_LDS_PTR(xring + slot * KC_SLOT + ((w - 4) * 4 + i) * 1024), 16, 0, 0);
    advance_requests();
  };
  // s_waitcnt vmcnt for a wave-uniform count that is not a compile-time constant: `n` steps of x (four instructions each) may stay in flight
  auto wait_x = [&](int n) {
    switch (n) {
      case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  static_assert(KC_SLOTS - 2 == 4, "wait_x covers 0 .. KC_SLOTS - 2 steps in flight");
  const unsigned lds0 = (unsigned)(size_t)U2_LDS_PTR(ks_smem);
  const unsigned boff = (unsigned)(fr * 64 + ((fg ^ ks_swz(fr)) << 4));                               // B fragment of block nb: + nb * 1024
  const unsigned xoff = (unsigned)(size_t)U2_LDS_PTR(xring) + (unsigned)(w * 2048 + lane * 16);      // A fragment of point block m: + m * 1024

  // the plain load above is the oldest entry of the queue: it has left it when any of the counted waits below is satisfied
  if (!xrole) {                                  // c(0), c(1); complete before step 0: c(0)
    stage_c(0);
    if (T > 1) {
      stage_c(1);
      asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else {                                       // x(0 .. KC_SLOTS - 2) (step g requests x(g + KC_SLOTS - 1)); complete before step 0: x(0)
    const int npre = min(KC_SLOTS - 1, T);
    for (int j = 0; j < npre; ++j) stage_x(j);
    wait_x(npre - 1);
  }
  U2_KM_STAMP(1);
  const float cmax = sqrtf(__uint_as_float(cs[0])), clomax = sqrtf(__uint_as_float(cs[1])), cmax_o = sqrtf(__uint_as_float(*cmax2));
  const float m2s = -2.f * aux[1] * aux[1];      // -2 / S^2
  int cbuf = 0, xslot = 0, g = 0;                // g % 3, g % KC_SLOTS
  for (int it = 0; it < nmine; ++it) {
    f32x4 acc[2][KS_NB];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int nb = 0; nb < KS_NB; ++nb) acc[m][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the point this lane writes (lanes fr < 8: point block fr >> 2, row fg * 4 + (fr & 3)); xnorm is padded to whole tiles
    const int pmine = ((int)blockIdx.x + it * (int)gridDim.x) * KS_PTS + w * 32 + ((fr >> 2) & 1) * 16 + fg * 4 + (fr & 3);
    float xn, xlo, xno;       // |y|, |y - bf16(y)| for y = x_p - mu, and |x_p|
    for (int s = 0; s < nsteps; ++s, ++g) {
      if (!(U2_KC_ABL & 8)) __builtin_amdgcn_s_barrier();   // stage / slot g are complete for every wave, and every wave is done with step g - 1
      asm volatile("" ::: "memory");
      s16x8 ah[2];
      const unsigned xs = xoff + (unsigned)(xslot * KC_SLOT);
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(ah[0]), "=&v"(ah[1]) : "v"(xs) : "memory");
      // this step's requests go to the stage / slot that every wave read in step g - 1 (the barrier above has seen that)
      if (U2_KC_ABL & 4) {
      } else if (!xrole) {
        if (g + 2 < T) stage_c(cbuf == 0 ? 2 : cbuf - 1);
      } else {
        if (g + KC_SLOTS - 1 < T) stage_x(xslot == 0 ? KC_SLOTS - 1 : xslot - 1);
      }
      // centroid fragments four blocks at a time, group g + 1 requested in front of the eight MFMAs of group g (in-order LDS returns,
      // counted waits that name the registers they release)
      s16x8 bq[2][4];
      const unsigned sba = lds0 + (unsigned)(cbuf * KC_STAGE) + boff;
#define U2_KC_LDQ(SET, NB)                                                                                                    \
      if (!(U2_KC_ABL & 2) || (NB) < 8)                                                                                        \
      asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\t"       \
                   "ds_read_b128 %3, %4 offset:%8"                                                                             \
                   : "=&v"(bq[SET][0]), "=&v"(bq[SET][1]), "=&v"(bq[SET][2]), "=&v"(bq[SET][3])                               \
                   : "v"(sba), "n"((NB) * 1024), "n"(((NB) + 1) * 1024), "n"(((NB) + 2) * 1024), "n"(((NB) + 3) * 1024) : "memory")
#define U2_KC_WAIT(SET, CNT)                                                                                                  \
      asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(bq[SET][0]), "+v"(bq[SET][1]), "+v"(bq[SET][2]), "+v"(bq[SET][3]) : "n"(CNT) : "memory")
#define U2_KC_QUAD(SET, NB)                                                                                                   \
      if (!(U2_KC_ABL & 1)) {                                                                                                  \
        const s16x8 b0 = bq[SET][0], b1 = bq[SET][1], b2 = bq[SET][2], b3 = bq[SET][3];                                        \
        acc[0][NB] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ks_h16x8, ah[0]), __builtin_bit_cast(ks_h16x8, b0), acc[0][NB], 0, 0, 0);                                  \
        acc[1][NB] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ks_h16x8, ah[1]), __builtin_bit_cast(ks_h16x8, b0), acc[1][NB], 0, 0, 0);                                  \
        acc[0][NB + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ks_h16x8, ah[0]), __builtin_bit_cast(ks_h16x8, b1), acc[0][NB + 1], 0, 0, 0);                          \
        acc[1][NB + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ks_h16x8, ah[1]), __builtin_bit_cast(ks_h16x8, b1), acc[1][NB + 1], 0, 0, 0);                          \
        acc[0][NB + 2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ks_h16x8, ah[0]), __builtin_bit_cast(ks_h16x8, b2), acc[0][NB + 2], 0, 0, 0);                          \
        acc[1][NB + 2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ks_h16x8, ah[1]), __builtin_bit_cast(ks_h16x8, b2), acc[1][NB + 2], 0, 0, 0);                          \
        acc[0][NB + 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ks_h16x8, ah[0]), __builtin_bit_cast(ks_h16x8, b3), acc[0][NB + 3], 0, 0, 0);                          \
        acc[1][NB + 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ks_h16x8, ah[1]), __builtin_bit_cast(ks_h16x8, b3), acc[1][NB + 3], 0, 0, 0);                          \
      }
      static_assert(KS_NB == 20, "the unrolled group sequence below covers 20 centroid blocks");
      U2_KC_LDQ(0, 0);
      U2_KC_LDQ(1, 4);
      // the x fragments were requested first: behind this wait they and group 0 are there, group 1 is in flight
      asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah[0]), "+v"(ah[1]), "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[0][2]), "+v"(bq[0][3])::"memory");
      U2_KC_QUAD(0, 0)
      U2_KC_LDQ(0, 8);
      U2_KC_WAIT(1, 4);
      U2_KC_QUAD(1, 4)
      U2_KC_LDQ(1, 12);
      U2_KC_WAIT(0, 4);
      U2_KC_QUAD(0, 8)
      U2_KC_LDQ(0, 16);
      U2_KC_WAIT(1, 4);
      U2_KC_QUAD(1, 12)
      U2_KC_WAIT(0, 0);
      U2_KC_QUAD(0, 16)
#undef U2_KC_QUAD
#undef U2_KC_WAIT
#undef U2_KC_LDQ
      // close the step: the centroid waves leave c(g + 2) in flight, the x waves x(g + 2 .. g + KC_SLOTS - 1) as far as they exist
      if (!xrole) {
        if (g + 2 < T) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        wait_x(min(max(T - g - 2, 0), KC_SLOTS - 2));
      }
      cbuf = cbuf == 2 ? 0 : cbuf + 1;
      xslot = xslot == KC_SLOTS - 1 ? 0 : xslot + 1;
    }
    if (it == 0) { U2_KM_STAMP(2); }
    // ---- arg-min of the tile (registers and cnl only: no barrier - the waves meet again at the barrier of the next step) ----
    {
      // The two norms, requested now and waited for at the end.  Inline asm: as plain loads the compiler sinks them into the conditional
      // block at the end, right in front of their use.  They are younger than every ring request, so the wait also waits for those -
      // which have had a step and the arg-min to arrive.  (Requested in front of the tile's last ring request instead, so that the wait
      // can leave that one in flight: equal, 1.123-1.134 ms per iteration for all three placements, tools/exp/km_norms_ab.sh.)
      const float* np = xnorm + pmine;
      asm volatile("global_load_dword %0, %3, off\n\tglobal_load_dword %1, %4, off\n\tglobal_load_dword %2, %5, off"
                   : "=&v"(xn), "=&v"(xlo), "=&v"(xno) : "v"(np), "v"(np + (size_t)ntiles * KS_PTS), "v"(np + 2 * (size_t)ntiles * KS_PTS)
                   : "memory");
    }
    float cnr[KS_NB];
#pragma unroll
    for (int nb = 0; nb < KS_NB; ++nb) cnr[nb] = cnl[nb * 16 + fr];
    float kb = INFINITY, ks = INFINITY;   // (best | centroid index, second best) of the point this lane writes
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float b[4], s2[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { b[r] = INFINITY; s2[r] = INFINITY; }
#pragma unroll
      for (int nb = 0; nb < KS_NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaf(acc[m][nb][r], m2s, cnr[nb]);                         // cn - 2 x.c (the products carry the shadow's scale twice)
          v = __uint_as_float((__float_as_uint(v) & 0xffffffe0u) | (unsigned)nb);
          s2[r] = __builtin_amdgcn_fmed3f(b[r], s2[r], v);                       // b <= s2: the middle one is the new second best
          b[r] = kc_min(b[r], v);
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned ub = __float_as_uint(b[r]);
        float bb = __uint_as_float((ub & 0xfffffe00u) | ((ub & 31u) << 4) | (unsigned)fr), ss = s2[r];
        // the 16 lanes of a DPP row hold the 16 centroids of every block: xor 1, xor 2 (quad permutes), the other quad of the half
        // row, the other half row
#define U2_KC_MERGE(CTRL)                                                                                                   \
        {                                                                                                                    \
          const float ob = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bb), CTRL, 0xf, 0xf, true));         \
          const float os = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ss), CTRL, 0xf, 0xf, true));         \
          ss = kc_min(kc_min(ss, os), kc_max(bb, ob));                                                                       \
          bb = kc_min(bb, ob);                                                                                               \
        }
        U2_KC_MERGE(0xB1)    // quad_perm [1, 0, 3, 2]
        U2_KC_MERGE(0x4E)    // quad_perm [2, 3, 0, 1]
        U2_KC_MERGE(0x141)   // row_half_mirror
        U2_KC_MERGE(0x140)   // row_mirror
#undef U2_KC_MERGE
        const bool mine = fr == m * 4 + r;
        kb = mine ? bb : kb;
        ks = mine ? ss : ks;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(xn), "+v"(xlo), "+v"(xno)::"memory");
    if (part) {
      if (fr < 8 && pmine < N) part[pmine] = float2{kb, ks};
    } else if (fr < 8 && pmine < N) {
      labels[pmine] = (long long)(__float_as_uint(kb) & 511u);
      // Per-point margin, in the translated frame (x, c stand for x - mu, c - mu).  What this pass does not see of a product is
      // x.c - x_hi.c_hi = x_lo.c + x_hi.c_lo, at most |x_lo| |c| + |x_hi| |c_lo| (Cauchy-Schwarz; x_lo = x - (its fp16 shadow) / S and its norm are
      // exact, from u2_kmeans_prepare; |x_hi| <= 1.004 |x|): twice that per distance, four times between two distances.  With the norms
      // as they are instead of a worst case (bf16: 2^-9 of |x|, |c| each, 2^-6 |x| max|c| in all - the 0.02 of kmeans_screen_kernel<1>;
      // measured 0.013 with bf16 operands) the margin is ~0.002 |x| max|c| on fp32 data with the fp16 operands, and zero for operands
      // that are fp16 values already.  + the mantissa bits the indices
      // took (2^-14 of a distance at most, |distance| <= |c|^2 + 2 |x| |c|; twice that here).  + margin_rel |x| max|c| of the
      // UNTRANSLATED operands for everything else: the labels must be the exact kernel's, whose own fp32 rounding is proportional to
      // the norms it sees (the fine pass's margin, three times), and the roundings of x - mu, c - mu.
      const float margin = 4.04f * (xlo * cmax + 1.004f * xn * clomax) + margin_rel * cmax_o * xno +
                           1.2207031e-4f * (cmax * cmax + 2.f * xn * cmax);
      // also: NaN anywhere; and an operand beyond the shadow's range (a centroid far outside the data: its products are infinite)
      if (!(ks - kb >= margin) || !(fabsf(kb) < 3.0e38f)) list[atomicAdd(nlist, 1)] = pmine;
    }
    if (it == 0) { U2_KM_STAMP(3); }
  }
  U2_KM_STAMP(4);
}

// K > 320: the per-block candidates of kmeans_coarse_kernel (part[b][N]) -> label and, where the two best of ALL centroids are closer
// than the coarse margin, an entry of the exact kernel's list.  Second best overall = min(second of the best block, best of the others).
__global__ __launch_bounds__(256) void km_chunk_merge_kernel(const float2* __restrict__ part, int nblocks, const float* __restrict__ xnorm,
                                                             int npad, const unsigned* __restrict__ cs, const unsigned* __restrict__ scal,
                                                             long long* __restrict__ labels,
                                                             int* __restrict__ list, int* __restrict__ nlist, int N, float margin_rel) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= N) return;
  float bb = INFINITY, ss = INFINITY;
  int bj = 0;
  for (int b = 0; b < nblocks; ++b) {
    const float2 v = part[(size_t)b * N + p];
    const bool lt = v.x < bb;                                  // ties: the lower block (a zero gap goes to the exact kernel anyway)
    ss = fminf(fminf(ss, v.y), lt ? bb : v.x);
    bj = lt ? b * KS_KMAX + (int)(__float_as_uint(v.x) & 511u) : bj;
    bb = fminf(bb, v.x);
  }
  labels[p] = (long long)bj;
  const float cmax = sqrtf(__uint_as_float(cs[0])), clomax = sqrtf(__uint_as_float(cs[1])), cmax_o = sqrtf(__uint_as_float(scal[0]));
  const float xn = xnorm[p], xlo = xnorm[npad + p], xno = xnorm[2 * (size_t)npad + p];
  const float margin = 4.04f * (xlo * cmax + 1.004f * xn * clomax) + margin_rel * cmax_o * xno +   // kmeans_coarse_kernel's margin
                       1.2207031e-4f * (cmax * cmax + 2.f * xn * cmax);
  if (!(ss - bb >= margin) || !(fabsf(bb) < 3.0e38f)) list[atomicAdd(nlist, 1)] = p;
}

// csum[K][D] += x rows by label; counts[K] += 1.  grid = (point chunks, D / DS); LDS holds [K][DS] partial sums.
template <int DS>
__global__ __launch_bounds__(256) void kmeans_update_kernel(const float* __restrict__ x, const long long* __restrict__ labels,
                                                            float* __restrict__ csum, float* __restrict__ counts, int N, int D,
                                                            int K, int pts_per_block) {
  // A work-group owns pts_per_block points x DS dimensions and privatises the [K][DS] partial sums in LDS.  A thread moves
  // 16 bytes (4 dimensions) of a point per load and keeps UNR independent points in flight (the first version issued one
  // 4-byte load per thread and then waited for it: 0.77 TB/s); the LDS float atomics only collide when two of the 64 points
  // of an iteration share a cluster.
  extern __shared__ float part[];  // [K][DS] then [K] counts
  float* pc = part + (size_t)K * DS;
  const int tid = threadIdx.x;
  const int d0 = blockIdx.y * DS;
  const int pb = blockIdx.x * pts_per_block;
  const int pe = min(N, pb + pts_per_block);
  for (int i = tid; i < K * DS + K; i += 256) part[i] = 0.f;
  __syncthreads();
  constexpr int QPP = DS / 4;        // 16-byte quads per point
  constexpr int PPI = 256 / QPP;     // points per pass of the work-group
  constexpr int UNR = 4;
  const int sub = tid / QPP, q = tid % QPP;
  const int dq = d0 + q * 4;
  const bool vec = (D & 3) == 0 && dq + 4 <= D;
  for (int p0 = pb + sub; p0 < pe; p0 += PPI * UNR) {
    float4 v[UNR];
    int l[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int p = p0 + u * PPI;
      l[u] = -1;
      v[u] = float4{0.f, 0.f, 0.f, 0.f};
      if (p < pe) {
        l[u] = (int)labels[p];
        const float* src = x + (size_t)p * D + dq;
        if (vec) {
          v[u] = *reinterpret_cast<const float4*>(src);
        } else {
          if (dq + 0 < D) v[u].x = src[0];
          if (dq + 1 < D) v[u].y = src[1];
          if (dq + 2 < D) v[u].z = src[2];
          if (dq + 3 < D) v[u].w = src[3];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (l[u] < 0) continue;
      float* dst = part + l[u] * DS + q * 4;
      atomicAdd(dst + 0, v[u].x);
      atomicAdd(dst + 1, v[u].y);
      atomicAdd(dst + 2, v[u].z);
      atomicAdd(dst + 3, v[u].w);
      if (q == 0 && blockIdx.y == 0) atomicAdd(&pc[l[u]], 1.f);
    }
  }
  __syncthreads();
  for (int i = tid; i < K * DS; i += 256) {
    const float v = part[i];
    const int j = i / DS, dd = i % DS;
    if (v != 0.f && d0 + dd < D) atomicAdd(csum + (size_t)j * D + d0 + dd, v);
  }
  if (blockIdx.y == 0)
    for (int j = tid; j < K; j += 256)
      if (pc[j] != 0.f) atomicAdd(counts + j, pc[j]);
}

// ---- M step as a segmented reduction (the form used when the caller provides a workspace) --------------------------------
// The privatised-LDS kernel above reads x in 256-byte pieces (64 of the 768 dimensions of a row per work-group): 0.75 TB/s.
// Here the points are first bucketed by label (histogram -> exclusive scan -> scatter of the point indices), then every
// work-group walks a run of the label-ordered index list and reads whole rows (3 KB, fully coalesced) with several rows in
// flight, summing in registers and flushing to csum only when the label changes: x is streamed exactly once.
__global__ __launch_bounds__(256) void km_hist_kernel(const long long* __restrict__ labels, int* __restrict__ counts, int N, int K) {
  extern __shared__ int hist[];
  for (int j = threadIdx.x; j < K; j += 256) hist[j] = 0;
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)N; i += (size_t)gridDim.x * 256) {
    const long long l = labels[i];
    if (l >= 0 && l < K) atomicAdd(&hist[(int)l], 1);   // a label outside [0, K) is not a member of any cluster
  }
  __syncthreads();
  for (int j = threadIdx.x; j < K; j += 256)
    if (hist[j]) atomicAdd(counts + j, hist[j]);
}

__global__ __launch_bounds__(256) void km_scan_kernel(const int* __restrict__ counts, int* __restrict__ cursor,
                                                      float* __restrict__ fcounts, int K) {
  // exclusive scan: a chunk of consecutive labels per thread, the 256 chunk sums through LDS (the first version - thread 0 walking
  // all K counts, one dependent global load each - took 17 us at K = 300, a launch that should cost 4)
  __shared__ int part[256];
  const int per = (K + 255) / 256, j0 = threadIdx.x * per, j1 = min(K, j0 + per);
  int sum = 0;
  for (int j = j0; j < j1; ++j) sum += counts[j];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int acc = part[threadIdx.x] - sum;
  for (int j = j0; j < j1; ++j) {
    const int c = counts[j];
    cursor[j] = acc;
    acc += c;
    fcounts[j] += (float)c;
  }
}

// A work-group takes KM_SCAT consecutive points: per-label counts in LDS, ONE global atomic per (work-group, label) to
// reserve that many slots behind the label's cursor, then every point takes its slot by an LDS atomic.  (One global atomic
// per point on 300 hot addresses took 0.41 ms per iteration at N = 1M; the whole M step went from 0.99 to 0.59 ms.)
constexpr int KM_SCAT = 4096;
__global__ __launch_bounds__(256) void km_scatter_kernel(const long long* __restrict__ labels, int* __restrict__ cursor,
                                                         int* __restrict__ order, int* __restrict__ lab_sorted, int N, int K) {
  extern __shared__ int slots[];  // [K] counts, then the reserved base of each label
  const int p0 = blockIdx.x * KM_SCAT, p1 = min(N, p0 + KM_SCAT);
  for (int j = threadIdx.x; j < K; j += 256) slots[j] = 0;
  __syncthreads();
  int lab[KM_SCAT / 256];
#pragma unroll
  for (int u = 0; u < KM_SCAT / 256; ++u) {
    const int i = p0 + u * 256 + threadIdx.x;
    const long long l = i < p1 ? labels[i] : -1;
    lab[u] = (l >= 0 && l < K) ? (int)l : -1;   // out-of-range labels are skipped (they would index past the LDS table)
    if (lab[u] >= 0) atomicAdd(&slots[lab[u]], 1);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < K; j += 256) {
    const int c = slots[j];
    slots[j] = c ? atomicAdd(cursor + j, c) : 0;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < KM_SCAT / 256; ++u) {
    if (lab[u] >= 0) {
      const int pos = atomicAdd(&slots[lab[u]], 1);
      order[pos] = p0 + u * 256 + threadIdx.x;
      lab_sorted[pos] = lab[u];
    }
  }
}

constexpr int KM_SEG = 256;  // label-ordered entries per work-group
template <int DPT>           // dimensions per thread: D <= 256 * DPT
__global__ __launch_bounds__(256) void km_segsum_kernel(const float* __restrict__ x, const int* __restrict__ order,
                                                        const int* __restrict__ lab_sorted, float* __restrict__ csum, int N, int D,
                                                        const int* __restrict__ total) {
  const int tid = threadIdx.x;
  // *total: the entries of the label-ordered list (the cursor of the last label once km_scatter_kernel has run); points whose label is
  // outside [0, K) were not scattered and leave unwritten slots behind it
  const int e0 = blockIdx.x * KM_SEG, e1 = min(min(N, *total), e0 + KM_SEG);
  if (e0 >= e1) return;
  float acc[DPT];
#pragma unroll
  for (int t = 0; t < DPT; ++t) acc[t] = 0.f;
  int cur = lab_sorted[e0];
  auto flush = [&](int label) {
#pragma unroll
    for (int t = 0; t < DPT; ++t) {
      const int d = tid + t * 256;
      if (label >= 0 && d < D && acc[t] != 0.f) atomicAdd(csum + (size_t)label * D + d, acc[t]);
      acc[t] = 0.f;
    }
  };
  constexpr int UNR = 8;  // rows in flight per thread
  for (int e = e0; e < e1; e += UNR) {
    float v[UNR][DPT];
    int lab[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      lab[u] = e + u < e1 ? lab_sorted[e + u] : -1;   // -1: beyond the list (e1 stops at the last scattered entry)
      const bool ok = lab[u] >= 0;
      const float* row = x + (size_t)(ok ? order[e + u] : 0) * D;
#pragma unroll
      for (int t = 0; t < DPT; ++t) {
        const int d = tid + t * 256;
        v[u][t] = (ok && d < D) ? row[d] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (lab[u] < 0) continue;
      if (lab[u] != cur) { flush(cur); cur = lab[u]; }  // wave-uniform: all threads walk the same entries
#pragma unroll
      for (int t = 0; t < DPT; ++t) acc[t] += v[u][t];
    }
  }
  flush(cur);
}

__global__ void kmeans_finalize_kernel(const float* __restrict__ csum, const float* __restrict__ counts, float* __restrict__ c,
                                       int D, int K) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)K * D) return;
  c[i] = csum[i] / counts[i / D];
}

}  // namespace

// ---- two-level screening (round 4) ---------------------------------------------------------------------------------------
// Workspace words behind the lists: {magic, coarse_off, calls since the coarse pass was switched off, -}.  The coarse (hi.hi)
// pass costs a third of the fine one and decides everything on clustered data; on unstructured data (randn: the two best of 300
// distances are closer than its margin for most points) it decides little and would only be overhead.  The flag is sticky per
// workspace: a coarse pass that leaves more than a quarter of the points undecided (over the shadow of x, where it costs a third of
// a fine pass instead of half: more than 55 %) switches itself off, every 64th call tries again.  Labels are those of the exact
// kernel in either mode - the modes differ in time only.
constexpr unsigned KS_MAGIC = 0x4b533431u;
__global__ void km_state_begin_kernel(unsigned* __restrict__ scal, unsigned* __restrict__ state, unsigned* __restrict__ cs) {
  if (threadIdx.x < 4) { scal[threadIdx.x] = 0u; cs[threadIdx.x] = 0u; }
  if (threadIdx.x == 0 && state[0] != KS_MAGIC) { state[0] = KS_MAGIC; state[1] = 0u; state[2] = 0u; state[3] = 0u; }
}

static long long km_blocks(int K) { return (K + KS_KMAX - 1) / KS_KMAX; }
static long long km_ws_base_floats(int N, int D, int K) {
  const long long blocks = km_blocks(K);
  return (((long long)K + 16 + blocks * KS_KMAX * D + 2LL * N + 32 + (blocks > 1 ? blocks * 2LL * N + 8 : 0)) + 3) & ~3LL;
}
extern "C" long long u2_kmeans_assign_workspace_floats(int N, int D, int K) {
  // |c|^2 [K] | max |c|^2, exact-list length, coarse-list length [4] | split centroids [2][320][D] bf16 | exact re-check list [N]
  // | undecided list of the coarse pass [N] | screening state [4]
  // K > 320 (u2_kmeans_assign_shadow only): the split centroids of every block of 320, and the blocks' candidates [blocks][N]{best, second}
  // behind all of that, for u2_kmeans_assign_shadow: {max |c - mu|^2, max residual^2} [4] | |c_j - mu|^2 [blocks][320] | fp16(S (c - mu)) [blocks][320][D]
  return km_ws_base_floats(N, D, K) + 8 + km_blocks(K) * KS_KMAX + km_blocks(K) * (KS_KMAX / 2) * (long long)D;
}

#ifdef U2_KM_TRACE
extern "C" int u2_km_trace_dump(unsigned long long* host, int words) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(km_trace), (size_t)words * 8);
}

#endif
// compute units of the current device (the persistent coarse pass launches one work-group per CU)
static int km_cu_count() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

// words (floats) of the 16-bit shadow itself; the norms, mu and the scale follow it
static size_t km_shadow_words(int N, int D) { return (size_t)(D >> 5) * (size_t)((N + KS_PTS - 1) / KS_PTS) * 16 * 256; }

extern "C" long long u2_kmeans_shadow_floats(int N, int D) {
  if (N <= 0 || D % 32 != 0) return 0;
  // + |x_p - mu|, |(x_p - mu) - shadow_p / S|, |x_p| (each padded to whole tiles) + mu [D]
  // + aux [64]: the scale S, 1 / S, max |x_p - mu|^2
  return (long long)(km_shadow_words(N, D) + 3 * (size_t)((N + KS_PTS - 1) / KS_PTS) * KS_PTS + (size_t)((D + 63) & ~63) + 64);
}

extern "C" int u2_kmeans_prepare(const float* x, float* shadow, int N, int D, void* stream) {
  if (N <= 0 || D % 32 != 0 || !shadow) return -1;
  hipStream_t s = (hipStream_t)stream;
  const int G = ((N + KS_PTS - 1) / KS_PTS) * 16;
  float* mu = shadow + km_shadow_words(N, D) + 3 * (size_t)G * 16;
  // the column mean first; its slab sums use the (not yet written) shadow as scratch: slabs x D floats <= N D / 2
  const int slabs = (N + KM_MU_ROWS - 1) / KM_MU_ROWS;
  hipLaunchKernelGGL(km_colsum_kernel, dim3(slabs), dim3(256), 0, s, x, shadow, N, D);
  U2_CHECK_LAUNCH();
  hipLaunchKernelGGL(km_mu_kernel, dim3((D + 255) / 256), dim3(256), 0, s, (const float*)shadow, slabs, mu, N, D);
  U2_CHECK_LAUNCH();
  float* aux = mu + ((D + 63) & ~63);
  float* xn = shadow + km_shadow_words(N, D);
  u2_zero_words(reinterpret_cast<unsigned*>(aux), 4, s);
  const int xblocks = std::min((G * 16 + 3) / 4, 4096);
  hipLaunchKernelGGL(km_xnorm_kernel, dim3(xblocks), dim3(256), 0, s, x, (const float*)mu, xn, N, D, G * 16, 0, aux);
  U2_CHECK_LAUNCH();
  hipLaunchKernelGGL(km_scale_kernel, dim3(1), dim3(64), 0, s, aux);
  U2_CHECK_LAUNCH();
  const size_t threads = (size_t)(D >> 5) * G * 64;
  hipLaunchKernelGGL(km_shadow_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, x, (const float*)mu, reinterpret_cast<uint4*>(shadow),
                     N, D, G, (const float*)aux);
  U2_CHECK_LAUNCH();
  hipLaunchKernelGGL(km_xnorm_kernel, dim3(xblocks), dim3(256), 0, s, x, (const float*)mu, xn, N, D, G * 16, 1, aux);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_kmeans_assign(const float* x, const float* c, float* workspace, long long* labels, int N, int D, int K,
                                int exact_only, void* stream) {
  return u2_kmeans_assign_shadow(x, nullptr, c, workspace, labels, N, D, K, exact_only, stream);
}

extern "C" int u2_kmeans_assign_shadow(const float* x, const float* shadow, const float* c, float* workspace, long long* labels, int N,
                                       int D, int K, int exact_only, void* stream) {
  if (D % KM_BD != 0 || K < 1 || !workspace) return -1;
  if (N <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  float* cn = workspace;
  const bool screen = !exact_only && D % 32 == 0 && K <= KS_KMAX && K >= 2 && N >= KS_PTS;
  // K > 320 (the 800 clusters of u2seg_R50_800): the coarse pass once per block of 320 centroids over the shadow, the blocks' candidates
  // merged per point, what stays undecided straight to the exact kernel (13.6 -> 1.9 ms at N = 1 M, K = 800 on clustered data)
  const int blocks = (K + KS_KMAX - 1) / KS_KMAX;
  // centred centroids for the first pass over the shadow (csplit_kernel), behind everything else in the workspace
  unsigned* cs = reinterpret_cast<unsigned*>(workspace + km_ws_base_floats(N, D, K));
  float* cnc = reinterpret_cast<float*>(cs + 8);
  bf16_t* chc = reinterpret_cast<bf16_t*>(cnc + (size_t)blocks * KS_KMAX);
  const float* mu = shadow ? shadow + km_shadow_words(N, D) + 3 * (size_t)((N + KS_PTS - 1) / KS_PTS) * KS_PTS : nullptr;
  const float* aux = mu ? mu + ((D + 63) & ~63) : nullptr;    // {S, 1 / S, ...} of the shadow
  if (!exact_only && shadow && D % 32 == 0 && K > KS_KMAX && blocks <= KM_MAX_BLOCKS && N >= KS_PTS) {
    unsigned* scal = reinterpret_cast<unsigned*>(workspace + ((K + 3) & ~3));
    bf16_t* chl = reinterpret_cast<bf16_t*>(workspace + ((K + 3) & ~3) + 4);
    int* list2 = reinterpret_cast<int*>(workspace + ((K + 3) & ~3) + 4 + (size_t)blocks * KS_KMAX * D);
    float* pbase = reinterpret_cast<float*>(list2 + 2 * (size_t)N + 32);
    pbase += (2 - ((reinterpret_cast<size_t>(pbase) >> 2) & 1)) & 1;     // 8-byte aligned
    float2* part = reinterpret_cast<float2*>(pbase);
    u2_zero_words(scal, 4, s);
    u2_zero_words(cs, 4, s);
    for (int b = 0; b < blocks; ++b) {
      hipLaunchKernelGGL(csplit_kernel, dim3(KS_KMAX / 4), dim3(256), 0, s, c + (size_t)b * KS_KMAX * D, chl + (size_t)b * 2 * KS_KMAX * D,
                         cn + b * KS_KMAX, scal, D, std::min(KS_KMAX, K - b * KS_KMAX), mu, chc + (size_t)b * KS_KMAX * D, cnc + b * KS_KMAX,
                         cs, aux);
      U2_CHECK_LAUNCH();
    }
    static PerDeviceOnce attr_set;
    if (auto once_guard = attr_set.first())
      (void)hipFuncSetAttribute((const void*)kmeans_coarse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const unsigned tiles = (unsigned)((N + KS_PTS - 1) / KS_PTS), cus = (unsigned)km_cu_count();
    const float* xn = shadow + km_shadow_words(N, D);
    for (int b = 0; b < blocks; ++b) {
      hipLaunchKernelGGL(kmeans_coarse_kernel, dim3(tiles < cus ? tiles : cus), dim3(512), KC_LDS, s, reinterpret_cast<const unsigned char*>(shadow),
                         xn, chc + (size_t)b * KS_KMAX * D, cnc + b * KS_KMAX, cs, scal, labels, (int*)nullptr, (int*)nullptr, N, D,
                         std::min(KS_KMAX, K - b * KS_KMAX), 3e-4f, (const int*)nullptr, 0, part + (size_t)b * N, aux);
      U2_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(km_chunk_merge_kernel, dim3((N + 255) / 256), dim3(256), 0, s, part, blocks, xn, (int)(tiles * KS_PTS), cs, scal, labels, list2,
                       reinterpret_cast<int*>(scal + 1), N, 3e-4f);
    U2_CHECK_LAUNCH();
    hipLaunchKernelGGL(kmeans_assign_kernel, dim3((N + KM_PTS - 1) / KM_PTS), dim3(256), 0, s, x, c, cn, labels, N, D, K, (const int*)list2,
                       (const int*)(scal + 1), (const unsigned*)nullptr, (unsigned*)nullptr, 0u);
    U2_CHECK_LAUNCH();
    return 0;
  }
  if (!screen) {
    hipLaunchKernelGGL(cnorm_kernel, dim3((K + 3) / 4), dim3(256), 0, s, c, cn, D, K);
    U2_CHECK_LAUNCH();
    hipLaunchKernelGGL(kmeans_assign_kernel, dim3((N + KM_PTS - 1) / KM_PTS), dim3(256), 0, s, x, c, cn, labels, N, D, K,
                       (const int*)nullptr, (const int*)nullptr, (const unsigned*)nullptr, (unsigned*)nullptr, 0u);
    U2_CHECK_LAUNCH();
    return 0;
  }
  unsigned* scal = reinterpret_cast<unsigned*>(workspace + ((K + 3) & ~3));  // [0] max |c|^2 bits, [1] exact-list length, [2] coarse-list length
  bf16_t* chl = reinterpret_cast<bf16_t*>(workspace + ((K + 3) & ~3) + 4);
  int* list2 = reinterpret_cast<int*>(workspace + ((K + 3) & ~3) + 4 + (size_t)KS_KMAX * D);
  int* list1 = list2 + N;
  unsigned* state = reinterpret_cast<unsigned*>(list1 + N);
  state += (4 - ((reinterpret_cast<size_t>(state) >> 2) & 3)) & 3;   // 16-byte aligned (the slack is in the + 32 of the size)
  hipLaunchKernelGGL(km_state_begin_kernel, dim3(1), dim3(64), 0, s, scal, state, cs);
  U2_CHECK_LAUNCH();
  hipLaunchKernelGGL(csplit_kernel, dim3(KS_KMAX / 4), dim3(256), 0, s, c, chl, cn, scal, D, K, mu, chc, cnc, cs, aux);
  U2_CHECK_LAUNCH();
  static PerDeviceOnce attr_set;
  if (auto once_guard = attr_set.first()) {
    (void)hipFuncSetAttribute((const void*)kmeans_screen_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)kmeans_screen_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)kmeans_screen_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)kmeans_coarse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const dim3 grid((N + KS_PTS - 1) / KS_PTS), block(512);
  const size_t lds = KS_RING * KS_STAGE;                               // fine pass: both centroid planes, x through registers
  const size_t lds1 = U2_KM_X3 ? (size_t)KS_RING * KS_KMAX * 64 + 3 * KS_PTS * 128     // coarse pass: hi plane (20 KB stages) + three x slots
                               : (size_t)KS_RING * 3 * 8 * 1024 + 2 * KS_PTS * 128;   // round 5: 24 KB stages with filler rows + two x slots
  const size_t lds3 = 2 * KS_STAGE + 2 * KS_PTS * 128;                // fine pass with the x ring: two centroid stages + two x slots
  static const int xring3 = getenv("U2_KM_XRING3") ? atoi(getenv("U2_KM_XRING3")) : 1;   // measurement knob: 0 = x through registers
  const int* gate = reinterpret_cast<const int*>(state + 1);
  static const int two_level = getenv("U2_KM_ONE_LEVEL") ? 0 : 1;   // measurement knob: the round-3 single (fine) pass
  if (two_level) {
    // coarse pass over everything -> list1; fine pass over list1 -> list2 (both skipped while the coarse pass is switched off)
    if (shadow)
      hipLaunchKernelGGL(kmeans_coarse_kernel, dim3(grid.x < (unsigned)km_cu_count() ? grid.x : (unsigned)km_cu_count()), block, KC_LDS, s, reinterpret_cast<const unsigned char*>(shadow),
                         shadow + km_shadow_words(N, D), chc, cnc, cs, scal, labels, list1, reinterpret_cast<int*>(scal + 2), N, D, K, 3e-4f, gate, 0,
                         (float2*)nullptr, aux);
    else
      hipLaunchKernelGGL(kmeans_screen_kernel<1>, grid, block, lds1, s, x, chl, cn, scal, labels, list1, reinterpret_cast<int*>(scal + 2), N, D,
                         K, 0.02f, (const int*)nullptr, (const int*)nullptr, gate, 0);
    U2_CHECK_LAUNCH();
  }
  // fine pass -> list2: over the coarse pass's list, or over everything while the coarse pass is switched off (or not built in)
  {
    const int* rows = two_level ? (const int*)list1 : (const int*)nullptr;
    const int* nrows = two_level ? reinterpret_cast<const int*>(scal + 2) : (const int*)nullptr;
    if (xring3)
      hipLaunchKernelGGL((kmeans_screen_kernel<3, true>), grid, block, lds3, s, x, chl, cn, scal, labels, list2, reinterpret_cast<int*>(scal + 1), N,
                         D, K, 1e-4f, rows, nrows, two_level ? gate : (const int*)nullptr, 2);
    else
      hipLaunchKernelGGL(kmeans_screen_kernel<3>, grid, block, lds, s, x, chl, cn, scal, labels, list2, reinterpret_cast<int*>(scal + 1), N, D, K,
                         1e-4f, rows, nrows, two_level ? gate : (const int*)nullptr, 2);
    U2_CHECK_LAUNCH();
  }
  // The undecided points, exactly; the grid covers the worst case, work-groups beyond the list return immediately.  Its first thread
  // also keeps the screening state.  The coarse pass pays while it costs less than the fine pass saves on the points it decides: over
  // the fp32 x it costs half a fine pass (off above a quarter undecided), over the shadow a third (0.45 vs 1.38 ms at N = 1 M; the
  // fine pass over a list is ~10 % dearer per point): off above 55 %.
  const unsigned limit = shadow ? (unsigned)((unsigned long long)N * 55u / 100u) : (unsigned)N / 4u;
  hipLaunchKernelGGL(kmeans_assign_kernel, dim3((N + KM_PTS - 1) / KM_PTS), dim3(256), 0, s, x, c, cn, labels, N, D, K,
                     (const int*)list2, (const int*)(scal + 1), (const unsigned*)scal, two_level ? state : (unsigned*)nullptr, limit);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" long long u2_kmeans_update_workspace_floats(int N, int D, int K) {
  (void)D;
  return 2LL * N + 2LL * K + 16;  // int32 order[N], lab_sorted[N], counts[K], cursor[K]
}

extern "C" int u2_kmeans_update(const float* x, const long long* labels, float* csum, float* counts, int N, int D, int K,
                                float* workspace, void* stream) {
  if (N <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (workspace && D <= 256 * 8 && (size_t)K * sizeof(int) <= 64 * 1024) {  // the [K] tables of the bucketing kernels live in LDS
    int* order = reinterpret_cast<int*>(workspace);
    int* lab_sorted = order + N;
    int* icounts = lab_sorted + N;
    int* cursor = icounts + K;
    u2_zero_words(icounts, (size_t)K, s);
    U2_CHECK_LAUNCH();
    hipLaunchKernelGGL(km_hist_kernel, dim3(256), dim3(256), (size_t)K * sizeof(int), s, labels, icounts, N, K);
    hipLaunchKernelGGL(km_scan_kernel, dim3(1), dim3(256), 0, s, icounts, cursor, counts, K);
    hipLaunchKernelGGL(km_scatter_kernel, dim3((N + KM_SCAT - 1) / KM_SCAT), dim3(256), (size_t)K * sizeof(int), s, labels, cursor, order,
                       lab_sorted, N, K);
    const dim3 grid((N + KM_SEG - 1) / KM_SEG);
    const int dpt = (D + 255) / 256;
    if (dpt <= 1) hipLaunchKernelGGL(km_segsum_kernel<1>, grid, dim3(256), 0, s, x, order, lab_sorted, csum, N, D, cursor + K - 1);
    else if (dpt <= 2) hipLaunchKernelGGL(km_segsum_kernel<2>, grid, dim3(256), 0, s, x, order, lab_sorted, csum, N, D, cursor + K - 1);
    else if (dpt <= 3) hipLaunchKernelGGL(km_segsum_kernel<3>, grid, dim3(256), 0, s, x, order, lab_sorted, csum, N, D, cursor + K - 1);
    else if (dpt <= 4) hipLaunchKernelGGL(km_segsum_kernel<4>, grid, dim3(256), 0, s, x, order, lab_sorted, csum, N, D, cursor + K - 1);
    else hipLaunchKernelGGL(km_segsum_kernel<8>, grid, dim3(256), 0, s, x, order, lab_sorted, csum, N, D, cursor + K - 1);
    U2_CHECK_LAUNCH();
    return 0;
  }
  int DS = 64;
  while (DS > 16 && (size_t)K * (DS + 1) * 4 > 144 * 1024) DS >>= 1;
  if ((size_t)K * (DS + 1) * 4 > 144 * 1024) return -1;
  const size_t lds = (size_t)K * (DS + 1) * 4;
  const int ppb = 8192;
  const dim3 grid((N + ppb - 1) / ppb, (D + DS - 1) / DS);
  hipError_t e;
  if (DS == 64) {
    e = hipFuncSetAttribute((const void*)kmeans_update_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kmeans_update_kernel<64>, grid, dim3(256), lds, s, x, labels, csum, counts, N, D, K, ppb);
  } else if (DS == 32) {
    e = hipFuncSetAttribute((const void*)kmeans_update_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kmeans_update_kernel<32>, grid, dim3(256), lds, s, x, labels, csum, counts, N, D, K, ppb);
  } else {
    e = hipFuncSetAttribute((const void*)kmeans_update_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kmeans_update_kernel<16>, grid, dim3(256), lds, s, x, labels, csum, counts, N, D, K, ppb);
  }
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_kmeans_finalize(const float* csum, const float* counts, float* c, int D, int K, void* stream) {
  const size_t n = (size_t)K * D;
  hipLaunchKernelGGL(kmeans_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, csum,
                     counts, c, D, K);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_abi_version(void) { return 1; }
