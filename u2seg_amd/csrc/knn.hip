// Brute-force k nearest neighbours over fp32 embeddings (first-order density estimate before k-means).
//
// Replaces kNN / partitioned_kNN in u2seg/Instance_Clustering/shared/utils/nn_utils.py:204-299:
//   D_ij = sum_d (x_test_id - x_train_jd)^2;  d_knn, ind_knn = D_ij.Kmin_argKmin(K, dim=1)       (:210-218, pykeops)
//   the reference tiles both sides in partitions of 130 000 rows to fit its GPU and merges the per-partition lists by
//   an argsort (:226-266); the merged result is the global K smallest per row (its own verify branch asserts that,
//   :268-287).  With 288 GB of HBM the whole train set stays resident and one pass produces the global lists.
//
// Two kernels:
//   select  for 128 query rows per workgroup, stream every train row through LDS; dot products on the exact-fp32 MFMA
//           (v_mfma_f32_32x32x2_f32, an fma chain in k order); rank key |y_j|^2 - 2 x_i.y_j (|x_i|^2 is constant per
//           row), evaluated on rows translated by the train set's column mean (distances do not change, the cancellation
//           in the expanded form shrinks from |x|^2 to the spread of the data; the translation happens while staging).
//           A per-row threshold (current KK-th smallest) filters the tile; survivors go through a per-row LDS queue into
//           a per-row sorted list of KK = K + 4 candidates.
//   refine  one wave per query row: recomputes sum_d (x - y)^2 for the KK candidates in the reference's difference form
//           (no cancellation; the row's own distance is exactly 0), ranks them by (distance, index) and writes the first K.
// The 4 spare candidates absorb the rounding difference between the two formulations at the K-th / (K+1)-th boundary.
#include <limits.h>

#include <type_traits>

#include "common.h"
#include "u2seg_hip.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int KN_ROWS = 128;   // query rows per workgroup (32 per wave)
constexpr int KN_BD = 16;      // dims per staged chunk
constexpr int KN_PITCH = 17;   // padded LDS row pitch (floats)
constexpr int KN_TILES = 10;   // 32-row train tiles per pass (320 train rows)
constexpr int KN_KMAX = 32;    // candidate list capacity
constexpr int KN_LP = 33;      // padded pitch of the per-row lists
constexpr int KN_SPARE = 4;

constexpr int KN_STAGE = (KN_ROWS + KN_TILES * 32) * KN_PITCH;  // floats of one staging buffer (query chunk + train chunk)
constexpr size_t KN_LDS_FLOATS = 2 * (size_t)KN_STAGE                                    // two staging buffers
                                 + 2 * KN_ROWS * KN_LP                                   // topv, topj
                                 + 2 * KN_ROWS * 32                                      // qv, qj
                                 + 2 * KN_ROWS;                                          // qcnt, thr
constexpr size_t KN_LDS_BYTES = KN_LDS_FLOATS * 4;

constexpr int KN_MEAN_SLICES = 64;

// partial[s][d] = sum over the rows of slice s of y[.][d]  (fixed order: deterministic)
__global__ __launch_bounds__(256) void knn_colsum_kernel(const float* __restrict__ y, float* __restrict__ partial, int D, int N) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int d = blockIdx.x * 64 + lane;
  const int per = (N + KN_MEAN_SLICES - 1) / KN_MEAN_SLICES;
  const int r0 = blockIdx.y * per, r1 = min(N, r0 + per);
  float s = 0.f;
  if (d < D)
    for (int r = r0 + w; r < r1; r += 4) s += y[(size_t)r * D + d];
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && d < D) partial[(size_t)blockIdx.y * D + d] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

__global__ void knn_colmean_kernel(const float* __restrict__ partial, float* __restrict__ mu, int D, int N) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  float s = 0.f;
  for (int i = 0; i < KN_MEAN_SLICES; ++i) s += partial[(size_t)i * D + d];
  mu[d] = s / (float)N;
}

__global__ __launch_bounds__(256) void knn_rownorm_kernel(const float* __restrict__ y, const float* __restrict__ mu,
                                                          float* __restrict__ yn, int D, int N) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= N) return;
  float s = 0.f;
  for (int d = threadIdx.x & 63; d < D; d += 64) { const float v = y[(size_t)j * D + d] - mu[d]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) yn[j] = s;
}

__device__ __forceinline__ bool kn_less(float v, int j, float bv, int bj) { return v < bv || (v == bv && j < bj); }

__global__ __launch_bounds__(256, 1) void knn_select_kernel(const float* __restrict__ xq, const float* __restrict__ xt,
                                                            const float* __restrict__ tn, const float* __restrict__ mu,
                                                            int* __restrict__ cand, int Nq, int Nt, int D, int KK) {
  extern __shared__ float kn_lds[];
  float* stage = kn_lds;  // [2][xs: KN_ROWS x KN_PITCH | cs: 320 x KN_PITCH]
  float* topv = stage + 2 * KN_STAGE;
  int* topj = reinterpret_cast<int*>(topv + KN_ROWS * KN_LP);
  float* qv = reinterpret_cast<float*>(topj + KN_ROWS * KN_LP);
  int* qj = reinterpret_cast<int*>(qv + KN_ROWS * 32);
  int* qcnt = qj + KN_ROWS * 32;
  float* thr = reinterpret_cast<float*>(qcnt + KN_ROWS);

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int p0 = blockIdx.x * KN_ROWS;
  const int li = lane & 31, lk = lane >> 5;

  for (int i = tid; i < KN_ROWS * KN_LP; i += 256) { topv[i] = INFINITY; topj[i] = INT_MAX; }
  if (tid < KN_ROWS) { qcnt[tid] = 0; thr[tid] = INFINITY; }
  float thr_reg[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) thr_reg[r] = INFINITY;

  for (int k0 = 0; k0 < Nt; k0 += KN_TILES * 32) {
    const int ntile = min(KN_TILES, (Nt - k0 + 31) / 32);
    f32x16 acc[KN_TILES];
#pragma unroll
    for (int t = 0; t < KN_TILES; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // Staging registers for one 16-dim chunk: 128 query rows + 320 train rows x 4 float4 per row over 256 threads.  Rows
    // past the end are clamped to the last row instead of predicated (they are never ranked): the loads stay straight-line
    // code, so the compiler can count them (s_waitcnt vmcnt(N)) instead of draining everything at a join.
    struct Chunk { float4 xr[2], cr[5], m; };
    const int c4 = tid & 3;  // every float4 index f = q * 256 + tid below has f % 4 == tid % 4
    const float* xrow[2];
    const float* crow[5];
#pragma unroll
    for (int q = 0; q < 2; ++q) xrow[q] = xq + (size_t)min(p0 + ((q * 256 + tid) >> 2), Nq - 1) * D + c4 * 4;
#pragma unroll
    for (int q = 0; q < 5; ++q) crow[q] = xt + (size_t)min(k0 + ((q * 256 + tid) >> 2), Nt - 1) * D + c4 * 4;
    auto gload = [&](Chunk& R, int d0) {
      R.m = *reinterpret_cast<const float4*>(mu + d0 + c4 * 4);
#pragma unroll
      for (int q = 0; q < 2; ++q) R.xr[q] = *reinterpret_cast<const float4*>(xrow[q] + d0);
#pragma unroll
      for (int q = 0; q < 5; ++q) R.cr[q] = *reinterpret_cast<const float4*>(crow[q] + d0);
    };
    auto lstore = [&](const Chunk& R, int buf) {
      float* xs = stage + buf * KN_STAGE;
      float* cs = xs + KN_ROWS * KN_PITCH;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int f = q * 256 + tid;
        float* dst = xs + (f >> 2) * KN_PITCH + (f & 3) * 4;  // translated by the column mean here, not at load time:
        dst[0] = R.xr[q].x - R.m.x; dst[1] = R.xr[q].y - R.m.y;  // the loads stay in flight across the MFMA sequence
        dst[2] = R.xr[q].z - R.m.z; dst[3] = R.xr[q].w - R.m.w;
      }
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const int f = q * 256 + tid;
        float* dst = cs + (f >> 2) * KN_PITCH + (f & 3) * 4;
        dst[0] = R.cr[q].x - R.m.x; dst[1] = R.cr[q].y - R.m.y; dst[2] = R.cr[q].z - R.m.z; dst[3] = R.cr[q].w - R.m.w;
      }
    };

    // One barrier per 16-dim chunk and a global prefetch one full chunk ahead: late in the MFMA sequence of chunk c the
    // registers holding chunk c + 1 (requested at the same point of chunk c - 1) move into the other LDS buffer (its
    // readers finished before the previous barrier) and are immediately re-used to request chunk c + 2.  The LDS operands
    // of step ks + 1 are read while the 10 MFMAs of step ks run; full passes (all 10 train tiles present) carry no per-tile
    // branch.
    auto dloop = [&](auto full) {
      constexpr bool FULL = decltype(full)::value;
      Chunk R;
      gload(R, 0);
      lstore(R, 0);
      if (KN_BD < D) gload(R, KN_BD);
      __syncthreads();
      int buf = 0;
      for (int d0 = 0; d0 < D; d0 += KN_BD, buf ^= 1) {
        const bool more = d0 + KN_BD < D;
        const float* xa = stage + buf * KN_STAGE + (w * 32 + li) * KN_PITCH + lk;
        const float* cb = stage + buf * KN_STAGE + KN_ROWS * KN_PITCH + li * KN_PITCH + lk;
        float a = xa[0], b[KN_TILES];
#pragma unroll
        for (int t = 0; t < KN_TILES; ++t) b[t] = (FULL || t < ntile) ? cb[t * 32 * KN_PITCH] : 0.f;
#pragma unroll
        for (int ks = 0; ks < KN_BD / 2; ++ks) {
          float an = 0.f, bn[KN_TILES];
          if (ks + 1 < KN_BD / 2) {
            an = xa[(ks + 1) * 2];
#pragma unroll
            for (int t = 0; t < KN_TILES; ++t) bn[t] = (FULL || t < ntile) ? cb[t * 32 * KN_PITCH + (ks + 1) * 2] : 0.f;
          }
          if (ks == KN_BD / 2 - 3 && more) {
            lstore(R, buf ^ 1);
            if (d0 + 2 * KN_BD < D) gload(R, d0 + 2 * KN_BD);
          }
#pragma unroll
          for (int t = 0; t < KN_TILES; ++t)
            if (FULL || t < ntile) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
          if (ks + 1 < KN_BD / 2) {
            a = an;
#pragma unroll
            for (int t = 0; t < KN_TILES; ++t) b[t] = bn[t];
          }
        }
        __syncthreads();
      }
    };
    if (ntile == KN_TILES) dloop(std::true_type{}); else dloop(std::false_type{});
    // D[i = query][j = train]: lane holds column j = li of tile t, rows (r&3) + 8*(r>>2) + 4*lk of its wave's 32
#pragma unroll
    for (int t = 0; t < KN_TILES; ++t) {
      if (t < ntile) {
        const int j = k0 + t * 32 + li;
        const float tj = (j < Nt) ? tn[j] : 0.f;
        int any = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = tj - 2.f * acc[t][r];
          if (j < Nt && v < thr_reg[r]) {
            const int row = w * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            const int slot = atomicAdd(&qcnt[row], 1);  // at most 32 per row and tile: one per column
            qv[row * 32 + slot] = v;
            qj[row * 32 + slot] = j;
            any = 1;
          }
        }
        if (__syncthreads_or(any)) {
          if (tid < KN_ROWS) {
            const int n = qcnt[tid];
            float* tv = topv + tid * KN_LP;
            int* tj2 = topj + tid * KN_LP;
            float th = tv[KK - 1];
            int thj = tj2[KK - 1];
            for (int c = 0; c < n; ++c) {
              const float v = qv[tid * 32 + c];
              const int jj = qj[tid * 32 + c];
              if (!kn_less(v, jj, th, thj)) continue;
              int pos = KK - 1;
              while (pos > 0) {
                const float pv = tv[pos - 1];
                const int pj = tj2[pos - 1];
                if (!kn_less(v, jj, pv, pj)) break;
                tv[pos] = pv;
                tj2[pos] = pj;
                --pos;
              }
              tv[pos] = v;
              tj2[pos] = jj;
              th = tv[KK - 1];
              thj = tj2[KK - 1];
            }
            qcnt[tid] = 0;
            thr[tid] = th;
          }
          __syncthreads();
#pragma unroll
          for (int r = 0; r < 16; ++r) thr_reg[r] = thr[w * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk];
        }
      }
    }
    __syncthreads();
  }
  for (int i = tid; i < KN_ROWS * KK; i += 256) {
    const int row = i / KK, k = i % KK;
    if (p0 + row < Nq) cand[(size_t)(p0 + row) * KK + k] = topj[row * KN_LP + k];
  }
}

// One wave per query row: exact difference-form distances of its KK candidates, rank by (distance, index), keep K.
__global__ __launch_bounds__(256) void knn_refine_kernel(const float* __restrict__ xq, const float* __restrict__ xt,
                                                         const int* __restrict__ cand, float* __restrict__ d_knn,
                                                         long long* __restrict__ ind_knn, int Nq, int D, int KK, int K) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= Nq) return;
  const float* xi = xq + (size_t)row * D;
  float my_d = INFINITY;
  int my_j = INT_MAX;
  for (int k = 0; k < KK; ++k) {
    const int j = cand[(size_t)row * KK + k];
    float s = INFINITY;
    if (j != INT_MAX) {
      const float* yj = xt + (size_t)j * D;
      s = 0.f;
      for (int d = lane; d < D; d += 64) { const float df = xi[d] - yj[d]; s += df * df; }
      s = wave_sum(s);
    }
    if (lane == k) { my_d = s; my_j = j; }
  }
  int rank = 0;
  for (int k = 0; k < KK; ++k) {
    const float od = __shfl(my_d, k, 64);
    const int oj = __shfl(my_j, k, 64);
    rank += kn_less(od, oj, my_d, my_j) ? 1 : 0;
  }
  if (lane < KK && rank < K && my_j != INT_MAX) {
    d_knn[(size_t)row * K + rank] = my_d;
    ind_knn[(size_t)row * K + rank] = (long long)my_j;
  }
}

}  // namespace

extern "C" int u2_knn_workspace_ints(int Nq, int Nt, int D, int K, long long* n_ints) {
  if (K < 1 || K + KN_SPARE > KN_KMAX || D < 1 || !n_ints) return -1;
  const int KK = K + KN_SPARE;
  // |y_j - mu|^2 (fp32), candidate lists (int32), column mean and its partial sums (fp32)
  *n_ints = (long long)Nt + (long long)Nq * KK + (long long)(KN_MEAN_SLICES + 1) * D;
  return 0;
}

extern "C" int u2_knn(const float* x_query, const float* x_train, void* workspace, float* d_knn, long long* ind_knn, int Nq,
                      int Nt, int D, int K, void* stream) {
  if (K < 1 || K + KN_SPARE > KN_KMAX || D < KN_BD || D % KN_BD != 0 || !workspace) return -1;
  if (Nt < K) return -2;  // the reference assumes every partition holds at least K rows (nn_utils.py:236)
  if (Nq <= 0) return 0;
  const int KK = K + KN_SPARE;
  hipStream_t s = (hipStream_t)stream;
  float* tn = reinterpret_cast<float*>(workspace);
  int* cand = reinterpret_cast<int*>(workspace) + Nt;
  float* mu = reinterpret_cast<float*>(cand + (size_t)Nq * KK);
  float* partial = mu + D;
  hipLaunchKernelGGL(knn_colsum_kernel, dim3((D + 63) / 64, KN_MEAN_SLICES), dim3(256), 0, s, x_train, partial, D, Nt);
  U2_CHECK_LAUNCH();
  hipLaunchKernelGGL(knn_colmean_kernel, dim3((D + 255) / 256), dim3(256), 0, s, partial, mu, D, Nt);
  U2_CHECK_LAUNCH();
  hipLaunchKernelGGL(knn_rownorm_kernel, dim3((Nt + 3) / 4), dim3(256), 0, s, x_train, mu, tn, D, Nt);
  U2_CHECK_LAUNCH();
  hipError_t e = hipFuncSetAttribute((const void*)knn_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)KN_LDS_BYTES);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(knn_select_kernel, dim3((Nq + KN_ROWS - 1) / KN_ROWS), dim3(256), KN_LDS_BYTES, s, x_query, x_train, tn,
                     mu, cand, Nq, Nt, D, KK);
  U2_CHECK_LAUNCH();
  hipLaunchKernelGGL(knn_refine_kernel, dim3((Nq + 3) / 4), dim3(256), 0, s, x_query, x_train, cand, d_knn, ind_knn, Nq, D, KK,
                     K);
  U2_CHECK_LAUNCH();
  return 0;
}
