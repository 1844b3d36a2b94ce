// Lloyd k-means over fp32 embeddings: assignment (pairwise distance + argmin) and update (segmented sums).
//
// Replaces KMeans in u2seg/Instance_Clustering/shared/utils/nn_utils.py:304-379:
//   E-step  cl = argmin_j sum_d (x_id - c_jd)^2      (:353-355, a pykeops LazyTensor reduction)
//   M-step  c.zero_(); c.scatter_add_(0, cl.repeat(1,D), x); Ncl = bincount(cl); c /= Ncl   (:358-364)
// E-step here: dist_j = |c_j|^2 - 2 x.c_j (the |x|^2 term does not change the argmin) with the dot products on
// the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, an fma chain in k order), so labels agree with the reference
// except where two centroids are equidistant to within fp32 rounding of the two formulations.
// Empty clusters give 0/0 = NaN centroids exactly like the reference (noted at usl-imagenet.py:135).
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

__global__ __launch_bounds__(256, 1) void kmeans_assign_kernel(const float* __restrict__ x, const float* __restrict__ c,
                                                               const float* __restrict__ cn, long long* __restrict__ labels,
                                                               int N, int D, int K) {
  __shared__ float stage[2 * KM_STAGE];  // [2][xs: KM_PTS x KM_PITCH | cs: 320 x KM_PITCH]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int p0 = blockIdx.x * KM_PTS;
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
    for (int q = 0; q < 2; ++q) xrow[q] = x + (size_t)min(p0 + ((q * 256 + tid) >> 2), N - 1) * D + c4 * 4;
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
      if (p < N) labels[p] = (long long)j;
    }
  }
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
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)N; i += (size_t)gridDim.x * 256) atomicAdd(&hist[(int)labels[i]], 1);
  __syncthreads();
  for (int j = threadIdx.x; j < K; j += 256)
    if (hist[j]) atomicAdd(counts + j, hist[j]);
}

__global__ __launch_bounds__(256) void km_scan_kernel(const int* __restrict__ counts, int* __restrict__ cursor,
                                                      float* __restrict__ fcounts, int K) {
  __shared__ int run;
  if (threadIdx.x == 0) {  // K is a few hundred: a serial scan costs nothing next to the streaming passes
    int acc = 0;
    for (int j = 0; j < K; ++j) { cursor[j] = acc; acc += counts[j]; }
    run = acc;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < K; j += 256) fcounts[j] += (float)counts[j];
}

__global__ __launch_bounds__(256) void km_scatter_kernel(const long long* __restrict__ labels, int* __restrict__ cursor,
                                                         int* __restrict__ order, int* __restrict__ lab_sorted, int N) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)N) return;
  const int l = (int)labels[i];
  const int pos = atomicAdd(cursor + l, 1);
  order[pos] = (int)i;
  lab_sorted[pos] = l;
}

constexpr int KM_SEG = 256;  // label-ordered entries per work-group
template <int DPT>           // dimensions per thread: D <= 256 * DPT
__global__ __launch_bounds__(256) void km_segsum_kernel(const float* __restrict__ x, const int* __restrict__ order,
                                                        const int* __restrict__ lab_sorted, float* __restrict__ csum, int N, int D) {
  const int tid = threadIdx.x;
  const int e0 = blockIdx.x * KM_SEG, e1 = min(N, e0 + KM_SEG);
  if (e0 >= e1) return;
  float acc[DPT];
#pragma unroll
  for (int t = 0; t < DPT; ++t) acc[t] = 0.f;
  int cur = lab_sorted[e0];
  auto flush = [&](int label) {
#pragma unroll
    for (int t = 0; t < DPT; ++t) {
      const int d = tid + t * 256;
      if (d < D && acc[t] != 0.f) atomicAdd(csum + (size_t)label * D + d, acc[t]);
      acc[t] = 0.f;
    }
  };
  constexpr int UNR = 8;  // rows in flight per thread
  for (int e = e0; e < e1; e += UNR) {
    float v[UNR][DPT];
    int lab[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const bool ok = e + u < e1;
      lab[u] = ok ? lab_sorted[e + u] : -1;
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

extern "C" int u2_kmeans_assign(const float* x, const float* c, float* cnorm_ws, long long* labels, int N, int D, int K,
                                void* stream) {
  if (D % KM_BD != 0 || K < 1 || !cnorm_ws) return -1;
  if (N <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(cnorm_kernel, dim3((K + 3) / 4), dim3(256), 0, s, c, cnorm_ws, D, K);
  U2_CHECK_LAUNCH();
  hipLaunchKernelGGL(kmeans_assign_kernel, dim3((N + KM_PTS - 1) / KM_PTS), dim3(256), 0, s, x, c, cnorm_ws, labels, N, D, K);
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
  if (workspace && D <= 256 * 8) {
    int* order = reinterpret_cast<int*>(workspace);
    int* lab_sorted = order + N;
    int* icounts = lab_sorted + N;
    int* cursor = icounts + K;
    hipError_t e = hipMemsetAsync(icounts, 0, (size_t)K * sizeof(int), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(km_hist_kernel, dim3(1024), dim3(256), (size_t)K * sizeof(int), s, labels, icounts, N, K);
    hipLaunchKernelGGL(km_scan_kernel, dim3(1), dim3(256), 0, s, icounts, cursor, counts, K);
    hipLaunchKernelGGL(km_scatter_kernel, dim3((N + 255) / 256), dim3(256), 0, s, labels, cursor, order, lab_sorted, N);
    const dim3 grid((N + KM_SEG - 1) / KM_SEG);
    const int dpt = (D + 255) / 256;
    if (dpt <= 1) hipLaunchKernelGGL(km_segsum_kernel<1>, grid, dim3(256), 0, s, x, order, lab_sorted, csum, N, D);
    else if (dpt <= 2) hipLaunchKernelGGL(km_segsum_kernel<2>, grid, dim3(256), 0, s, x, order, lab_sorted, csum, N, D);
    else if (dpt <= 3) hipLaunchKernelGGL(km_segsum_kernel<3>, grid, dim3(256), 0, s, x, order, lab_sorted, csum, N, D);
    else if (dpt <= 4) hipLaunchKernelGGL(km_segsum_kernel<4>, grid, dim3(256), 0, s, x, order, lab_sorted, csum, N, D);
    else hipLaunchKernelGGL(km_segsum_kernel<8>, grid, dim3(256), 0, s, x, order, lab_sorted, csum, N, D);
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
