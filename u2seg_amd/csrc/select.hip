// Segmented top-k / selection with a total order, for the proposal bookkeeping of the RPN and the ROI heads.
//
// Replaces the torch.topk / torch.sort calls of detectron2/modeling/proposal_generator/proposal_utils.py:79-96 (per-level
// pre-NMS top-k on the objectness logits, then the score sort in front of batched_nms) and the randperm-based subsampling of
// detectron2/modeling/sampling.py:38-54 in its "k smallest random keys" form.  torch.topk leaves the order of equal scores
// unspecified (and bf16 objectness logits tie constantly); here every row is ranked by the total order
//     (value descending [or ascending], index ascending),
// which is what a stable sort gives and what the CPU oracle uses, so index lists are bit-exact and reproducible.
//
// One work-group (1024 threads) per row.  The k-th element of the 64-bit key (ordered value bits : ~index) is found by
// most-significant-digit radix selection, 8 bits per sweep with an LDS histogram: 2 sweeps for bf16 values, 4 for fp32,
// plus up to 3 over the index bits only when the k-th value is tied; the survivors are compacted into LDS, bitonic-sorted
// there and written out.  Rows are short (<= 268 569 elements, a few hundred KB): each sweep is an L2-resident stream.
#include "common.h"
#include "u2seg_hip.h"

namespace {

constexpr int SEL_THREADS = 1024;

struct SelectArgs {
  const void* vals;
  int dtype;              // 0: fp32, 1: bf16
  int rows, n;
  long long row_stride;   // elements between consecutive rows
  int group, pitch;       // element i of a row lives at (i / group) * pitch + i % group   (group = pitch = 1: contiguous)
  const signed char* mask;  // optional [rows][n]: only elements with mask == mask_value take part
  int mask_value;
  int k, largest;
  float* out_vals;        // [rows][k]
  int* out_idx;           // [rows][k]
  const int* idx_in;      // optional [rows][n]: the index an element stands for (tie-break and output) instead of its position
  int* out_cnt;           // [rows] (optional): number of real entries (min(k, participating elements))
  int cap;                // LDS slots for the survivors (power of two >= k)
};

// ordered key: larger = ranked earlier.  fp32 keeps all 32 bits; bf16 is ordered on its 16 bits and shifted up, so that the
// low 16 key bits are zero for every element (the selection then skips them).  -0.0 ranks equal to +0.0, as in a sort.
__device__ __forceinline__ uint32_t order_bits(uint32_t u, int largest, int is_bf16) {
  if (is_bf16) {
    uint32_t h = u & 0xffffu;
    if (h == 0x8000u) h = 0;
    const uint32_t asc = (h & 0x8000u) ? (~h & 0xffffu) : (h | 0x8000u);
    return (largest ? asc : (~asc & 0xffffu)) << 16;
  }
  if (u == 0x80000000u) u = 0;
  const uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return largest ? asc : ~asc;
}
__device__ __forceinline__ float unorder_bits(uint32_t key, int largest, int is_bf16) {
  if (is_bf16) {
    const uint32_t asc = largest ? (key >> 16) : (~(key >> 16) & 0xffffu);
    const uint32_t h = (asc & 0x8000u) ? (asc & 0x7fffu) : (~asc & 0xffffu);
    return __uint_as_float(h << 16);
  }
  const uint32_t asc = largest ? key : ~key;
  const uint32_t u = (asc & 0x80000000u) ? (asc & 0x7fffffffu) : ~asc;
  return __uint_as_float(u);
}

__global__ __launch_bounds__(SEL_THREADS) void topk_rows_kernel(const SelectArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* sel = reinterpret_cast<unsigned long long*>(smem_raw);            // [cap]
  unsigned int* hist = reinterpret_cast<unsigned int*>(smem_raw + (size_t)a.cap * 8);   // [256] + scratch [8]
  const int tid = threadIdx.x;
  const int row = blockIdx.x;
  const int n = a.n;
  const float inv_group = 1.0f / (float)a.group;
  const signed char* mrow = a.mask ? a.mask + (size_t)row * n : nullptr;
  const int* irow = a.idx_in ? a.idx_in + (size_t)row * n : nullptr;

  auto load_key = [&](int i, bool& ok) -> uint32_t {
    ok = !mrow || mrow[i] == (signed char)a.mask_value;
    size_t off = (size_t)i;
    if (a.group != 1 || a.pitch != 1) {
      int q = (int)((float)i * inv_group);
      int r = i - q * a.group;
      if (r < 0) { --q; r += a.group; }
      if (r >= a.group) { ++q; r -= a.group; }
      off = (size_t)q * a.pitch + r;
    }
    uint32_t u;
    if (a.dtype == 1) u = (uint32_t)reinterpret_cast<const bf16_t*>(a.vals)[(size_t)row * a.row_stride + off];
    else u = reinterpret_cast<const uint32_t*>(a.vals)[(size_t)row * a.row_stride + off];
    return order_bits(u, a.largest, a.dtype);
  };

  // one sweep over the participating elements of the row, eight independent loads per thread in flight (a row is streamed by a
  // single work-group: the sweep is latency-bound, not bandwidth-bound)
  auto for_each_key = [&](auto&& fn) {
    constexpr int UNR = 8;
    for (int base = 0; base < n; base += UNR * SEL_THREADS) {
      uint32_t vk[UNR];
      bool ok[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int i = base + u * SEL_THREADS + tid;
        ok[u] = false;
        vk[u] = 0;
        if (i < n) vk[u] = load_key(i, ok[u]);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u)
        if (ok[u]) fn(base + u * SEL_THREADS + tid, vk[u]);
    }
  };

  // ---- most-significant-digit radix selection of the k-th largest 64-bit key (value key : ~index).  `prefix` holds the
  // `fixed` high bits decided so far, `want` how many elements of the bucket that shares them are still to be taken.
  // Key bits 63..32 = value (bf16: bits 47..32 are zero for every element), 31..24 = 0xff (index < 2^24), 23..0 = ~index.
  unsigned long long prefix = 0;
  int fixed = 0;
  int want = a.k;
  bool take_all = false;  // no more than k elements take part: all of them are selected
  const int value_sweeps = a.dtype == 1 ? 2 : 4;
  for (int sweep = 0; sweep < value_sweeps + 3; ++sweep) {
    const int shift = sweep < value_sweeps ? 56 - 8 * sweep : 16 - 8 * (sweep - value_sweeps);
    for (int b = tid; b < 256; b += SEL_THREADS) hist[b] = 0;
    __syncthreads();
    const unsigned long long hi_mask = fixed == 0 ? 0ull : (~0ull << (64 - fixed));
    for_each_key([&](int i, uint32_t vk) {
      const unsigned long long key = ((unsigned long long)vk << 32) | (uint32_t)(0xffffffffu - (uint32_t)(irow ? irow[i] : i));
      if ((key & hi_mask) == prefix) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
    });
    __syncthreads();
    if (tid == 0) {
      unsigned total = 0;
      for (int b = 0; b < 256; ++b) total += hist[b];
      unsigned above = 0;
      int digit = 255;
      const bool all = sweep == 0 && total <= (unsigned)want;
      if (!all) {
        for (; digit > 0; --digit) {
          if (above + hist[digit] >= (unsigned)want) break;
          above += hist[digit];
        }
      }
      hist[256] = (unsigned)digit;
      hist[257] = above;          // elements of the bucket ranked above the chosen digit: all taken
      hist[258] = all ? 1u : 0u;
      hist[259] = hist[digit];    // size of the chosen digit's sub-bucket
      hist[260] = total;
    }
    __syncthreads();
    if (hist[258]) { take_all = true; break; }
    const int sub = (int)hist[259];
    want -= (int)hist[257];
    prefix |= (unsigned long long)hist[256] << shift;
    if (sweep + 1 < value_sweeps) {
      fixed = 8 * (sweep + 1);
    } else if (sweep + 1 == value_sweeps) {
      fixed = 40;                 // the whole value and the constant top byte of ~index
      prefix |= 0xffull << 24;
    } else {
      fixed = 40 + 8 * (sweep + 1 - value_sweeps);
    }
    if (sub == want) break;       // the sub-bucket is taken whole: nothing left to separate
  }

  // ---- compaction: everything >= the threshold key (or everything, if fewer than k take part) ----
  const unsigned long long hi_mask = fixed == 0 ? 0ull : (~0ull << (64 - fixed));
  for (int i = tid; i < a.cap; i += SEL_THREADS) sel[i] = 0ull;
  if (tid == 0) hist[261] = 0;
  __syncthreads();
  for_each_key([&](int i, uint32_t vk) {
    const unsigned long long key = ((unsigned long long)vk << 32) | (uint32_t)(0xffffffffu - (uint32_t)(irow ? irow[i] : i));
    // with `fixed` high bits decided, an element is taken iff its high bits are >= the prefix (== prefix: it lies in the
    // bucket that is taken whole; > prefix: ranked above)
    if (take_all || (key & hi_mask) >= prefix) {
      const unsigned pos = atomicAdd(&hist[261], 1u);
      if (pos < (unsigned)a.cap) sel[pos] = key;
    }
  });
  __syncthreads();
  const int got = min((int)hist[261], a.k);

  // ---- bitonic sort, descending, of the cap slots (empty slots are 0 = smallest) ----
  for (int lsize = 1; (1 << lsize) <= a.cap; ++lsize) {
    for (int ls = lsize - 1; ls >= 0; --ls) {
      for (int t = tid; t < (a.cap >> 1); t += SEL_THREADS) {
        const int lo = ((t >> ls) << (ls + 1)) + (t & ((1 << ls) - 1));
        const int hi = lo + (1 << ls);
        const bool desc = ((lo >> lsize) & 1) == 0;
        const unsigned long long x = sel[lo], y = sel[hi];
        if ((x < y) == desc) { sel[lo] = y; sel[hi] = x; }
      }
      __syncthreads();
    }
  }
  for (int j = tid; j < a.k; j += SEL_THREADS) {
    const unsigned long long key = sel[j];
    const bool real = j < got;
    a.out_idx[(size_t)row * a.k + j] = real ? (int)(0xffffffffu - (uint32_t)key) : 0;
    if (a.out_vals)
      a.out_vals[(size_t)row * a.k + j] = real ? unorder_bits((uint32_t)(key >> 32), a.largest, a.dtype)
                                               : (a.largest ? -__builtin_inff() : __builtin_inff());
  }
  if (a.out_cnt && tid == 0) a.out_cnt[row] = got;
}

}  // namespace

extern "C" int u2_topk_rows(const void* vals, int dtype, int rows, int n, long long row_stride, int group, int pitch,
                            const signed char* mask, int mask_value, int k, int largest, float* out_vals, int* out_idx,
                            int* out_cnt, const int* idx_in, void* stream) {
  if (rows <= 0 || k <= 0) return 0;
  if (n <= 0 || n >= (1 << 24) || k > 16384 || group < 1 || pitch < group || (dtype != 0 && dtype != 1)) return -1;
  SelectArgs a;
  a.vals = vals; a.dtype = dtype; a.rows = rows; a.n = n; a.row_stride = row_stride; a.group = group; a.pitch = pitch;
  a.mask = mask; a.mask_value = mask_value; a.k = k; a.largest = largest;
  a.out_vals = out_vals; a.out_idx = out_idx; a.out_cnt = out_cnt; a.idx_in = idx_in;
  int cap = 2;
  while (cap < k) cap <<= 1;
  a.cap = cap;
  const size_t lds = (size_t)cap * 8 + 272 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)topk_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(topk_rows_kernel, dim3(rows), dim3(SEL_THREADS), lds, (hipStream_t)stream, a);
  U2_CHECK_LAUNCH();
  return 0;
}
