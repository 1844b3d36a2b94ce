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
constexpr int SEL_MAX_SEGS = 8;

// One selection problem: `rows` rows of n elements each.  Several of them share a launch (a work-group finds its segment by
// its block index), so that the per-level selections of the RPN - independent, each a latency-bound chain of sweeps that
// occupies 16 to 256 of the chip's work-group slots - run side by side instead of one after the other.
struct SelectSeg {
  const void* vals;
  const signed char* mask;  // optional [rows][n]: only elements with mask == mask_value take part
  const int* idx_in;        // optional [rows][n]: the index an element stands for (tie-break and output) instead of its position
  const int* cnt_in;        // optional [rows][n / cnt_group]: element i takes part iff i % cnt_group < cnt_in[row][i / cnt_group]
  float* out_vals;          // [rows][k] (optional)
  int* out_idx;             // [rows][k]
  int* out_cnt;             // [rows] (optional): number of real entries (min(k, participating elements))
  long long row_stride;     // elements between consecutive rows
  int dtype;                // 0: fp32, 1: bf16, 2: fp32 storage whose values are bf16-representable (low 16 bits zero)
  int rows, n;
  int group, pitch;         // element i of a row lives at (i / group) * pitch + i % group   (group = pitch = 1: contiguous)
  int mask_value;
  int k, largest;
  int cnt_group;
  int idx_mod, idx_mul;     // reported index += (row % idx_mod) * idx_mul: rows that are segments of a longer row
  int cap;                  // LDS slots for the survivors (power of two >= k)
};
struct SelectArgs {
  SelectSeg seg[SEL_MAX_SEGS];
  int nseg;
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

__global__ __launch_bounds__(SEL_THREADS) void topk_rows_kernel(const SelectArgs m) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int row = blockIdx.x, si = 0;
  while (si + 1 < m.nseg && row >= m.seg[si].rows) { row -= m.seg[si].rows; ++si; }
  const SelectSeg& a = m.seg[si];
  const int cap = a.cap;
  unsigned long long* sel = reinterpret_cast<unsigned long long*>(smem_raw);          // [cap]
  unsigned int* hist = reinterpret_cast<unsigned int*>(smem_raw + (size_t)cap * 8);   // [256] + scratch [8]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int n = a.n, k = a.k, group = a.group, pitch = a.pitch, largest = a.largest;
  const int wide = a.dtype == 2;           // bf16 keys read from fp32 storage
  const int dtype = a.dtype == 0 ? 0 : 1;  // how the keys are ordered
  const int mask_value = a.mask_value, cnt_group = a.cnt_group;
  const float inv_group = 1.0f / (float)group, inv_cnt_group = 1.0f / (float)cnt_group;
  const signed char* mrow = a.mask ? a.mask + (size_t)row * n : nullptr;
  const int* irow = a.idx_in ? a.idx_in + (size_t)row * n : nullptr;
  const int* crow = a.cnt_in ? a.cnt_in + (size_t)row * (n / cnt_group) : nullptr;
  const unsigned char* vrow = reinterpret_cast<const unsigned char*>(a.vals) + (size_t)row * a.row_stride * (a.dtype == 1 ? 2 : 4);
  // the (group, pitch) view of a bf16 map with at most four valid columns per pixel: one 8-byte load fetches a pixel's
  // columns, instead of `group` 2-byte gathers that land in the same 8 bytes
  const bool grouped = a.dtype == 1 && group >= 2 && group <= 4 && (pitch & 3) == 0 && n % group == 0 && !mrow && !irow && !crow &&
                       ((reinterpret_cast<size_t>(vrow) & 7) == 0);

  auto load_key = [&](int i, bool& ok) -> uint32_t {
    ok = !mrow || mrow[i] == (signed char)mask_value;
    if (crow) {
      int g = (int)((float)i * inv_cnt_group);
      int r = i - g * cnt_group;
      if (r < 0) { --g; r += cnt_group; }
      if (r >= cnt_group) { ++g; r -= cnt_group; }
      ok = ok && r < crow[g];
    }
    size_t off = (size_t)i;
    if (group != 1 || pitch != 1) {
      int q = (int)((float)i * inv_group);
      int r = i - q * group;
      if (r < 0) { --q; r += group; }
      if (r >= group) { ++q; r -= group; }
      off = (size_t)q * pitch + r;
    }
    uint32_t u;
    if (wide) u = reinterpret_cast<const uint32_t*>(vrow)[off] >> 16;
    else if (dtype == 1) u = (uint32_t)reinterpret_cast<const bf16_t*>(vrow)[off];
    else u = reinterpret_cast<const uint32_t*>(vrow)[off];
    return order_bits(u, largest, dtype);
  };

  // one sweep over the participating elements of the row, several independent loads per thread in flight (a row is streamed by
  // a single work-group: the sweep is latency-bound, not bandwidth-bound)
  auto for_each_key = [&](auto&& fn) {
    if (grouped) {
      constexpr int UNR = 4;
      const int nq = n / group;
      for (int base = 0; base < nq; base += UNR * SEL_THREADS) {
        uint2 px[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int q = base + u * SEL_THREADS + tid;
          px[u] = make_uint2(0u, 0u);
          if (q < nq) px[u] = *reinterpret_cast<const uint2*>(vrow + (size_t)q * pitch * 2);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int q = base + u * SEL_THREADS + tid;
          if (q < nq) {
            fn(q * group, order_bits(px[u].x & 0xffffu, largest, 1));
            fn(q * group + 1, order_bits(px[u].x >> 16, largest, 1));
            if (group > 2) fn(q * group + 2, order_bits(px[u].y & 0xffffu, largest, 1));
            if (group > 3) fn(q * group + 3, order_bits(px[u].y >> 16, largest, 1));
          }
        }
      }
      return;
    }
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

  // histogram increment; the lanes of a wave that agree with its first active lane are counted with one atomic (objectness
  // logits and the high digits of any key pile up in a few buckets, and same-address LDS atomics serialise)
  auto hist_add = [&](unsigned digit) {
    const unsigned first = (unsigned)__builtin_amdgcn_readfirstlane((int)digit);
    const unsigned long long same = __ballot(digit == first);
    if (digit == first) {
      if (lane == __ffsll((long long)same) - 1) atomicAdd(&hist[first], (unsigned)__popcll(same));
    } else {
      atomicAdd(&hist[digit], 1u);
    }
  };

  // ---- most-significant-digit radix selection of the k-th largest 64-bit key (value key : ~index).  `prefix` holds the
  // `fixed` high bits decided so far, `want` how many elements of the bucket that shares them are still to be taken.
  // Key bits 63..32 = value (bf16: bits 47..32 are zero for every element), 31..24 = 0xff (index < 2^24), 23..0 = ~index.
  unsigned long long prefix = 0;
  int fixed = 0;
  int want = k;
  bool take_all = false;  // no more than k elements take part: all of them are selected
  const int value_sweeps = dtype == 1 ? 2 : 4;
  for (int sweep = 0; sweep < value_sweeps + 3; ++sweep) {
    const int shift = sweep < value_sweeps ? 56 - 8 * sweep : 16 - 8 * (sweep - value_sweeps);
    for (int b = tid; b < 256; b += SEL_THREADS) hist[b] = 0;
    __syncthreads();
    const unsigned long long hi_mask = fixed == 0 ? 0ull : (~0ull << (64 - fixed));
    for_each_key([&](int i, uint32_t vk) {
      const unsigned long long key = ((unsigned long long)vk << 32) | (uint32_t)(0xffffffffu - (uint32_t)(irow ? irow[i] : i));
      if ((key & hi_mask) == prefix) hist_add((unsigned)(key >> shift) & 255u);
    });
    __syncthreads();
    if (tid < 64) {
      // the digit of the want-th key, by the first wave: lane l holds buckets 4l .. 4l+3, suffix sums over the lanes by lane
      // shifts (a serial walk over the 256 buckets by one thread was 512 dependent LDS reads per sweep - with five to seven
      // sweeps per row most of the kernel's time on short rows)
      const uint4 c = *reinterpret_cast<const uint4*>(&hist[4 * tid]);
      const unsigned mine_total = c.x + c.y + c.z + c.w;
      unsigned suffix = mine_total;  // buckets of lanes >= this one
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned other = __shfl_down(suffix, d);
        if (tid + d < 64) suffix += other;
      }
      const unsigned total = (unsigned)__builtin_amdgcn_readfirstlane((int)suffix);
      const bool all = sweep == 0 && total <= (unsigned)want;
      // S(b) = number of keys in buckets >= b; the chosen digit is the largest b with S(b) >= want (0 if there is none)
      const unsigned s3 = suffix - mine_total + c.w, s2 = s3 + c.z, s1 = s2 + c.y, s0 = s1 + c.x;
      int local = -1;
      unsigned above = 0, size = 0;
      if (s0 >= (unsigned)want) { local = 0; above = s1; size = c.x; }
      if (s1 >= (unsigned)want) { local = 1; above = s2; size = c.y; }
      if (s2 >= (unsigned)want) { local = 2; above = s3; size = c.z; }
      if (s3 >= (unsigned)want) { local = 3; above = s3 - c.w; size = c.w; }
      const unsigned long long has = __ballot(local >= 0);
      const int owner = has ? 63 - __clzll((long long)has) : 0;
      if (tid == owner) {
        if (all) { hist[256] = 255u; hist[257] = 0u; hist[259] = 0u; }
        else if (has) { hist[256] = (unsigned)(4 * tid + local); hist[257] = above; hist[259] = size; }
        else { hist[256] = 0u; hist[257] = s1; hist[259] = c.x; }  // fewer than `want` keys in all: everything above bucket 0
        hist[258] = all ? 1u : 0u;
        hist[260] = total;
      }
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
  for (int i = tid; i < cap; i += SEL_THREADS) sel[i] = 0ull;
  if (tid == 0) hist[261] = 0;
  __syncthreads();
  for_each_key([&](int i, uint32_t vk) {
    const unsigned long long key = ((unsigned long long)vk << 32) | (uint32_t)(0xffffffffu - (uint32_t)(irow ? irow[i] : i));
    // with `fixed` high bits decided, an element is taken iff its high bits are >= the prefix (== prefix: it lies in the
    // bucket that is taken whole; > prefix: ranked above)
    if (take_all || (key & hi_mask) >= prefix) {
      const unsigned pos = atomicAdd(&hist[261], 1u);
      if (pos < (unsigned)cap) sel[pos] = key;
    }
  });
  __syncthreads();
  const int got = min((int)hist[261], k);
  const int idx_add = (row % a.idx_mod) * a.idx_mul;
  int* out_idx = a.out_idx + (size_t)row * k;
  float* out_vals = a.out_vals ? a.out_vals + (size_t)row * k : nullptr;
  if (a.out_cnt && tid == 0) a.out_cnt[row] = got;

  // ---- bitonic sort, descending, of the cap slots (empty slots are 0 = smallest).  (Runs of stages at distances < 64 as lane
  // exchanges in registers, and ranking by counting, were both measured slower than this plain form: 64-bit exchanges are two
  // dependent ds_bpermute each, and counting is 4 M 64-bit compares for 2048 slots.) ----
  for (int lsize = 1; (1 << lsize) <= cap; ++lsize) {
    for (int ls = lsize - 1; ls >= 0; --ls) {
      for (int t = tid; t < (cap >> 1); t += SEL_THREADS) {
        const int lo = ((t >> ls) << (ls + 1)) + (t & ((1 << ls) - 1));
        const int hi = lo + (1 << ls);
        const bool desc = ((lo >> lsize) & 1) == 0;
        const unsigned long long x = sel[lo], y = sel[hi];
        if ((x < y) == desc) { sel[lo] = y; sel[hi] = x; }
      }
      __syncthreads();
    }
  }
  for (int j = tid; j < k; j += SEL_THREADS) {
    const unsigned long long key = sel[j];
    const bool real = j < got;
    out_idx[j] = real ? (int)(0xffffffffu - (uint32_t)key) + idx_add : 0;
    if (out_vals)
      out_vals[j] = real ? unorder_bits((uint32_t)(key >> 32), largest, dtype) : (largest ? -__builtin_inff() : __builtin_inff());
  }
}

}  // namespace

static_assert(sizeof(U2TopkSeg) == 7 * sizeof(void*) + sizeof(long long) + 12 * sizeof(int), "U2TopkSeg layout");

extern "C" int u2_topk_rows_multi(const U2TopkSeg* segs, int nseg, void* stream) {
  if (nseg <= 0) return 0;
  if (nseg > SEL_MAX_SEGS || !segs) return -1;
  SelectArgs m;
  m.nseg = 0;
  long long grid = 0;
  int max_cap = 2;
  for (int j = 0; j < nseg; ++j) {
    const U2TopkSeg& g = segs[j];
    if (g.rows <= 0 || g.k <= 0) continue;
    if (g.n <= 0 || g.n >= (1 << 24) || g.k > 16384 || g.group < 1 || g.pitch < g.group || g.dtype < 0 || g.dtype > 2 ||
        !g.vals || !g.out_idx || g.idx_mod < 1 || (long long)(g.idx_mod - 1) * g.idx_mul + g.n > (1 << 24) ||
        (g.cnt_in && (g.cnt_group < 1 || g.n % g.cnt_group)))
      return -1;
    SelectSeg& a = m.seg[m.nseg++];
    a.vals = g.vals; a.mask = g.mask; a.idx_in = g.idx_in; a.cnt_in = g.cnt_in; a.out_vals = g.out_vals; a.out_idx = g.out_idx;
    a.out_cnt = g.out_cnt; a.row_stride = g.row_stride; a.dtype = g.dtype; a.rows = g.rows; a.n = g.n; a.group = g.group;
    a.pitch = g.pitch; a.mask_value = g.mask_value; a.k = g.k; a.largest = g.largest; a.cnt_group = g.cnt_in ? g.cnt_group : 1;
    a.idx_mod = g.idx_mod; a.idx_mul = g.idx_mul;
    int cap = 2;
    while (cap < g.k) cap <<= 1;
    a.cap = cap;
    if (cap > max_cap) max_cap = cap;
    grid += g.rows;
  }
  if (m.nseg == 0) return 0;
  if (grid >= (1LL << 31)) return -1;
  const size_t lds = (size_t)max_cap * 8 + 272 * 4;
  static PerDeviceOnce attr_set;
  if (auto once_guard = attr_set.first()) {
    (void)hipFuncSetAttribute((const void*)topk_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)grid), dim3(SEL_THREADS), lds, (hipStream_t)stream, m);
  U2_CHECK_LAUNCH();
  return 0;
}

extern "C" int u2_topk_rows(const void* vals, int dtype, int rows, int n, long long row_stride, int group, int pitch,
                            const signed char* mask, int mask_value, int k, int largest, float* out_vals, int* out_idx,
                            int* out_cnt, const int* idx_in, void* stream) {
  if (rows <= 0 || k <= 0) return 0;
  U2TopkSeg g;
  g.vals = vals; g.mask = mask; g.idx_in = idx_in; g.cnt_in = nullptr; g.out_vals = out_vals; g.out_idx = out_idx;
  g.out_cnt = out_cnt; g.row_stride = row_stride; g.dtype = dtype; g.rows = rows; g.n = n; g.group = group; g.pitch = pitch;
  g.mask_value = mask_value; g.k = k; g.largest = largest; g.cnt_group = 1; g.idx_mod = 1; g.idx_mul = 0;
  return u2_topk_rows_multi(&g, 1, stream);
}
