// merge.hip - merge sorted/unsorted candidate lists into the best k per
// experimental pattern.
//
// Reference: the host-side merge of `_dictionary_indexing`
// (indexing/_dictionary_indexing.py:120-128: hstack running + chunk results,
// argsort(-scores)[:, :keep_n], take_along_axis) and, across GPUs, the same
// operation over the per-shard lists.  Ordering key = (score descending,
// dictionary index ascending): a total order, so the result does not depend on
// how candidates were split over lanes, workgroups, chunks or ranks.
//
// One wave per experimental pattern; k rounds of "largest key below the previous
// winner" over all candidates.  Up to 48 * 64 candidates per pattern are read ONCE into
// registers as 64-bit keys (merge_cached_kernel; config 2: 32 lists x 20 + the running 20 =
// 660 candidates, 11 per lane); up to 64 * 256 with one workgroup per pattern
// (merge_block_kernel: few experimental patterns -> many lists each); more are re-read from
// L2 every round (merge_kernel).
// Latency-bound and tiny next to the match kernel.
#include <algorithm>
#include "kernels.h"
#include <limits.h>
#include <math.h>

namespace kpdi {

__device__ __forceinline__ unsigned long long topk_key(float s, int idx) {
  unsigned u = __float_as_uint(s + 0.f);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | (unsigned)(INT_MAX - idx);
}
__device__ __forceinline__ float key_score(unsigned long long key) {
  unsigned u = (unsigned)(key >> 32);
  u = (u & 0x80000000u) ? (u ^ 0x80000000u) : ~u;
  return __uint_as_float(u);
}
__device__ __forceinline__ int key_idx(unsigned long long key) {
  return INT_MAX - (int)(unsigned)(key & 0xffffffffu);
}

struct MergeArgs {
  int m, k, n_src;
  const float *s[3];
  const int *i[3];
  int lists[3];       // lists per pattern in source j
  const int *cnt[3];  // [m][lists] valid entries of every list of source j, or nullptr (all of them)
  int len[3];         // entries per list
  int stride[3];      // elements between patterns
  int list_stride[3]; // elements between lists
  float *out_s;
  int *out_i;
  int out_stride, out_offset;
  IndexSegments seg;
  unsigned seg_sources;
};

// row of a coalesced matrix -> dictionary index (kernels.h: IndexSegments); INT_MAX (no entry) stays
__device__ __forceinline__ int segment_index(const MergeArgs &a, int row) {
  int d = a.seg.delta[0];
#pragma unroll
  for (int t = 1; t < INDEX_SEGMENTS; ++t) d = row >= a.seg.row0[t] ? a.seg.delta[t] : d;
  return row == INT_MAX ? row : row + d;
}

__global__ __launch_bounds__(256) void merge_kernel(MergeArgs a) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= a.m) return;
  unsigned long long prev = ~0ull;
  for (int r = 0; r < a.k; ++r) {
    unsigned long long best = 0ull;
    for (int j = 0; j < a.n_src; ++j) {
      const float *ps = a.s[j] + (size_t)m * a.stride[j];
      const int *pi = a.i[j] + (size_t)m * a.stride[j];
      const int len = a.len[j];
      const int count = a.lists[j] * len;
      for (int c = lane; c < count; c += 64) {
        const int l = c / len;
        if (a.cnt[j] && c - l * len >= a.cnt[j][(size_t)m * a.lists[j] + l]) continue;  // (behind what the list holds: never written)
        const size_t e = (size_t)l * a.list_stride[j] + (c - l * len);
        int idx = pi[e];
        if (idx == INT_MAX) continue;
        if ((a.seg_sources >> j) & 1u) idx = segment_index(a, idx);
        const unsigned long long key = topk_key(ps[e], idx);
        if (key < prev && key > best) best = key;
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const unsigned lo = __shfl_xor((unsigned)(best & 0xffffffffu), o, 64);
      const unsigned hi = __shfl_xor((unsigned)(best >> 32), o, 64);
      const unsigned long long other = ((unsigned long long)hi << 32) | lo;
      best = other > best ? other : best;
    }
    if (lane == 0) {
      const size_t o = (size_t)m * a.out_stride + a.out_offset + r;
      if (best == 0ull) {
        a.out_s[o] = -INFINITY;
        a.out_i[o] = INT_MAX;
      } else {
        a.out_s[o] = key_score(best);
        a.out_i[o] = key_idx(best);
      }
    }
    if (best != 0ull) prev = best;
    else prev = 0ull;
  }
}

// candidate c (sources laid end to end) of pattern m as a key; 0 = none.  Branch-free: a divergent
// branch around the loads makes the compiler copy the whole key array at every join (312 VGPRs);
// slots past the last candidate read candidate 0 and are zeroed.
// `held[j]`: source j comes with counts and has at most 64 lists - lane l then holds the count of list l of this
// pattern (one coalesced load per wave; a candidate's count is fetched from its list's lane), INT_MAX in every lane
// otherwise (counts, if any, are then read from memory per candidate).
struct ListCounts {
  int held[3];
  bool in_lanes[3];
};
__device__ __forceinline__ ListCounts load_counts(const MergeArgs &a, int m, int lane) {
  ListCounts lc;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    lc.in_lanes[j] = a.cnt[j] != nullptr && a.lists[j] <= 64;
    lc.held[j] = INT_MAX;
    if (lc.in_lanes[j] && lane < a.lists[j]) lc.held[j] = a.cnt[j][(size_t)m * a.lists[j] + lane];
  }
  return lc;
}

__device__ __forceinline__ unsigned long long candidate_key(const MergeArgs &a, const ListCounts &lc, int m, int c0, int end0,
                                                            int end1, int end2) {
  const bool live = c0 < end2;
  const int c = live ? c0 : 0;
  // the source of candidate c, picked with constant indices into the argument arrays
  const bool in0 = c < end0, in1 = c < end1;
  const float *ps = in0 ? a.s[0] : (in1 ? a.s[1] : a.s[2]);
  const int *pi = in0 ? a.i[0] : (in1 ? a.i[1] : a.i[2]);
  const int len = in0 ? a.len[0] : (in1 ? a.len[1] : a.len[2]);
  const int stride = in0 ? a.stride[0] : (in1 ? a.stride[1] : a.stride[2]);
  const int list_stride = in0 ? a.list_stride[0] : (in1 ? a.list_stride[1] : a.list_stride[2]);
  const int local = c - (in0 ? 0 : (in1 ? end0 : end1));
  // local / len without an integer division: local < 16384, len <= 32, so the float quotient of
  // local + 0.5 is at least 1/64 away from an integer (and float holds it to 2^-9)
  const int l = (int)(((float)local + 0.5f) / (float)len);
  const size_t e = (size_t)m * stride + (size_t)l * list_stride + (local - l * len);
  // a source with counts: only the first cnt[m][l] entries of a list were written - what lies behind them is stale memory
  const int *pc = in0 ? a.cnt[0] : (in1 ? a.cnt[1] : a.cnt[2]);
  const bool lanes = in0 ? lc.in_lanes[0] : (in1 ? lc.in_lanes[1] : lc.in_lanes[2]);
  // the count of list l of THIS candidate's source sits in lane l's register OF THAT SOURCE: one shuffle per source that
  // keeps its counts in lanes (uniform over the launch), every lane taking part, each keeping its own source's answer
  int n_held = INT_MAX;
  if (lc.in_lanes[0]) {
    const int t = __shfl(lc.held[0], l & 63, 64);
    n_held = in0 ? t : n_held;
  }
  if (lc.in_lanes[1]) {
    const int t = __shfl(lc.held[1], l & 63, 64);
    n_held = (!in0 && in1) ? t : n_held;
  }
  if (lc.in_lanes[2]) {
    const int t = __shfl(lc.held[2], l & 63, 64);
    n_held = (!in0 && !in1) ? t : n_held;
  }
  if (pc != nullptr && !lanes) n_held = pc[(size_t)m * (in0 ? a.lists[0] : (in1 ? a.lists[1] : a.lists[2])) + l];
  const bool held = pc == nullptr || (local - l * len) < n_held;
  int idx = pi[e];
  if (a.seg_sources != 0) {  // (uniform over the launch)
    const unsigned bit = in0 ? 1u : (in1 ? 2u : 4u);
    idx = (a.seg_sources & bit) ? segment_index(a, idx) : idx;
  }
  const unsigned long long key = topk_key(ps[e], idx);
  return (live && held && idx != INT_MAX) ? key : 0ull;
}

__device__ __forceinline__ unsigned long long wave_max_key(unsigned long long best) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)(best & 0xffffffffu), o, 64);
    const unsigned hi = __shfl_xor((unsigned)(best >> 32), o, 64);
    const unsigned long long other = ((unsigned long long)hi << 32) | lo;
    best = other > best ? other : best;
  }
  return best;
}

__device__ __forceinline__ void store_rank(const MergeArgs &a, int m, int r, unsigned long long best) {
  const size_t o = (size_t)m * a.out_stride + a.out_offset + r;
  if (best == 0ull) {
    a.out_s[o] = -INFINITY;
    a.out_i[o] = INT_MAX;
  } else {
    a.out_s[o] = key_score(best);
    a.out_i[o] = key_idx(best);
  }
}

// candidates in registers, one WAVE per pattern: lane holds candidates lane, lane + 64, ...
template <int NK>
__global__ __launch_bounds__(256) void merge_cached_kernel(MergeArgs a) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= a.m) return;
  const int end0 = a.lists[0] * a.len[0];
  const int end1 = end0 + (a.n_src > 1 ? a.lists[1] * a.len[1] : 0);
  const int end2 = end1 + (a.n_src > 2 ? a.lists[2] * a.len[2] : 0);
  const ListCounts lc = load_counts(a, m, lane);
  unsigned long long keys[NK];
#pragma unroll
  for (int i = 0; i < NK; ++i) keys[i] = candidate_key(a, lc, m, lane + 64 * i, end0, end1, end2);
  // Most candidate slots hold nothing (a partial list of the match kernel keeps one or two entries of its 20): the k
  // selection rounds below scan every slot of a lane, 24 x k compare-selects per lane - the kernel was VALU-issue-bound on
  // them (4096 waves x 5000 instructions: 38 us at configs[1]).  The real candidates are first packed into the wave's
  // part of LDS (ballot + prefix count), at most CK per lane; the rounds then scan ceil(n / 64) keys.  More than 64 CK of
  // them (long lists, many sources): the rounds run over the registers as before.  The order of candidates is immaterial
  // (keys are totally ordered): the result is the same bit for bit.
  constexpr int CK = NK < 8 ? NK : 8;
  __shared__ unsigned long long packed[4][64 * CK];
  unsigned long long *mine = packed[threadIdx.x >> 6];
  int n = 0;
#pragma unroll
  for (int i = 0; i < NK; ++i) {
    const unsigned long long b = __builtin_amdgcn_ballot_w64(keys[i] != 0ull);
    const int pos = n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
    if (keys[i] != 0ull && pos < 64 * CK) mine[pos] = keys[i];
    n += __builtin_popcountll(b);
  }
  unsigned long long prev = ~0ull;
#ifdef KPDI_MERGE_NO_COMPACT  // (developer build: the rounds over the registers, always)
  n = 64 * CK + 1;
#endif
  if (n <= 64 * CK) {
    unsigned long long ck[CK];
#pragma unroll
    for (int j = 0; j < CK; ++j) ck[j] = lane + 64 * j < n ? mine[lane + 64 * j] : 0ull;  // (this wave's own writes: no barrier)
    const int per_lane = (n + 63) >> 6;
    for (int r = 0; r < a.k; ++r) {
      unsigned long long best = 0ull;
#pragma unroll
      for (int j = 0; j < CK; ++j)
        if (j < per_lane && ck[j] < prev && ck[j] > best) best = ck[j];
      best = wave_max_key(best);
      if (lane == 0) store_rank(a, m, r, best);
      prev = best;  // 0 once the candidates are exhausted: nothing is below it
    }
    return;
  }
  for (int r = 0; r < a.k; ++r) {
    unsigned long long best = 0ull;
#pragma unroll
    for (int i = 0; i < NK; ++i)
      if (keys[i] < prev && keys[i] > best) best = keys[i];
    best = wave_max_key(best);
    if (lane == 0) store_rank(a, m, r, best);
    prev = best;  // 0 once the candidates are exhausted: nothing is below it
  }
}

// candidates in registers, one WORKGROUP per pattern: few experimental patterns make the launch
// plan split the dictionary tiles over up to 256 workgroups, i.e. up to 2 * 256 lists per pattern
template <int NK>
__global__ __launch_bounds__(256) void merge_block_kernel(MergeArgs a) {
  __shared__ unsigned long long wave_best[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int m = blockIdx.x;
  const int end0 = a.lists[0] * a.len[0];
  const int end1 = end0 + (a.n_src > 1 ? a.lists[1] * a.len[1] : 0);
  const int end2 = end1 + (a.n_src > 2 ? a.lists[2] * a.len[2] : 0);
  ListCounts lc;  // (one workgroup per pattern: up to 1024 lists - counts, if any, are read per candidate)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    lc.held[j] = INT_MAX;
    lc.in_lanes[j] = false;
  }
  unsigned long long keys[NK];
#pragma unroll
  for (int i = 0; i < NK; ++i) keys[i] = candidate_key(a, lc, m, tid + 256 * i, end0, end1, end2);
  unsigned long long prev = ~0ull;
  for (int r = 0; r < a.k; ++r) {
    unsigned long long best = 0ull;
#pragma unroll
    for (int i = 0; i < NK; ++i)
      if (keys[i] < prev && keys[i] > best) best = keys[i];
    best = wave_max_key(best);
    if (lane == 0) wave_best[r & 1][wv] = best;  // two buffers: one barrier per round
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const unsigned long long other = wave_best[r & 1][w];
      best = other > best ? other : best;
    }
    if (tid == 0) store_rank(a, m, r, best);
    prev = best;
  }
}

hipError_t launch_merge(const MergeLaunch &l, hipStream_t s) {
  if (l.m <= 0 || l.k <= 0) return hipSuccess;
  MergeArgs a;
  a.m = l.m;
  a.k = l.k;
  a.n_src = l.n_src;
  for (int j = 0; j < 3; ++j) {
    a.s[j] = j < l.n_src ? l.src_scores[j] : nullptr;
    a.i[j] = j < l.n_src ? l.src_idx[j] : nullptr;
    a.lists[j] = j < l.n_src ? l.src_lists[j] : 0;
    a.len[j] = j < l.n_src ? l.src_len[j] : 1;
    a.cnt[j] = j < l.n_src ? l.src_cnt[j] : nullptr;
    a.stride[j] = j < l.n_src ? l.src_row_stride[j] : 0;
    a.list_stride[j] = j < l.n_src ? l.src_list_stride[j] : 0;
  }
  a.out_s = l.out_scores;
  a.out_i = l.out_idx;
  a.out_stride = l.out_stride;
  a.out_offset = l.out_offset;
  a.seg = l.seg;
  a.seg_sources = l.seg.n > 0 ? l.seg_sources : 0u;
  for (int t = std::max(l.seg.n, 0); t < INDEX_SEGMENTS; ++t) {
    a.seg.row0[t] = INT_MAX;
    a.seg.delta[t] = 0;
  }
  int candidates = 0;
  for (int j = 0; j < l.n_src; ++j) candidates += l.src_lists[j] * l.src_len[j];
  const dim3 grid((l.m + 3) / 4), block(256);
  if (candidates <= 4 * 64)
    hipLaunchKernelGGL(merge_cached_kernel<4>, grid, block, 0, s, a);
  else if (candidates <= 12 * 64)
    hipLaunchKernelGGL(merge_cached_kernel<12>, grid, block, 0, s, a);
  else if (candidates <= 24 * 64)
    hipLaunchKernelGGL(merge_cached_kernel<24>, grid, block, 0, s, a);
  else if (candidates <= 48 * 64)
    hipLaunchKernelGGL(merge_cached_kernel<48>, grid, block, 0, s, a);
  else if (candidates <= 24 * 256)
    hipLaunchKernelGGL(merge_block_kernel<24>, dim3(l.m), block, 0, s, a);
  else if (candidates <= 64 * 256)
    hipLaunchKernelGGL(merge_block_kernel<64>, dim3(l.m), block, 0, s, a);
  else
    hipLaunchKernelGGL(merge_kernel, grid, block, 0, s, a);
  return hipGetLastError();
}

__global__ void fill_topk_kernel(float *scores, int *idx, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    scores[i] = -INFINITY;
    idx[i] = INT_MAX;
  }
}

hipError_t launch_fill_topk(float *scores, int *idx, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(fill_topk_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, scores, idx, n);
  return hipGetLastError();
}

__global__ void fill_u32_kernel(unsigned *p, unsigned value, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = value;
}

hipError_t launch_fill_u32(unsigned *p, unsigned value, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, value, n);
  return hipGetLastError();
}

// see kernels.h: FillSegments.  blockIdx.y = segment; 16-byte stores when the range allows them.
__global__ __launch_bounds__(256) void fill_segments_kernel(FillSegments f) {
  const int seg = blockIdx.y;
  unsigned *p = f.p[seg];
  const unsigned long long n = f.words[seg];
  const unsigned value = f.value[seg];
  const int used = f.bound_used[seg];
  const unsigned long long stride = (unsigned long long)gridDim.x * 256;
  unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  if (used < 0 && (((uintptr_t)p | (uintptr_t)(n * 4)) & 15) == 0) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = {value, value, value, value};
    for (; i < n / 4; i += stride) ((u32x4 *)p)[i] = v;
    return;
  }
  for (; i < n; i += stride)
    p[i] = used < 0 ? value : ((int)(i & (BOUND_SLOTS - 1)) < used ? THRESHOLD_NONE : 0xffffffffu);
}

hipError_t launch_fill_segments(const FillSegments &f, hipStream_t s) {
  if (f.n <= 0) return hipSuccess;
  unsigned long long most = 0;
  for (int i = 0; i < f.n; ++i) most = std::max(most, f.words[i]);
  if (most == 0) return hipSuccess;
  const unsigned gx = (unsigned)std::min<unsigned long long>((most + 1023) / 1024, 1024);
  hipLaunchKernelGGL(fill_segments_kernel, dim3(gx, (unsigned)f.n), dim3(256), 0, s, f);
  return hipGetLastError();
}

__global__ void init_bound_kernel(unsigned *gthr, int64_t n, int used) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) gthr[i] = (int)(i & (BOUND_SLOTS - 1)) < used ? THRESHOLD_NONE : 0xffffffffu;
}

hipError_t launch_init_bound(unsigned *gthr, int m_pad, int used_slots, hipStream_t s) {
  const int64_t n = (int64_t)m_pad * BOUND_SLOTS;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(init_bound_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gthr, n, used_slots);
  return hipGetLastError();
}

__global__ void last_column_kernel(const float *scores, const int *idx, int m, int stride, int col,
                                   float *bs, int *bi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) {
    bs[i] = scores[(size_t)i * stride + col];
    bi[i] = idx[(size_t)i * stride + col];
  }
}

hipError_t launch_last_column(const float *scores, const int *idx, int m, int stride, int col,
                              float *bound_score, int *bound_idx, hipStream_t s) {
  if (m <= 0) return hipSuccess;
  hipLaunchKernelGGL(last_column_kernel, dim3((m + 255) / 256), dim3(256), 0, s, scores, idx, m, stride,
                     col, bound_score, bound_idx);
  return hipGetLastError();
}

}  // namespace kpdi
