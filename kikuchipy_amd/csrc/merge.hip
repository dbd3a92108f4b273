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
// winner" over all candidates (they sit in L2).  Latency-bound and tiny next to
// the match kernel (M*k*(lists*len) key compares).
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
  int len[3];         // entries per list
  int stride[3];      // elements between patterns
  int list_stride[3]; // elements between lists
  float *out_s;
  int *out_i;
  int out_stride, out_offset;
};

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
        const size_t e = (size_t)l * a.list_stride[j] + (c - l * len);
        const int idx = pi[e];
        if (idx == INT_MAX) continue;
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
    a.stride[j] = j < l.n_src ? l.src_row_stride[j] : 0;
    a.list_stride[j] = j < l.n_src ? l.src_list_stride[j] : 0;
  }
  a.out_s = l.out_scores;
  a.out_i = l.out_idx;
  a.out_stride = l.out_stride;
  a.out_offset = l.out_offset;
  hipLaunchKernelGGL(merge_kernel, dim3((l.m + 3) / 4), dim3(256), 0, s, a);
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
