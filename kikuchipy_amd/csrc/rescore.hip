// rescore.hip - float64 arithmetic for `dtype=float64` (KPDI_COMPUTE_F64), the MI355X way.
//
// The reference computes everything in float64 when the metric's dtype is float64
// (_similarity_metric.py:244-253 allows f32 / f64; _normalized_cross_correlation.py:88-159 casts, centres
// and normalises in that dtype; the einsum of :161-183 is then a dgemm).  A dense f64 GEMM is the wrong
// tool here: only keep_n of the N scores of a pattern survive.  So the exact-f32 MFMA path (match.hip,
// 141 TFLOP/s) SCREENS - it yields, per pattern and dictionary chunk, the keep_n + 12 best candidates in its
// own arithmetic - and this file RESCORES those candidates in float64 from the RAW patterns
// (rescore_kernel: centre, normalise and dot product in double, straight from the caller's data), merges
// them into the running float64 best-k (merge64_kernel) and CERTIFIES the result: a candidate that was not
// rescored has an f32 score <= the last screened candidate's, hence an f64 score <= that + eps, where eps
// bounds |f32 score - f64 score| (8 x the largest difference observed over all rescored pairs of the sweep,
// >= 1e-6); if the running k-th best f64 score exceeds that, no unscreened candidate can belong to the
// top k.  Patterns that fail the test get further screening passes (api.hip); what is still uncertified
// after those is counted and reported (kpdi_counters.uncertified_patterns) - it takes more than 100
// dictionary patterns within eps of the k-th best.
//
// Scores agree with a float64 evaluation of the reference's formula to ~1e-15 (summation order differs).
#include "prep_device.h"
#include <limits.h>
#include <math.h>

namespace kpdi {

__device__ __forceinline__ double raw_value(const void *p, int dtype, size_t i) {
  switch (dtype) {
    case KPDI_U8: return (double)((const uint8_t *)p)[i];
    case KPDI_I8: return (double)((const int8_t *)p)[i];
    case KPDI_U16: return (double)((const uint16_t *)p)[i];
    case KPDI_I16: return (double)((const int16_t *)p)[i];
    case KPDI_F16: return (double)((const _Float16 *)p)[i];
    case KPDI_F32: return (double)((const float *)p)[i];
    case KPDI_I32: return (double)((const int32_t *)p)[i];
    case KPDI_U32: return (double)((const uint32_t *)p)[i];
    default: return ((const double *)p)[i];
  }
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

constexpr int RESCORE_THREADS = 256;

// One workgroup per experimental pattern; wave w rescores candidates w, w + 4, ...
__global__ __launch_bounds__(RESCORE_THREADS) void rescore_kernel(RescoreLaunch a) {
  __shared__ double red[RESCORE_THREADS / 64];
  __shared__ double xstat[3];
  __shared__ float diff_red[RESCORE_THREADS / 64];
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t xrow = (size_t)(a.row_map ? a.row_map[m] : m) * a.npix;
  const bool centre = a.metric != KPDI_METRIC_NDP;
  // ---- the experimental pattern: mean, then the sum and the sum of squares of the centred kept pixels
  double s = 0.0;
  if (centre)
    for (int i = tid; i < a.k; i += RESCORE_THREADS) s += raw_value(a.exp_raw, a.exp_dtype, xrow + (a.pix_map ? a.pix_map[i] : i));
  s = wave_sum_f64(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const double mx = centre ? ((red[0] + red[1]) + (red[2] + red[3])) / (double)a.k : 0.0;
  __syncthreads();
  // (x0 = the first kept pixel: sum (x - x0)^2 is EXACTLY zero for a constant pattern and for no other - the exact
  // test of include/kpdi.h "Degenerate patterns"; an all-zero pattern under `ndp` has sxx == 0)
  const double x0 = raw_value(a.exp_raw, a.exp_dtype, xrow + (a.pix_map ? a.pix_map[0] : 0));
  double q = 0.0, r1 = 0.0, qc = 0.0;
  for (int i = tid; i < a.k; i += RESCORE_THREADS) {
    const double raw = raw_value(a.exp_raw, a.exp_dtype, xrow + (a.pix_map ? a.pix_map[i] : i));
    const double v = raw - mx;
    q += v * v;
    r1 += v;
    qc += (raw - x0) * (raw - x0);
  }
  q = wave_sum_f64(q);
  r1 = wave_sum_f64(r1);
  qc = wave_sum_f64(qc);
  if (lane == 0) {
    red[wave] = q;
    diff_red[wave] = 0.f;
  }
  __syncthreads();
  if (tid == 0) xstat[1] = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  if (lane == 0) red[wave] = r1;
  __syncthreads();
  if (tid == 0) xstat[2] = (red[0] + red[1]) + (red[2] + red[3]);  // sum of the centred pixels: ~1e-13, not 0
  __syncthreads();
  if (lane == 0) red[wave] = qc;
  __syncthreads();
  const bool x_constant = centre && !((red[0] + red[1]) + (red[2] + red[3]) > 0.0);  // (NaN: not > 0 -> degenerate)
  const double sxx = xstat[1], sx_res = xstat[2];
  // ---- candidates: ONE pass over a dictionary row.  With y0 = its first kept pixel (a shift that keeps the
  // one-pass variance free of cancellation) and x' = x - mean(x):
  //   sum (y - my)^2 = sum (y - y0)^2 - (sum (y - y0))^2 / K
  //   sum x' (y - my) = sum x' (y - y0) - (my - y0) sum x'
  float worst = 0.f;
  for (int j = wave; j < a.n_cand; j += RESCORE_THREADS / 64) {
    const size_t ci = (size_t)m * a.cand_stride + a.cand_offset + j;
    const int idx = a.cand_i[ci];
    const float s32 = a.cand_s[ci];
    double score = -INFINITY;
    const int64_t local = (int64_t)idx - a.global_start;
    if (idx != INT_MAX && local >= 0 && local < a.n_chunk) {
      const size_t yrow = (size_t)local * a.npix;
      const double y0 = centre ? raw_value(a.dict_raw, a.dict_dtype, yrow + (a.pix_map ? a.pix_map[0] : 0)) : 0.0;
      double s1 = 0.0, s2 = 0.0, sxy = 0.0;
      for (int i = lane; i < a.k; i += 64) {
        const int p = a.pix_map ? a.pix_map[i] : i;
        const double x = raw_value(a.exp_raw, a.exp_dtype, xrow + p) - mx;
        const double y = raw_value(a.dict_raw, a.dict_dtype, yrow + p) - y0;
        sxy += x * y;
        s1 += y;
        s2 += y * y;
      }
      sxy = wave_sum_f64(sxy);
      s1 = wave_sum_f64(s1);
      s2 = wave_sum_f64(s2);
      const double syy = centre ? s2 - s1 * s1 / (double)a.k : s2;
      if (centre) sxy -= (s1 / (double)a.k) * sx_res;
      // degenerate patterns (zero variance / all zeros / NaN or inf in the data: include/kpdi.h) score exactly 0, as
      // in the f32 path, which prepares them as all-zero rows; the reference divides 0 by 0 there
      // (s2 = sum (y - y0)^2: exactly zero for a constant dictionary pattern, and only then)
      const bool degenerate = x_constant || (centre && !(s2 > 0.0)) || degenerate_pattern(sxx, 0.0, 1.0, false) ||
                              degenerate_pattern(syy, 0.0, 1.0, false);
      score = degenerate ? 0.0 : sxy / (sqrt(sxx) * sqrt(syy));
      if (!(score == score)) score = 0.0;
      if (score > -INFINITY) worst = fmaxf(worst, fabsf((float)(score - (double)s32)));
    }
    if (lane == 0) a.cand_s64[ci] = score;
  }
  if (lane == 0) diff_red[wave] = worst;
  __syncthreads();
  if (tid == 0) {
    const float w = fmaxf(fmaxf(diff_red[0], diff_red[1]), fmaxf(diff_red[2], diff_red[3]));
    if (w > 0.f) atomicMax(a.max_diff, __float_as_uint(w));  // non-negative floats order like their bits
  }
}

hipError_t launch_rescore(const RescoreLaunch &a, hipStream_t s) {
  if (a.m <= 0 || a.n_cand <= 0) return hipSuccess;
  hipLaunchKernelGGL(rescore_kernel, dim3(a.m), dim3(RESCORE_THREADS), 0, s, a);
  return hipGetLastError();
}

// Running float64 best-k of a pattern <- its running list + `lists` candidate lists of `len` entries, by
// (score descending, dictionary index ascending, position): every thread ranks its elements by counting.
// In place: everything is read into LDS before anything is written.
constexpr int MERGE64_THREADS = 256;
__global__ __launch_bounds__(MERGE64_THREADS) void merge64_kernel(Merge64Launch a) {
  extern __shared__ __attribute__((aligned(16))) char smem64[];
  const int m = blockIdx.x, tid = threadIdx.x;
  const int n_run = a.run_s ? a.k : 0;
  const int n = n_run + a.lists * a.len;
  double *es = (double *)smem64;
  int *ei = (int *)(es + n);
  for (int e = tid; e < n; e += MERGE64_THREADS) {
    double sc;
    int id;
    if (e < n_run) {
      sc = a.run_s[(size_t)m * a.k + e];
      id = a.run_i[(size_t)m * a.k + e];
    } else {
      const int l = (e - n_run) / a.len, j = (e - n_run) % a.len;
      const size_t o = (size_t)m * a.row_stride + (size_t)l * a.list_stride + j;
      sc = a.cand_s64[o];
      id = a.cand_i[o];
    }
    if (id == INT_MAX || !(sc == sc)) sc = -INFINITY;
    es[e] = sc;
    ei[e] = sc == -INFINITY ? INT_MAX : id;
  }
  __syncthreads();
  for (int e = tid; e < n; e += MERGE64_THREADS) {
    const double sc = es[e];
    const int id = ei[e];
    int rank = 0;
    for (int u = 0; u < n; ++u) {
      const double su = es[u];
      const int iu = ei[u];
      rank += (su > sc) || (su == sc && (iu < id || (iu == id && u < e)));
    }
    if (rank < a.k) {
      a.out_s[(size_t)m * a.k + rank] = sc;
      a.out_i[(size_t)m * a.k + rank] = id;
    }
    // certification, by whoever holds the k-th best
    if (rank == a.k - 1 && a.uncertified) {
      bool ok = a.enumerated_all != 0;
      if (!ok) {
        const float last32 = a.cand_s32[(size_t)m * a.s32_stride + a.s32_col];
        const float eps = fmaxf(8.f * __uint_as_float(*a.max_diff), a.eps_floor);
        ok = last32 == -INFINITY || sc > (double)last32 + (double)eps;
      }
      if (!ok) atomicAdd(a.uncertified, 1);
    }
  }
}

hipError_t launch_merge64(const Merge64Launch &a, hipStream_t s) {
  if (a.m <= 0) return hipSuccess;
  const size_t n = (size_t)(a.run_s ? a.k : 0) + (size_t)a.lists * a.len;
  const size_t lds = n * (sizeof(double) + sizeof(int));
  if (lds > 150 * 1024) return hipErrorInvalidValue;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)merge64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(merge64_kernel, dim3(a.m), dim3(MERGE64_THREADS), lds, s, a);
  return hipGetLastError();
}

__global__ void fill_topk64_kernel(double *s, int *i, int64_t n) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) {
    s[t] = -INFINITY;
    i[t] = INT_MAX;
  }
}
hipError_t launch_fill_topk64(double *scores, int *idx, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(fill_topk64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, scores, idx, n);
  return hipGetLastError();
}

}  // namespace kpdi
