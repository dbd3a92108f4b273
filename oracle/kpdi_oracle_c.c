/* kpdi_oracle_c.c - plain C restatement of the match / top-k / merge stage.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (only tests/ load the resulting
 * oracle/libkpdi_oracle.so; kikuchipy_amd/ never does).  An independent second
 * statement of the algorithm, cross-checked against the NumPy oracle
 * (oracle/kpdi_oracle.py, which is pinned to the reference's golden vectors) in
 * tests/test_oracle_c.py.  Build: `make -C oracle` (gcc -O3 -mavx2 -fopenmp).
 *
 * Reference lines restated (under /root/reference/src/kikuchipy/):
 *   zero-mean + L2 normalise   indexing/similarity_metrics/_normalized_cross_correlation.py:228-233
 *   L2 normalise               indexing/similarity_metrics/_normalized_dot_product.py:181-194
 *   S = X . Y^T                ..._normalized_cross_correlation.py:181-183 (einsum "ik,mk->im")
 *   k largest, descending      indexing/_dictionary_indexing.py:197-198 (argtopk / topk)
 *   chunk loop + merge         indexing/_dictionary_indexing.py:94-128
 * Tie rule: lower dictionary index first (the engine's documented rule).
 */
#include <immintrin.h>
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* OpenMP threads of everything below (a container's CPU quota can be far below its visible cores) */
void kpdi_c_set_threads(int n) {
  if (n > 0) omp_set_num_threads(n);
}

/* Degenerate patterns - a constant pattern (ncc) / all zeros (ndp) / NaN or inf among the pixels: the reference divides
 * 0 by 0 (NaN, ranked first by Dask's topk); the ENGINE's documented rule (include/kpdi.h, "Degenerate patterns";
 * csrc/prep_device.h: degenerate_pattern), which this checker follows: the row becomes all zeros, every score of it is
 * exactly 0.  norm2 = the sum of squares the row is divided by the root of; constant = (ncc only) all kept pixels are
 * EQUAL - an exact test, no contrast floor. */
static int degenerate_pattern(double norm2, int constant) {
  return !(norm2 > 0.0 && norm2 < INFINITY) || constant;
}

/* rows: n x k, in place.  metric 0 = ncc (subtract mean first), 1 = ndp. */
void kpdi_c_normalize(float *rows, int64_t n, int64_t k, int metric) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    float *p = rows + r * k;
    float mean = 0.f;
    float lo = p[0], hi = p[0];
    for (int64_t i = 1; i < k; ++i) {
      lo = p[i] < lo ? p[i] : lo;
      hi = p[i] > hi ? p[i] : hi;
    }
    if (metric == 0) {
      double s = 0.0;
      for (int64_t i = 0; i < k; ++i) s += p[i];
      mean = (float)(s / (double)k);
      for (int64_t i = 0; i < k; ++i) p[i] -= mean;
    }
    double q = 0.0;
    for (int64_t i = 0; i < k; ++i) q += (double)p[i] * (double)p[i];
    if (degenerate_pattern(q, metric == 0 && lo == hi)) {
      for (int64_t i = 0; i < k; ++i) p[i] = 0.f;
      continue;
    }
    const float norm = (float)sqrt(q);
    for (int64_t i = 0; i < k; ++i) p[i] /= norm;
  }
}

/* does candidate (s, i) rank before (t, j)?  score descending, index ascending */
static int before(float s, int64_t i, float t, int64_t j) { return s > t || (s == t && i < j); }

/* insert into a sorted best-list of length keep (worst entry last) */
static void insert(float *bs, int64_t *bi, int keep, float s, int64_t i) {
  if (!before(s, i, bs[keep - 1], bi[keep - 1])) return;
  int pos = keep - 1;
  while (pos > 0 && before(s, i, bs[pos - 1], bi[pos - 1])) {
    bs[pos] = bs[pos - 1];
    bi[pos] = bi[pos - 1];
    --pos;
  }
  bs[pos] = s;
  bi[pos] = i;
}

/* exp: m x k and dic: n x k, both already normalised.  For every experimental
 * pattern keep the `keep` best dictionary entries of THIS chunk merged into the
 * running lists scores/indices (m x keep), which the caller initialises with
 * (-INFINITY, INT64_MAX) before the first chunk.  index_base = chunk start. */
void kpdi_c_match_topk(const float *exp, const float *dic, int64_t m, int64_t n, int64_t k, int keep,
                       int64_t index_base, float *scores, int64_t *indices) {
  enum { BN = 64 };
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t r = 0; r < m; ++r) {
    const float *x = exp + r * k;
    float *bs = scores + r * keep;
    int64_t *bi = indices + r * keep;
    for (int64_t j0 = 0; j0 < n; j0 += BN) {
      const int64_t jn = (n - j0 < BN) ? n - j0 : BN;
      float acc[BN];
      for (int64_t j = 0; j < jn; ++j) {
        const float *y = dic + (j0 + j) * k;
        float a = 0.f;
        for (int64_t i = 0; i < k; ++i) a += x[i] * y[i];
        acc[j] = a;
      }
      for (int64_t j = 0; j < jn; ++j) insert(bs, bi, keep, acc[j], index_base + j0 + j);
    }
  }
}

/* ---- the TIMING variant of the sweep (bench.py's cpu_baseline, "c_openmp"): every host core, AVX2 + FMA,
 * a 4 x 4 register tile of dot products (16 vector accumulators), experimental rows in blocks of 16 so
 * that a group of 4 dictionary rows is used 4 times out of L1 while the dictionary streams through once
 * per block.  float32 accumulation in 8 partial sums per dot product (a different rounding order than
 * kpdi_c_match_topk: equally a float32 evaluation, within the 1e-5 contract). */
static inline float hsum8(__m256 v) {
  __m128 lo = _mm256_castps256_ps128(v), hi = _mm256_extractf128_ps(v, 1);
  lo = _mm_add_ps(lo, hi);
  lo = _mm_hadd_ps(lo, lo);
  lo = _mm_hadd_ps(lo, lo);
  return _mm_cvtss_f32(lo);
}

__attribute__((target("avx2,fma"))) static void tile_4x4(const float *x, const float *y, int64_t k, float *out) {
  __m256 a[16];
  for (int t = 0; t < 16; ++t) a[t] = _mm256_setzero_ps();
  int64_t i = 0;
  for (; i + 8 <= k; i += 8) {
    const __m256 x0 = _mm256_loadu_ps(x + i), x1 = _mm256_loadu_ps(x + k + i), x2 = _mm256_loadu_ps(x + 2 * k + i),
                 x3 = _mm256_loadu_ps(x + 3 * k + i);
    for (int j = 0; j < 4; ++j) {
      const __m256 yj = _mm256_loadu_ps(y + j * k + i);
      a[j] = _mm256_fmadd_ps(x0, yj, a[j]);
      a[4 + j] = _mm256_fmadd_ps(x1, yj, a[4 + j]);
      a[8 + j] = _mm256_fmadd_ps(x2, yj, a[8 + j]);
      a[12 + j] = _mm256_fmadd_ps(x3, yj, a[12 + j]);
    }
  }
  for (int t = 0; t < 16; ++t) out[t] = hsum8(a[t]);
  for (; i < k; ++i)
    for (int r = 0; r < 4; ++r)
      for (int j = 0; j < 4; ++j) out[4 * r + j] += x[r * k + i] * y[j * k + i];
}

/* exp: m x k, dic: n x k, both normalised, m % 4 == 0 not required (a scalar path takes the rest) */
void kpdi_c_match_topk_fast(const float *exp, const float *dic, int64_t m, int64_t n, int64_t k, int keep,
                            int64_t index_base, float *scores, int64_t *indices) {
  enum { EB = 16 };
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t r0 = 0; r0 < m; r0 += EB) {
    const int64_t rn = (m - r0 < EB) ? m - r0 : EB;
    for (int64_t j0 = 0; j0 < n; j0 += 4) {
      const int64_t jn = (n - j0 < 4) ? n - j0 : 4;
      for (int64_t q = 0; q < rn; q += 4) {
        const int64_t qn = (rn - q < 4) ? rn - q : 4;
        float out[16];
        if (qn == 4 && jn == 4) {
          tile_4x4(exp + (r0 + q) * k, dic + j0 * k, k, out);
        } else {
          for (int64_t r = 0; r < qn; ++r)
            for (int64_t j = 0; j < jn; ++j) {
              float a = 0.f;
              for (int64_t i = 0; i < k; ++i) a += exp[(r0 + q + r) * k + i] * dic[(j0 + j) * k + i];
              out[4 * r + j] = a;
            }
        }
        for (int64_t r = 0; r < qn; ++r)
          for (int64_t j = 0; j < jn; ++j)
            insert(scores + (r0 + q + r) * keep, indices + (r0 + q + r) * keep, keep, out[4 * r + j],
                   index_base + j0 + j);
      }
    }
  }
}

/* rows: n x k_in raw (float32) -> out: n x k, normalised in float32 like the reference's NumPy code
 * (mean, subtract, norm, divide), all cores. */
void kpdi_c_prepare_f32(const float *raw, int64_t n, int64_t k_in, const int64_t *pix_map, int64_t k, int metric,
                        float *out) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    const float *p = raw + r * k_in;
    float *o = out + r * k;
    double s = 0.0;
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t i = 0; i < k; ++i) {
      o[i] = p[pix_map ? pix_map[i] : i];
      s += o[i];
      lo = o[i] < lo ? o[i] : lo;
      hi = o[i] > hi ? o[i] : hi;
    }
    const float mean = metric == 0 ? (float)(s / (double)k) : 0.f;
    double q = 0.0;
    for (int64_t i = 0; i < k; ++i) {
      o[i] -= mean;
      q += (double)o[i] * o[i];
    }
    if (degenerate_pattern(q, metric == 0 && lo == hi)) {
      for (int64_t i = 0; i < k; ++i) o[i] = 0.f;
      continue;
    }
    const float inv = 1.f / (float)sqrt(q);
    for (int64_t i = 0; i < k; ++i) o[i] *= inv;
  }
}

/* Same sweep with every dot product accumulated in float64 and rounded once to float32: the
 * (almost) exact arbiter for row samples at BASELINE.json's full sizes (the reference's own
 * float32 sgemm and the engine's float32 MFMA chain both sit within a few 1e-7 of it for `ncc`).
 * `rows`: the n_rows experimental patterns (indices into exp) to evaluate; scores/indices are
 * n_rows x keep. */
void kpdi_c_match_topk_rows_f64(const float *exp, const int64_t *rows, int64_t n_rows, const float *dic, int64_t n,
                                int64_t k, int keep, int64_t index_base, float *scores, int64_t *indices) {
  enum { BN = 256 };
#pragma omp parallel for schedule(dynamic, 1) collapse(1)
  for (int64_t rr = 0; rr < n_rows; ++rr) {
    const float *x = exp + rows[rr] * k;
    float *bs = scores + rr * keep;
    int64_t *bi = indices + rr * keep;
    for (int64_t j0 = 0; j0 < n; j0 += BN) {
      const int64_t jn = (n - j0 < BN) ? n - j0 : BN;
      float acc[BN];
      for (int64_t j = 0; j < jn; ++j) {
        const float *y = dic + (j0 + j) * k;
        double a = 0.0;
#pragma omp simd reduction(+ : a)
        for (int64_t i = 0; i < k; ++i) a += (double)x[i] * (double)y[i];
        acc[j] = (float)a;
      }
      for (int64_t j = 0; j < jn; ++j) insert(bs, bi, keep, acc[j], index_base + j0 + j);
    }
  }
}

/* rows: n x k_in raw -> out: n x k kept pixels (pix_map[k], or NULL = all), normalised in float64
 * (mean, norm), stored as float32.  metric 0 = ncc, 1 = ndp. */
void kpdi_c_prepare_f64(const float *raw, int64_t n, int64_t k_in, const int64_t *pix_map, int64_t k, int metric,
                        float *out) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    const float *p = raw + r * k_in;
    float *o = out + r * k;
    double s = 0.0;
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t i = 0; i < k; ++i) {
      const float x = p[pix_map ? pix_map[i] : i];
      s += x;
      lo = x < lo ? x : lo;
      hi = x > hi ? x : hi;
    }
    const double mean = metric == 0 ? s / (double)k : 0.0;
    double q = 0.0;
    for (int64_t i = 0; i < k; ++i) {
      const double d = (double)p[pix_map ? pix_map[i] : i] - mean;
      q += d * d;
    }
    if (degenerate_pattern(q, metric == 0 && lo == hi)) {
      for (int64_t i = 0; i < k; ++i) o[i] = 0.f;
      continue;
    }
    const double norm = sqrt(q);
    for (int64_t i = 0; i < k; ++i) o[i] = (float)(((double)p[pix_map ? pix_map[i] : i] - mean) / norm);
  }
}

void kpdi_c_init_topk(float *scores, int64_t *indices, int64_t count) {
  for (int64_t i = 0; i < count; ++i) {
    scores[i] = -INFINITY;
    indices[i] = INT64_MAX;
  }
}
