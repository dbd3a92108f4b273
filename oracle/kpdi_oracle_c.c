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
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* rows: n x k, in place.  metric 0 = ncc (subtract mean first), 1 = ndp. */
void kpdi_c_normalize(float *rows, int64_t n, int64_t k, int metric) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    float *p = rows + r * k;
    if (metric == 0) {
      double s = 0.0;
      for (int64_t i = 0; i < k; ++i) s += p[i];
      const float mean = (float)(s / (double)k);
      for (int64_t i = 0; i < k; ++i) p[i] -= mean;
    }
    double q = 0.0;
    for (int64_t i = 0; i < k; ++i) q += (double)p[i] * (double)p[i];
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
    for (int64_t i = 0; i < k; ++i) s += p[pix_map ? pix_map[i] : i];
    const double mean = metric == 0 ? s / (double)k : 0.0;
    double q = 0.0;
    for (int64_t i = 0; i < k; ++i) {
      const double d = (double)p[pix_map ? pix_map[i] : i] - mean;
      q += d * d;
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
