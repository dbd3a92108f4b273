// prep.hip - pattern preparation: what SimilarityMetric.prepare_experimental /
// prepare_dictionary do before match():
//   cast to float32 -> drop masked-out patterns (navigation mask) -> drop
//   masked-out pixels (signal mask) -> NCC: subtract the mean, divide by the L2
//   norm; NDP: divide by the L2 norm only.
// Reference: similarity_metrics/_normalized_cross_correlation.py:88-159, :228-241
//            similarity_metrics/_normalized_dot_product.py:80-150, :181-194
//
// One workgroup per output pattern.  The kept pixels are gathered once into
// registers (K <= 4096, i.e. up to 64x64 detectors) or re-read from L2 (larger
// detectors), reduced with wave shuffles, and written as one K-padded f32 row
// (zero tail) of the matrix match.hip streams.  HBM-bound: algorithmic bytes =
// npix*sizeof(in) read + kpad*4 written per pattern.
//
// A pattern with zero norm (constant pattern; 0/0 = NaN in the reference, out of
// contract per SURVEY.md 8(a)) becomes an all-zero row: it scores exactly 0
// against everything.
#include "kernels.h"
#include "../../include/kpdi.h"

namespace kpdi {

size_t dtype_size(int dtype) {
  switch (dtype) {
    case KPDI_U8: case KPDI_I8: return 1;
    case KPDI_U16: case KPDI_I16: return 2;
    case KPDI_F32: case KPDI_I32: case KPDI_U32: return 4;
    case KPDI_F64: return 8;
  }
  return 0;
}

constexpr int PREP_THREADS = 256;
constexpr int PREP_VPT = 16;  // values per thread held in registers

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float block_sum(float v, float *red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();  // protect `red` from the previous use
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < PREP_THREADS / 64; ++i) t += red[i];
  return t;
}

template <typename T>
__global__ __launch_bounds__(PREP_THREADS) void prep_kernel(const T *raw, int npix, const int *row_map,
                                                            const int *pix_map, int k, int kpad,
                                                            int metric, float *out) {
  __shared__ float red[PREP_THREADS / 64];
  const int r = blockIdx.x;
  const int64_t src = row_map ? row_map[r] : r;
  const T *p = raw + src * (int64_t)npix;
  float *o = out + (int64_t)r * kpad;
  const int tid = threadIdx.x;

  if (k <= PREP_THREADS * PREP_VPT) {
    float v[PREP_VPT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PREP_VPT; ++i) {
      const int c = tid + i * PREP_THREADS;
      v[i] = 0.f;
      if (c < k) v[i] = (float)p[pix_map ? pix_map[c] : c];
      s += v[i];
    }
    float mean = 0.f;
    if (metric == KPDI_METRIC_NCC) mean = block_sum(s, red) / (float)k;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PREP_VPT; ++i) {
      const int c = tid + i * PREP_THREADS;
      if (c < k) {
        v[i] -= mean;
        q += v[i] * v[i];
      }
    }
    const float norm = sqrtf(block_sum(q, red));
    const float inv = norm > 0.f ? 1.f / norm : 0.f;
#pragma unroll
    for (int i = 0; i < PREP_VPT; ++i) {
      const int c = tid + i * PREP_THREADS;
      if (c < kpad) o[c] = (c < k) ? v[i] * inv : 0.f;
    }
    for (int c = tid + PREP_VPT * PREP_THREADS; c < kpad; c += PREP_THREADS) o[c] = 0.f;
  } else {
    float s = 0.f;
    for (int c = tid; c < k; c += PREP_THREADS) s += (float)p[pix_map ? pix_map[c] : c];
    float mean = 0.f;
    if (metric == KPDI_METRIC_NCC) mean = block_sum(s, red) / (float)k;
    float q = 0.f;
    for (int c = tid; c < k; c += PREP_THREADS) {
      const float d = (float)p[pix_map ? pix_map[c] : c] - mean;
      q += d * d;
    }
    const float norm = sqrtf(block_sum(q, red));
    const float inv = norm > 0.f ? 1.f / norm : 0.f;
    for (int c = tid; c < kpad; c += PREP_THREADS)
      o[c] = (c < k) ? ((float)p[pix_map ? pix_map[c] : c] - mean) * inv : 0.f;
  }
}

hipError_t launch_prep(const PrepLaunch &a, hipStream_t s) {
  if (a.n_out <= 0) return hipSuccess;
  dim3 grid(a.n_out), block(PREP_THREADS);
#define KPDI_PREP(T)                                                                             \
  hipLaunchKernelGGL((prep_kernel<T>), grid, block, 0, s, (const T *)a.raw, a.npix, a.row_map, \
                     a.pix_map, a.k, a.kpad, a.metric, a.out);                                   \
  break;
  switch (a.dtype) {
    case KPDI_U8: KPDI_PREP(uint8_t)
    case KPDI_I8: KPDI_PREP(int8_t)
    case KPDI_U16: KPDI_PREP(uint16_t)
    case KPDI_I16: KPDI_PREP(int16_t)
    case KPDI_I32: KPDI_PREP(int32_t)
    case KPDI_U32: KPDI_PREP(uint32_t)
    case KPDI_F32: KPDI_PREP(float)
    case KPDI_F64: KPDI_PREP(double)
    default: return hipErrorInvalidValue;
  }
#undef KPDI_PREP
  return hipGetLastError();
}

}  // namespace kpdi
