// prep.hip - pattern preparation: what SimilarityMetric.prepare_experimental /
// prepare_dictionary do before match():
//   cast to float32 -> drop masked-out patterns (navigation mask) -> drop
//   masked-out pixels (signal mask) -> NCC: subtract the mean, divide by the L2
//   norm; NDP: divide by the L2 norm only.
// Reference: similarity_metrics/_normalized_cross_correlation.py:88-159, :228-241
//            similarity_metrics/_normalized_dot_product.py:80-150, :181-194
//
// K <= 4096 kept pixels (up to 64x64 detectors): one WAVE per pattern, values
// in registers, shuffle reductions, vector loads/stores when unmasked.  Larger
// detectors: one workgroup per pattern re-reading the pattern from L2.  Output:
// the K-padded f32 row (zero tail), scattered into the tiled/swizzled layout
// match.hip streams (kernels.h: prepared_offset; 128-byte pieces = full lines).  HBM-bound: algorithmic bytes =
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

__device__ __forceinline__ size_t out_offset(int exp_layout, int r, int c, int nslab) {
  return exp_layout ? prepared_exp_offset(r, c, nslab) : prepared_offset(r, c, nslab);
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
                                                            int metric, int exp_layout, float *out) {
  __shared__ float red[PREP_THREADS / 64];
  const int r = blockIdx.x;
  const int64_t src = row_map ? row_map[r] : r;
  const T *p = raw + src * (int64_t)npix;
  const int nslab = kpad / TILE_K;
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
      if (c < kpad) out[out_offset(exp_layout, r, c, nslab)] = (c < k) ? v[i] * inv : 0.f;
    }
    for (int c = tid + PREP_VPT * PREP_THREADS; c < kpad; c += PREP_THREADS)
      out[out_offset(exp_layout, r, c, nslab)] = 0.f;
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
      out[out_offset(exp_layout, r, c, nslab)] = (c < k) ? ((float)p[pix_map ? pix_map[c] : c] - mean) * inv : 0.f;
  }
}

// ---- fast path: ONE WAVE per pattern, K <= 4096 kept pixels (up to 64x64 detectors).
// All of a lane's loads are issued before the first use (64 values in VGPRs), the two
// reductions are wave shuffles (no LDS, no barrier), 4 patterns per workgroup.
// VEC = 4: no signal mask and K % 4 == 0 -> 4-element vector loads / float4 stores.
template <typename T>
struct alignas(sizeof(T) * 4) Quad {
  T v[4];
};

template <typename T, int VEC>
__global__ __launch_bounds__(PREP_THREADS) void prep_wave_kernel(const T *raw, int npix, const int *row_map,
                                                                 const int *pix_map, int k, int kpad,
                                                                 int metric, int exp_layout, int n_out, float *out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * (PREP_THREADS / 64) + (threadIdx.x >> 6);
  if (r >= n_out) return;
  const int64_t src = row_map ? row_map[r] : r;
  const T *p = raw + src * (int64_t)npix;
  const int nslab = kpad / TILE_K;
  constexpr int N = 64;  // values per lane
  float v[N];
  float s = 0.f;
  if (VEC == 4) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
      const int c = 4 * (lane + 64 * i);
      Quad<T> q;
      q.v[0] = q.v[1] = q.v[2] = q.v[3] = (T)0;
      if (c < k) q = *reinterpret_cast<const Quad<T> *>(p + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[4 * i + e] = (float)q.v[e];
        s += v[4 * i + e];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int c = lane + 64 * i;
      v[i] = 0.f;
      if (c < k) v[i] = (float)p[pix_map ? pix_map[c] : c];
      s += v[i];
    }
  }
  float mean = 0.f;
  if (metric == KPDI_METRIC_NCC) mean = wave_sum(s) / (float)k;
  float q2 = 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int c = VEC == 4 ? 4 * (lane + 64 * (i / 4)) + (i & 3) : lane + 64 * i;
    if (c < k) {
      v[i] -= mean;
      q2 += v[i] * v[i];
    } else {
      v[i] = 0.f;
    }
  }
  const float norm = sqrtf(wave_sum(q2));
  const float inv = norm > 0.f ? 1.f / norm : 0.f;
  if (VEC == 4) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
      const int c = 4 * (lane + 64 * i);
      if (c < kpad) {
        float4 w;
        w.x = v[4 * i] * inv;
        w.y = v[4 * i + 1] * inv;
        w.z = v[4 * i + 2] * inv;
        w.w = v[4 * i + 3] * inv;
        *reinterpret_cast<float4 *>(out + out_offset(exp_layout, r, c, nslab)) = w;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int c = lane + 64 * i;
      if (c < kpad) out[out_offset(exp_layout, r, c, nslab)] = v[i] * inv;
    }
  }
}

hipError_t launch_prep(const PrepLaunch &a, hipStream_t s) {
  if (a.n_out <= 0) return hipSuccess;
  const bool wave_path = a.k <= 4096;
  const bool vec4 = wave_path && a.pix_map == nullptr && (a.k % 4) == 0 && (a.npix % 4) == 0 &&
                    ((uintptr_t)a.raw % (4 * dtype_size(a.dtype))) == 0;
  dim3 block(PREP_THREADS);
  dim3 grid(wave_path ? (a.n_out + 3) / 4 : a.n_out);
#define KPDI_PREP(T)                                                                                   \
  if (vec4)                                                                                            \
    hipLaunchKernelGGL((prep_wave_kernel<T, 4>), grid, block, 0, s, (const T *)a.raw, a.npix,         \
                       a.row_map, a.pix_map, a.k, a.kpad, a.metric, a.exp_layout, a.n_out, a.out);                   \
  else if (wave_path)                                                                                  \
    hipLaunchKernelGGL((prep_wave_kernel<T, 1>), grid, block, 0, s, (const T *)a.raw, a.npix,         \
                       a.row_map, a.pix_map, a.k, a.kpad, a.metric, a.exp_layout, a.n_out, a.out);                   \
  else                                                                                                 \
    hipLaunchKernelGGL((prep_kernel<T>), grid, block, 0, s, (const T *)a.raw, a.npix, a.row_map,      \
                       a.pix_map, a.k, a.kpad, a.metric, a.exp_layout, a.out);                                       \
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
