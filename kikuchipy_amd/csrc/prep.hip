// prep.hip - pattern preparation: what SimilarityMetric.prepare_experimental /
// prepare_dictionary do before match():
//   cast to float32 -> drop masked-out patterns (navigation mask) -> drop
//   masked-out pixels (signal mask) -> NCC: subtract the mean, divide by the L2
//   norm; NDP: divide by the L2 norm only.
// Reference: similarity_metrics/_normalized_cross_correlation.py:88-159, :228-241
//            similarity_metrics/_normalized_dot_product.py:80-150, :181-194
//
// Output: the K-padded f32 row (zero tail) of every pattern, scattered into the
// tiled / swizzled layout match.hip streams (kernels.h: prepared_offset; the pieces of
// one pattern are full 128-byte lines).  HBM-bound: algorithmic bytes per pattern =
// npix*sizeof(in) read + kpad*4 written.
//
//   K <= 4096 kept pixels (up to 64x64 detectors): ONE WAVE per pattern, 64 values per
//   lane in VGPRs, all loads in flight before the first use, shuffle reductions.
//     unmasked:  4-element vector loads straight to registers      (prep_wave_kernel<T,4>)
//     masked:    the raw row is staged in LDS with vector loads, the kept pixels are
//                gathered from LDS through the LDS copy of the pixel map
//                                                                   (prep_wave_masked_kernel)
//     otherwise: scalar gather from global memory                   (prep_wave_kernel<T,1>)
//   K <= 16384 (up to 128x128): ONE WORKGROUP per pattern, the same scheme with 256 threads
//                                                                   (prep_block_kernel)
//   larger still: one workgroup per pattern re-reading the pattern from L2 (prep_kernel).
//
// NDP is evaluated in a centred form (`metric` = NORM_NDP_CENTRED inside this file): with the
// row mean mu and d = x - mu,  sum x y = sum d_x d_y + K mu_x mu_y,  so the prepared row holds
// d / ||x|| in its K columns and sqrt(K) mu / ||x|| in column K (the product of two such entries is
// the second term).  The match kernel then accumulates small numbers of both signs and adds the one
// large term last, instead of ~K/2 roundings at the magnitude of the score: against exact float64
// scores `ndp` drops from ~3e-6 (1e-5 on integer x integer pattern pairs, whose equal increments
// round with a systematic bias) to the 1e-7 of `ncc`.  Not in the float16 form, whose 11 bits
// cannot carry the constant.
//
// A DEGENERATE pattern - zero variance (`ncc`) / all zeros (`ndp`) / NaN or inf among its kept pixels; 0/0 = NaN in
// the reference - becomes an all-zero row: it scores exactly 0 against everything (prep_device.h: degenerate_pattern,
// include/kpdi.h "Degenerate patterns").
#include "prep_device.h"

#include <algorithm>
#include <cstring>

namespace kpdi {

size_t dtype_size(int dtype) {
  switch (dtype) {
    case KPDI_U8: case KPDI_I8: return 1;
    case KPDI_U16: case KPDI_I16: case KPDI_F16: return 2;
    case KPDI_F32: case KPDI_I32: case KPDI_U32: return 4;
    case KPDI_F64: return 8;
  }
  return 0;
}

// ---- large detectors: one workgroup per pattern ----------------------------------------
template <typename T>
__global__ __launch_bounds__(PREP_THREADS) void prep_kernel(const T *raw, int npix, const int *row_map,
                                                            const int *pix_map, int k, int kpad,
                                                            int metric, float *out, int form) {
  __shared__ float red[PREP_THREADS / 64];
  const int r = blockIdx.x;
  const int64_t src = row_map ? row_map[r] : r;
  const T *p = raw + src * (int64_t)npix;
  const int nslab = kpad / TILE_K;
  const int tid = threadIdx.x;

  float s = 0.f, lo = INFINITY, hi = -INFINITY;
  for (int c = tid; c < k; c += PREP_THREADS) {
    const float x = (float)p[pix_map ? pix_map[c] : c];
    s += x;
    lo = fminf(lo, x);
    hi = fmaxf(hi, x);
  }
  float mean = 0.f;
  if (metric != KPDI_METRIC_NDP) mean = block_sum(s, red) / (float)k;
  if (metric == KPDI_METRIC_NCC) block_minmax_n<PREP_THREADS>(lo, hi, red);
  float q = 0.f;
  for (int c = tid; c < k; c += PREP_THREADS) {
    const float d = (float)p[pix_map ? pix_map[c] : c] - mean;
    q += d * d;
  }
  q = block_sum(q, red);
  const bool centred = metric == NORM_NDP_CENTRED;
  const float norm2 = centred ? q + (float)k * mean * mean : q;
  const bool degenerate = degenerate_pattern(norm2, lo, hi, metric == KPDI_METRIC_NCC);  // -> an all-zero row (prep_device.h)
  const float inv = degenerate ? 0.f : 1.f / sqrtf(norm2);
  const float cval = degenerate ? 0.f : sqrtf((float)k) * mean * inv;
  if (degenerate) mean = 0.f;
  auto value = [&](int c) { return degenerate ? 0.f : ((float)p[pix_map ? pix_map[c] : c] - mean) * inv; };
  if ((form & 0xff) == 2) {
    for (int c = tid; c < 2 * kpad; c += PREP_THREADS)
      *(_Float16 *)half_slot(out, r, c, kpad, form) = (_Float16)((c < k) ? value(c) * 4096.f : 0.f);
    return;
  }
  if ((form & 0xff) == 3) {
    for (int c = tid; c < kpad; c += PREP_THREADS)
      *(float *)half_slot(out, r, 2 * c, kpad, form) = (c < k) ? value(c) : ((centred && c == k) ? cval : 0.f);
    return;
  }
  for (int c = tid; c < kpad; c += PREP_THREADS)
    out[prepared_offset(r, c, nslab)] = (c < k) ? value(c) : ((centred && c == k) ? cval : 0.f);
}

// ---- shared tail of the wave-per-pattern kernels: v[i] holds kept pixel lane + 64*i -----
__device__ __forceinline__ void normalise_and_store(float (&v)[WAVE_VALUES], float s, int lane, int r, int k,
                                                    int kpad, int metric, float *out, int form) {
  const int nslab = kpad / TILE_K;
  float mean = 0.f;
  if (metric != KPDI_METRIC_NDP) mean = wave_sum(s) / (float)k;
  float q2 = 0.f, lo = INFINITY, hi = -INFINITY;
#pragma unroll
  for (int i = 0; i < WAVE_VALUES; ++i) {
    const int c = lane + 64 * i;
    if (c < k) {
      lo = fminf(lo, v[i]);
      hi = fmaxf(hi, v[i]);
      v[i] -= mean;
      q2 += v[i] * v[i];
    } else {
      v[i] = 0.f;
    }
  }
  q2 = wave_sum(q2);
  if (metric == KPDI_METRIC_NCC) {
    lo = wave_min(lo);
    hi = wave_max(hi);
  }
  const bool centred = metric == NORM_NDP_CENTRED;
  const float norm2 = centred ? q2 + (float)k * mean * mean : q2;
  const bool degenerate = degenerate_pattern(norm2, lo, hi, metric == KPDI_METRIC_NCC);  // -> an all-zero row (prep_device.h)
  const float inv = degenerate ? 0.f : 1.f / sqrtf(norm2);
  const float cval = degenerate ? 0.f : sqrtf((float)k) * mean * inv;
  if (degenerate) {
#pragma unroll
    for (int i = 0; i < WAVE_VALUES; ++i) v[i] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < WAVE_VALUES; ++i) {
    const int c = lane + 64 * i;
    if ((form & 0xff) == 2) {
      if (c < 2 * kpad) *(_Float16 *)half_slot(out, r, c, kpad, form) = (_Float16)(v[i] * inv * 4096.f);
    } else if ((form & 0xff) == 3) {
      if (c < kpad) *(float *)half_slot(out, r, 2 * c, kpad, form) = (centred && c == k) ? cval : v[i] * inv;
    } else if (c < kpad) {
      out[prepared_offset(r, c, nslab)] = (centred && c == k) ? cval : v[i] * inv;
    }
  }
}

// ---- one wave per pattern, values straight from global memory ---------------------------
// VEC = 4: no signal mask and K % 4 == 0 -> 4-element vector loads / float4 stores.
// VEC = 1: scalar gather through the pixel map.
template <typename T, int VEC, bool H16>
__global__ __launch_bounds__(PREP_THREADS) void prep_wave_kernel(const T *raw, int npix, const int *row_map,
                                                                 const int *pix_map, int k, int kpad,
                                                                 int metric, int n_out, float *out, int split) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * (PREP_THREADS / 64) + (threadIdx.x >> 6);
  if (r >= n_out) return;
  const int64_t src = row_map ? row_map[r] : r;
  const T *p = raw + src * (int64_t)npix;
  float v[WAVE_VALUES];
  float s = 0.f;
  if (VEC == 4) {
#pragma unroll
    for (int i = 0; i < WAVE_VALUES / 4; ++i) {
      const int c = 4 * (lane + 64 * i);
      Quad<T> q;
      q.v[0] = q.v[1] = q.v[2] = q.v[3] = (T)0;
      if (c < k) q = load_quad(p + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[4 * i + e] = (float)q.v[e];
        s += v[4 * i + e];
      }
    }
    normalise_and_store_quads<64, WAVE_VALUES, H16>(v, s, lane, r, k, kpad, metric, out, split);
  } else {
#pragma unroll
    for (int i = 0; i < WAVE_VALUES; ++i) {
      const int c = lane + 64 * i;
      v[i] = 0.f;
      if (c < k) v[i] = (float)p[pix_map ? pix_map[c] : c];
      s += v[i];
    }
    normalise_and_store(v, s, lane, r, k, kpad, metric, out, split);
  }
}

// ---- 4096 < K <= 16384 kept pixels (up to 128x128 detectors): one workgroup per pattern ----
// The same register-resident scheme with 256 threads x 64 values: every pixel is read once
// (vector loads when there is no signal mask, else a gather through the pixel map - the row
// is then served by L2 after its first touch) and stored once as whole 16-byte slots.
//
// Plane-major forms (H16): a pattern owns 32 bytes of every 128-byte line it touches, the other three quarters belong
// to its three neighbours in the tile.  `n_affine` > 0 makes those four workgroups neighbours in TIME and PLACE: block b
// runs on XCD b % 8 (observed, used for speed only), so pattern 4 g + j goes to block ((g / 8) * 4 + j) * 8 + g % 8 -
// the four quarters of a line then meet in ONE XCD's L2 within microseconds and leave it as a whole line.
template <typename T, bool MASKED, bool H16>
__global__ __launch_bounds__(PREP_THREADS) void prep_block_kernel(const T *raw, int npix, const int *row_map,
                                                                  const int *pix_map, int k, int kpad,
                                                                  int metric, float *out, int split, int n_affine) {
  __shared__ float red[PREP_THREADS / 64];
  const int tid = threadIdx.x;
  int r = blockIdx.x;
  if (H16 && n_affine > 0) {
    const int x = r & 7, q = r >> 3;
    r = 4 * ((q >> 2) * 8 + x) + (q & 3);
    if (r >= n_affine) return;
  }
  const int64_t src = row_map ? row_map[r] : r;
  const T *p = raw + src * (int64_t)npix;
  float v[WAVE_VALUES];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < WAVE_VALUES / 4; ++i) {
    const int c = 4 * (tid + PREP_THREADS * i);
    if (!MASKED) {
      Quad<T> q;
      q.v[0] = q.v[1] = q.v[2] = q.v[3] = (T)0;
      if (c < k) q = load_quad(p + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[4 * i + e] = (float)q.v[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[4 * i + e] = c + e < k ? (float)p[pix_map[c + e]] : 0.f;
    }
    s += (v[4 * i] + v[4 * i + 1]) + (v[4 * i + 2] + v[4 * i + 3]);
  }
  normalise_and_store_quads<PREP_THREADS, WAVE_VALUES, H16>(v, s, tid, r, k, kpad, metric, out, split, red);
}

// ---- float16 form, 4096 < K <= 16384: FOUR patterns per 1024-thread workgroup ------------------
// In the float16 layout (prep_device.h: half_slot) a pattern owns 32 contiguous bytes per 16-pixel plane and
// four consecutive patterns share each 128-byte line.  One workgroup per pattern (prep_block_kernel) writes
// 32-byte pieces of lines whose other pieces arrive from other workgroups at other times: 1.9 TB/s on the
// 28.8 GB dictionary of configs[4].  Here every 256-thread group normalises one of four consecutive patterns
// in registers (as prep_block_kernel), the float16 rows are staged in LDS, and the workgroup writes them out
// as whole lines.  LDS: 4 x (2 * kpad + 8) float16.
constexpr int PREP16_THREADS = 1024;
// NP = patterns per workgroup.  4 (1024 threads): whole 128-byte lines, but 64 values per thread x 1024 threads is all of
// a CU's registers - ONE workgroup per CU, whose load, reduce and store phases nothing overlaps (3.7 TB/s on configs[4]).
// 2 (512 threads): two workgroups per CU in different phases; a plane's 64 bytes of the two rows are one aligned half
// line = one HBM burst, and the workgroup holding the other half runs on the same XCD at the same time (`affine`: block b
// runs on XCD b % 8, so pair 2 G + j of row group G goes to block ((G / 8) * 2 + j) * 8 + G % 8), i.e. the halves meet in
// that XCD's L2.
template <typename T, bool MASKED, int NP>
__global__ __launch_bounds__(256 * NP) void prep16_block4_kernel(const T *raw, int npix, const int *row_map,
                                                                 const int *pix_map, int k, int kpad,
                                                                 int metric, int n_out, float *out, int form) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ float red[4 * NP];
  constexpr int THREADS = 256 * NP;
  const int tid = threadIdx.x, g = tid >> 8, t = tid & 255, wave = tid >> 6;
  int unit = blockIdx.x;  // group of NP consecutive patterns
  if (NP == 2) {
    const int x = unit & 7, q = unit >> 3;
    unit = 2 * ((q >> 1) * 8 + x) + (q & 1);
  }
  const int r0 = unit * NP;
  if (r0 >= n_out) return;  // (whole workgroup)
  const int r = r0 + g;
  const bool live = r < n_out;
  const int row_halves = 2 * kpad + 8;  // + 16 bytes: the rows start in different banks
  _Float16 *stage = (_Float16 *)smem_raw + (size_t)g * row_halves;
  // a thread's 64 values are VW consecutive pixels per load: 4 (one 4- / 8- / 16-byte load), or 8 for 2-byte raw types
  // without a mask (a float16-resident dictionary: 16-byte loads as for float32 rows - with 8-byte loads the kernel
  // read half the bytes in MORE time) - value idx of a thread is pixel VW * (t + 256 * (idx / VW)) + idx % VW
  constexpr int VW = (!MASKED && sizeof(T) == 2) ? 8 : 4;
  float v[WAVE_VALUES];
  float s = 0.f;
  if (live && VW == 8) {
    const int64_t src = row_map ? row_map[r] : r;
    const T *p = raw + src * (int64_t)npix;
#pragma unroll
    for (int j = 0; j < WAVE_VALUES / 8; ++j) {
      const int c = 8 * (t + 256 * j);
      Quad<T> q0, q1;
      q0.v[0] = q0.v[1] = q0.v[2] = q0.v[3] = q1.v[0] = q1.v[1] = q1.v[2] = q1.v[3] = (T)0;
      if (c + 8 <= k) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p + c));
        __builtin_memcpy(&q0, &x, 8);
        __builtin_memcpy(&q1, (const char *)&x + 8, 8);
      } else {  // (k % 8 == 4: the last quad of the row)
        if (c < k) q0 = load_quad(p + c);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[8 * j + e] = (float)q0.v[e];
        v[8 * j + 4 + e] = (float)q1.v[e];
      }
      s += ((v[8 * j] + v[8 * j + 1]) + (v[8 * j + 2] + v[8 * j + 3])) + ((v[8 * j + 4] + v[8 * j + 5]) + (v[8 * j + 6] + v[8 * j + 7]));
    }
  } else if (live) {
    const int64_t src = row_map ? row_map[r] : r;
    const T *p = raw + src * (int64_t)npix;
#pragma unroll
    for (int i = 0; i < WAVE_VALUES / 4; ++i) {
      const int c = 4 * (t + 256 * i);
      if (!MASKED) {
        Quad<T> q;
        q.v[0] = q.v[1] = q.v[2] = q.v[3] = (T)0;
        if (c < k) q = load_quad(p + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * i + e] = (float)q.v[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * i + e] = c + e < k ? (float)p[pix_map[c + e]] : 0.f;
      }
      s += (v[4 * i] + v[4 * i + 1]) + (v[4 * i + 2] + v[4 * i + 3]);
    }
  }
  // sums over the 4 waves of a group
  auto group_total = [&](float x) {
    x = wave_sum(x);
    __syncthreads();
    if ((tid & 63) == 0) red[wave] = x;
    __syncthreads();
    return (red[4 * g] + red[4 * g + 1]) + (red[4 * g + 2] + red[4 * g + 3]);
  };
  // minimum / maximum over the 4 waves of a group (the exact test for a constant pattern, prep_device.h)
  auto group_minmax = [&](float &lo, float &hi) {
    lo = wave_min(lo);
    hi = wave_max(hi);
    __syncthreads();
    if ((tid & 63) == 0) red[wave] = lo;
    __syncthreads();
    lo = fminf(fminf(red[4 * g], red[4 * g + 1]), fminf(red[4 * g + 2], red[4 * g + 3]));
    __syncthreads();
    if ((tid & 63) == 0) red[wave] = hi;
    __syncthreads();
    hi = fmaxf(fmaxf(red[4 * g], red[4 * g + 1]), fmaxf(red[4 * g + 2], red[4 * g + 3]));
  };
  float mean = 0.f;
  if (metric != KPDI_METRIC_NDP) mean = group_total(s) / (float)k;
  float q2 = 0.f, lo = INFINITY, hi = -INFINITY;
#pragma unroll
  for (int i = 0; i < WAVE_VALUES; ++i) {
    const int c = VW * (t + 256 * (i / VW)) + (i % VW);
    if (c < k) {
      lo = fminf(lo, v[i]);
      hi = fmaxf(hi, v[i]);
      v[i] -= mean;
      q2 += v[i] * v[i];
    } else {
      v[i] = 0.f;
    }
  }
  q2 = group_total(q2);
  if (metric == KPDI_METRIC_NCC) group_minmax(lo, hi);
  const bool degenerate = degenerate_pattern(q2, lo, hi, metric == KPDI_METRIC_NCC);  // -> an all-zero row (prep_device.h)
  const float inv = degenerate ? 0.f : 4096.f / sqrtf(q2);  // float16 operands are stored scaled by 2^12
  if (degenerate) {
#pragma unroll
    for (int i = 0; i < WAVE_VALUES; ++i) v[i] = 0.f;
  }
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int i = 0; i < WAVE_VALUES / 4; ++i) {
    const int c = VW * (t + 256 * ((4 * i) / VW)) + (4 * i) % VW;
    if (c < 2 * kpad) {
      h4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = (_Float16)(v[4 * i + e] * inv);
      *reinterpret_cast<h4 *>(stage + c) = h;
    }
  }
  __syncthreads();
  // ---- write-out: plane P of the NP rows = NP x 32 contiguous bytes (rows r0 .. of a tile, 32 bytes each, the two
  // 16-byte halves of a row swapped when its bit 3 is set - the same for all of them); 4 threads per row, 8 bytes each
  const int lr = (form >> 8) & 0xff, bk = (form >> 16) & 0xff;
  const unsigned magic = bk == 48 ? 89478486u : 134217728u;
  const int nsteps = (int)__umulhi(2u * (unsigned)kpad, magic);
  const int row0 = r0 & ((1 << lr) - 1);
  const int swz = (row0 >> 3) & 1;
  const int planes = (2 * kpad) / 16, planes_per_step = bk / 16;
  const int j = tid & (4 * NP - 1), row = j >> 2, piece = j & 3;  // piece: 8 bytes = 4 pixels of the row's 32 bytes
  const int px = (((piece >> 1) ^ swz) << 3) + ((piece & 1) << 2);
  for (int P = tid / (4 * NP); P < planes; P += THREADS / (4 * NP)) {
    const int step = (int)__umulhi((unsigned)(16 * P), magic);
    const int pl = P - step * planes_per_step;
    const size_t block = (size_t)(r0 >> lr) * nsteps + step;
    char *line = (char *)out + ((block * bk) << (lr + 1)) + (((size_t)pl << lr) + row0) * 32;
    // rows beyond n_out hold zeros (their group staged zeros): keeps the line whole
    const _Float16 *srcp = (const _Float16 *)smem_raw + (size_t)row * row_halves + 16 * P + px;
    store_out(reinterpret_cast<h4 *>(line + 32 * row + 8 * piece), *reinterpret_cast<const h4 *>(srcp));
  }
}

// ---- the same for the float32 plane-major form (form 3): a row of 16 384 float32 does not fit LDS four times, so the
// normalised rows (in registers) are staged and written HALF a row at a time (4 x kpad / 2 floats of LDS).
template <typename T, bool MASKED>
__global__ __launch_bounds__(PREP16_THREADS) void prep32_block4_kernel(const T *raw, int npix, const int *row_map,
                                                                       const int *pix_map, int k, int kpad,
                                                                       int metric, int n_out, float *out, int form) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ float red[PREP16_THREADS / 64];
  const int tid = threadIdx.x, g = tid >> 8, t = tid & 255, wave = tid >> 6;
  const int r = blockIdx.x * 4 + g;
  const bool live = r < n_out;
  float v[WAVE_VALUES];
  float s = 0.f;
  if (live) {
    const int64_t src = row_map ? row_map[r] : r;
    const T *p = raw + src * (int64_t)npix;
#pragma unroll
    for (int i = 0; i < WAVE_VALUES / 4; ++i) {
      const int c = 4 * (t + 256 * i);
      if (!MASKED) {
        Quad<T> q;
        q.v[0] = q.v[1] = q.v[2] = q.v[3] = (T)0;
        if (c < k) q = load_quad(p + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * i + e] = (float)q.v[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * i + e] = c + e < k ? (float)p[pix_map[c + e]] : 0.f;
      }
      s += (v[4 * i] + v[4 * i + 1]) + (v[4 * i + 2] + v[4 * i + 3]);
    }
  }
  auto group_total = [&](float x) {
    x = wave_sum(x);
    __syncthreads();
    if ((tid & 63) == 0) red[wave] = x;
    __syncthreads();
    return (red[4 * g] + red[4 * g + 1]) + (red[4 * g + 2] + red[4 * g + 3]);
  };
  // minimum / maximum over the 4 waves of a group (the exact test for a constant pattern, prep_device.h)
  auto group_minmax = [&](float &lo, float &hi) {
    lo = wave_min(lo);
    hi = wave_max(hi);
    __syncthreads();
    if ((tid & 63) == 0) red[wave] = lo;
    __syncthreads();
    lo = fminf(fminf(red[4 * g], red[4 * g + 1]), fminf(red[4 * g + 2], red[4 * g + 3]));
    __syncthreads();
    if ((tid & 63) == 0) red[wave] = hi;
    __syncthreads();
    hi = fmaxf(fmaxf(red[4 * g], red[4 * g + 1]), fmaxf(red[4 * g + 2], red[4 * g + 3]));
  };
  float mean = 0.f;
  if (metric != KPDI_METRIC_NDP) mean = group_total(s) / (float)k;
  float q2 = 0.f, lo = INFINITY, hi = -INFINITY;
#pragma unroll
  for (int i = 0; i < WAVE_VALUES; ++i) {
    const int c = 4 * (t + 256 * (i / 4)) + (i & 3);
    if (c < k) {
      lo = fminf(lo, v[i]);
      hi = fmaxf(hi, v[i]);
      v[i] -= mean;
      q2 += v[i] * v[i];
    } else {
      v[i] = 0.f;
    }
  }
  q2 = group_total(q2);
  if (metric == KPDI_METRIC_NCC) group_minmax(lo, hi);
  const bool centred = metric == NORM_NDP_CENTRED;
  const float norm2 = centred ? q2 + (float)k * mean * mean : q2;
  const bool degenerate = degenerate_pattern(norm2, lo, hi, metric == KPDI_METRIC_NCC);  // -> an all-zero row (prep_device.h)
  const float inv = degenerate ? 0.f : 1.f / sqrtf(norm2);
  const float cval = degenerate ? 0.f : sqrtf((float)k) * mean * inv;
  if (degenerate) {
#pragma unroll
    for (int i = 0; i < WAVE_VALUES; ++i) v[i] = 0.f;
  }
  // two passes over the row: planes [0, half_planes) and the rest
  const int planes = kpad / 8, half_planes = (planes + 1) / 2;
  const int row_floats = 8 * half_planes + 4;  // + 16 bytes: the four rows start in different banks
  float *stage = (float *)smem_raw + (size_t)g * row_floats;
  for (int pass = 0; pass < 2; ++pass) {
    const int c_first = pass * 8 * half_planes;
    const int c_last = pass == 0 ? 8 * half_planes : kpad;
    if (pass) __syncthreads();  // the first half has been written out
#pragma unroll
    for (int i = 0; i < WAVE_VALUES / 4; ++i) {
      const int c = 4 * (t + 256 * i);
      if (c >= c_first && c < c_last) {
        float4 w;
        w.x = (centred && c == k) ? cval : v[4 * i] * inv;
        w.y = (centred && c + 1 == k) ? cval : v[4 * i + 1] * inv;
        w.z = (centred && c + 2 == k) ? cval : v[4 * i + 2] * inv;
        w.w = (centred && c + 3 == k) ? cval : v[4 * i + 3] * inv;
        *reinterpret_cast<float4 *>(stage + (c - c_first)) = w;
      }
    }
    __syncthreads();
    // write-out as in write_lines4, 1024 threads: 8 threads per 128-byte line
    const int lr = (form >> 8) & 0xff, bk = (form >> 16) & 0xff;
    const unsigned magic = bk == 48 ? 89478486u : 134217728u;
    const int nsteps = (int)__umulhi(2u * (unsigned)kpad, magic);
    const int r0 = blockIdx.x * 4, row0 = r0 & ((1 << lr) - 1), swz = (row0 >> 3) & 1;
    const int j = tid & 7, row = j >> 1, half = j & 1;
    const int p_first = c_first / 8, p_last = c_last / 8;
    for (int P = p_first + (tid >> 3); P < p_last; P += PREP16_THREADS / 8) {
      const int step = (int)__umulhi((unsigned)(16 * P), magic);
      const int pl = P - step * (bk / 16);
      const size_t block = (size_t)(r0 >> lr) * nsteps + step;
      char *line = (char *)out + ((block * bk) << (lr + 1)) + (((size_t)pl << lr) + row0) * 32;
      const float4 q = *reinterpret_cast<const float4 *>((const float *)smem_raw + (size_t)row * row_floats +
                                                         8 * (P - p_first) + 4 * (half ^ swz));
      *reinterpret_cast<float4 *>(line + 32 * row + 16 * half) = q;
    }
  }
}

// ---- plane-major forms (float16, wide float32), K % 4 == 0, no mask: four patterns per workgroup as whole lines ----
// prep_wave_kernel<T, 4, true> stores 8 / 16 bytes per lane into 32-byte row segments (3-4.6 TB/s); here the four
// waves stage their normalised rows in LDS (4 x kpad floats) and the workgroup writes 128-byte lines (write_lines4).
template <typename T>
__global__ __launch_bounds__(PREP_THREADS) void prep_wave_lines_kernel(const T *raw, int npix, const int *row_map, int k,
                                                                       int kpad, int metric, int n_out, float *out, int form) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r0 = blockIdx.x * 4, r = r0 + wv;
  float *stage = (float *)smem_raw + (size_t)wv * kpad;
  float v[WAVE_VALUES];
  float s = 0.f;
  if (r < n_out) {
    const int64_t src = row_map ? row_map[r] : r;
    const T *p = raw + src * (int64_t)npix;
#pragma unroll
    for (int i = 0; i < WAVE_VALUES / 4; ++i) {
      const int c = 4 * (lane + 64 * i);
      Quad<T> q;
      q.v[0] = q.v[1] = q.v[2] = q.v[3] = (T)0;
      if (c < k) q = load_quad(p + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[4 * i + e] = (float)q.v[e];
        s += v[4 * i + e];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < WAVE_VALUES; ++i) v[i] = 0.f;  // rows beyond the chunk: zeros (their line is shared with valid rows)
  }
  normalise_and_store_quads<64, WAVE_VALUES, true>(v, s, lane, r, k, kpad, metric, out, form, nullptr, stage);
  __syncthreads();
  write_lines4(out, (const float *)smem_raw, kpad, r0, kpad, form, threadIdx.x);
}

// ---- one wave per pattern, signal mask, row staged in LDS -------------------------------
// LDS: [k ints pixel map][4 waves x npix floats].  Workgroups are persistent over groups of
// 4 patterns, so the pixel map is staged once per workgroup.
template <typename T, bool H16>
__global__ __launch_bounds__(PREP_THREADS) void prep_wave_masked_kernel(const T *raw, int npix, const int *row_map,
                                                                        const int *pix_map, int k, int kpad,
                                                                        int metric, int n_out, float *out, int split) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int *map = (int *)smem_raw;
  const int map_words = (k + 3) & ~3;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float *row = (float *)smem_raw + map_words + wv * npix;
  for (int c = threadIdx.x; c < map_words; c += PREP_THREADS) map[c] = c < k ? pix_map[c] : 0;
  __syncthreads();
  const int ngroups = (n_out + 3) / 4;
  for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int r = g * 4 + wv;
    const bool live = r < n_out;
    if (live) {
      const int64_t src = row_map ? row_map[r] : r;
      const T *p = raw + src * (int64_t)npix;
#pragma unroll
      for (int i = 0; i < WAVE_VALUES / 4; ++i) {
        const int c = 4 * (lane + 64 * i);
        if (c < npix) {
          const Quad<T> q = load_quad(p + c);
          float4 w;
          w.x = (float)q.v[0];
          w.y = (float)q.v[1];
          w.z = (float)q.v[2];
          w.w = (float)q.v[3];
          *reinterpret_cast<float4 *>(row + c) = w;
        }
      }
    }
    __syncthreads();  // rows staged (also orders this wave's own LDS writes before its gathers)
    if (live) {
      float v[WAVE_VALUES];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < WAVE_VALUES / 4; ++i) {
        const int c = 4 * (lane + 64 * i);
        int4 px = make_int4(0, 0, 0, 0);
        if (c < k) px = *reinterpret_cast<const int4 *>(map + c);
        v[4 * i] = c < k ? row[px.x] : 0.f;
        v[4 * i + 1] = c + 1 < k ? row[px.y] : 0.f;
        v[4 * i + 2] = c + 2 < k ? row[px.z] : 0.f;
        v[4 * i + 3] = c + 3 < k ? row[px.w] : 0.f;
        s += (v[4 * i] + v[4 * i + 1]) + (v[4 * i + 2] + v[4 * i + 3]);
      }
      normalise_and_store_quads<64, WAVE_VALUES, H16>(v, s, lane, r, k, kpad, metric, out, split);
    }
    __syncthreads();  // rows are overwritten by the next group
  }
}

// ---- the same for float32 patterns (dictionaries), rows brought in by LDS-DMA, double-buffered ---------------
// prep_wave_masked_kernel is latency-bound: 69 KB of LDS per workgroup = 2 waves per SIMD, each of which waits for
// its row, gathers, writes, and only then asks for the next row (SQ_WAIT_ANY / SQ_WAVE_CYCLES = 0.77; 3.2 TB/s).
// Here a wave's NEXT row is in flight (buffer_load ... lds, no registers involved) while it gathers, normalises and
// stores the current one.  LDS: [k ints pixel map][4 waves x 2 x row], a row rounded up to whole 1 KB pieces
// (reads past the row's end return zeros: the buffer descriptor is sized to the row).
template <bool H16>
__global__ __launch_bounds__(PREP_THREADS) void prep_wave_masked_dma_kernel(const float *raw, int npix, const int *row_map,
                                                                            const int *pix_map, int k, int kpad,
                                                                            int metric, int n_out, float *out, int split) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int *map = (int *)smem_raw;
  const int map_words = (k + 3) & ~3;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row_words = (npix + 255) & ~255;
  const int pieces = row_words / 256;
  float *rows = (float *)smem_raw + map_words + (size_t)wv * 2 * row_words;
  for (int c = threadIdx.x; c < map_words; c += PREP_THREADS) map[c] = c < k ? pix_map[c] : 0;
  __syncthreads();
  auto fetch = [&](int r, int buf) {
    const int64_t src = row_map ? row_map[r] : r;
    const float *p = raw + src * (int64_t)npix;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, npix * 4, 0x00020000);
#pragma unroll
    for (int q = 0; q < WAVE_VALUES / 4; ++q)
      if (q < pieces)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsrc, (__attribute__((address_space(3))) void *)(rows + (size_t)buf * row_words + q * 256), 16, lane * 16,
            q * 1024, 0, 0);
  };
  const int stride = gridDim.x * 4;
  int r = blockIdx.x * 4 + wv, buf = 0;
  if (r < n_out) fetch(r, 0);
  for (; r < n_out; r += stride, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this row has landed (and the previous row's stores are out)
    if (r + stride < n_out) fetch(r + stride, buf ^ 1);
    const float *row = rows + (size_t)buf * row_words;
    float v[WAVE_VALUES];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < WAVE_VALUES / 4; ++i) {
      const int c = 4 * (lane + 64 * i);
      int4 px = make_int4(0, 0, 0, 0);
      if (c < k) px = *reinterpret_cast<const int4 *>(map + c);
      v[4 * i] = c < k ? row[px.x] : 0.f;
      v[4 * i + 1] = c + 1 < k ? row[px.y] : 0.f;
      v[4 * i + 2] = c + 2 < k ? row[px.z] : 0.f;
      v[4 * i + 3] = c + 3 < k ? row[px.w] : 0.f;
      s += (v[4 * i] + v[4 * i + 1]) + (v[4 * i + 2] + v[4 * i + 3]);
    }
    normalise_and_store_quads<64, WAVE_VALUES, H16>(v, s, lane, r, k, kpad, metric, out, split);
    // the gathers of this row before the fetch that overwrites it (two iterations on): the s_waitcnt of the next
    // iteration orders them - LDS reads have returned into `v` by then (their values were used above)
  }
}

// ---- signal mask, float32 rows: the kept pixels gathered STRAIGHT from global memory ------------------------------
// A signal mask keeps runs of consecutive detector pixels (a circular mask: one run per detector row), so four
// consecutive KEPT pixels are almost always four consecutive floats of the raw row, and never more than two such
// runs when gather_descriptors() says so.  A lane's quad is then two unaligned 16-byte buffer loads - element e
// comes from the first while e < j, from the second (which starts j floats before the second run) after - instead
// of a staged copy of the whole row in LDS: no LDS for the input, no barrier, no wait for the acknowledgement of the
// previous row's stores, and the workgroup's LDS budget goes to writing whole 128-byte lines (write_lines4).
// Reads past the row's end return zero (the buffer descriptor is sized to the row); they are never selected.
bool gather_descriptors(const int *pix_map, int k, int npix, std::vector<unsigned> *out) {
  out->clear();
  if (npix > 4096 || k <= 0) return false;
  for (int q = 0; 4 * q < k; ++q) {
    const int n = std::min(4, k - 4 * q);
    const int *p = pix_map + 4 * q;
    int j = n;  // first element that does not continue the run of element 0
    for (int e = 1; e < n; ++e)
      if (p[e] != p[0] + e) {
        j = e;
        break;
      }
    for (int e = j + 1; e < n; ++e)
      if (p[e] != p[j] + (e - j)) return false;  // a third run
    const int off2 = j < n ? p[j] - j : p[0];      // >= 0: pix_map ascends, so p[j] > p[j - 1] >= j - 1
    if (off2 < 0) return false;
    out->push_back((unsigned)p[0] | ((unsigned)off2 << 12) | ((unsigned)(j < n ? j : 4) << 24));
  }
  return true;
}

template <bool LINES>
__global__ __launch_bounds__(PREP_THREADS) void prep_wave_gather_kernel(const float *raw, int npix, const int *row_map,
                                                                        const unsigned *quad_desc, int k, int kpad,
                                                                        int metric, int n_out, float *out, int form) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r0 = blockIdx.x * 4, r = r0 + wv;
  float *stage = LINES ? (float *)smem_raw + (size_t)wv * kpad : nullptr;
  float v[WAVE_VALUES];
  float s = 0.f;
  if (r < n_out) {
    const int64_t src = row_map ? row_map[r] : r;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(raw + src * (int64_t)npix), 0, npix * 4, 0x00020000);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    unsigned d[WAVE_VALUES / 4];
#pragma unroll
    for (int i = 0; i < WAVE_VALUES / 4; ++i) d[i] = 4 * (lane + 64 * i) < k ? quad_desc[lane + 64 * i] : 0u;
    u32x4 a[WAVE_VALUES / 4], b[WAVE_VALUES / 4];
#pragma unroll
    for (int i = 0; i < WAVE_VALUES / 4; ++i) {  // every load in flight before the first use
      if (4 * 64 * i < k) {                      // (wave-uniform)
        a[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((d[i] & 0xfffu) * 4u), 0, KPDI_PREP_NT ? 2 : 0);
        b[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((d[i] >> 12) & 0xfffu) * 4u), 0, KPDI_PREP_NT ? 2 : 0);
      }
    }
#pragma unroll
    for (int i = 0; i < WAVE_VALUES / 4; ++i) {
      const int c = 4 * (lane + 64 * i);
      const int j = (int)(d[i] >> 24);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = 0.f;
        if (4 * 64 * i < k) x = __uint_as_float(e < j ? a[i][e] : b[i][e]);
        v[4 * i + e] = c + e < k ? x : 0.f;
      }
      s += (v[4 * i] + v[4 * i + 1]) + (v[4 * i + 2] + v[4 * i + 3]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < WAVE_VALUES; ++i) v[i] = 0.f;  // rows beyond the chunk: zeros (their line is shared with valid rows)
  }
  if (LINES) {
    normalise_and_store_quads<64, WAVE_VALUES, true>(v, s, lane, r, k, kpad, metric, out, form, nullptr, stage);
    __syncthreads();
    write_lines4(out, (const float *)smem_raw, kpad, r0, kpad, form, threadIdx.x);
  } else if (r < n_out) {
    normalise_and_store_quads<64, WAVE_VALUES, false>(v, s, lane, r, k, kpad, metric, out, form);
  }
}

// ---- split-f16 form of a prepared matrix (KPDI_COMPUTE_F16X2), in place -------------------
// One thread per (pattern row, 32-pixel slab): its eight 16-byte slots hold 32 floats
// (slot q = pixels 4q..4q+3); they are rewritten as v * 2^12 = hi + lo with hi = f16(v * 2^12),
// lo = f16(v * 2^12 - hi): slots 0-3 = hi of pixels 8q..8q+7, slots 4-7 = lo of the same.
// A thread reads all of its slots before it writes any, and no two threads share a slot.
__global__ __launch_bounds__(256) void split_f16_kernel(float *prepared, int nslab, int64_t n_items) {
  const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (item >= n_items) return;
  const int row = (int)(item & 127);
  const int64_t block = item >> 7;  // tile * nslab + slab
  float4 *base = (float4 *)(prepared + block * 4096);
  const int rp = row >> 1, hb = (row & 1) << 3, sw = rp & 7;
  float v[32];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float4 f = base[rp * 16 + ((hb | q) ^ sw)];
    v[4 * q] = f.x;
    v[4 * q + 1] = f.y;
    v[4 * q + 2] = f.z;
    v[4 * q + 3] = f.w;
  }
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = v[8 * q + e] * 4096.f;
      const _Float16 h = (_Float16)x;
      hi[e] = h;
      lo[e] = (_Float16)(x - (float)h);
    }
    *(h8 *)&base[rp * 16 + ((hb | q) ^ sw)] = hi;
    *(h8 *)&base[rp * 16 + ((hb | (4 + q)) ^ sw)] = lo;
  }
}

hipError_t launch_split_f16(float *prepared, int n_rows_pad, int kpad, hipStream_t s) {
  const int nslab = kpad / TILE_K;
  const int64_t n_items = (int64_t)n_rows_pad * nslab;
  if (n_items <= 0) return hipSuccess;
  hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, s, prepared, nslab,
                     n_items);
  return hipGetLastError();
}

hipError_t launch_prep(const PrepLaunch &a, hipStream_t s) {
  if (a.n_out <= 0) return hipSuccess;
  const int cols = a.k + (a.metric == NORM_NDP_CENTRED ? 1 : 0);  // columns of a row that are not padding
  // columns a row-owning wave / workgroup has to write: everything up to the padded row length
  const int span = std::max(cols, a.operand_form == 2 ? 2 * a.kpad : a.kpad);
  const bool wave_path = span <= 64 * WAVE_VALUES;
  // what the kernels are told: the float16 form carries its block geometry
  const int form = a.operand_form == 2 ? f16_form(a.f16_rows, a.f16_step) : (a.operand_form == 3 ? wide32_form() : a.operand_form);
  const bool vec_ok = (a.npix % 4) == 0 && ((uintptr_t)a.raw % (4 * dtype_size(a.dtype))) == 0;
  const bool vec4 = wave_path && a.pix_map == nullptr && (a.k % 4) == 0 && vec_ok;
  const bool staged = wave_path && a.pix_map != nullptr && vec_ok && a.npix <= 64 * WAVE_VALUES && !getenv("KPDI_PREP_NO_STAGED");
  // larger detectors, still register-resident: one workgroup per pattern
  const bool block_path = !wave_path && span <= PREP_THREADS * WAVE_VALUES;
  const bool block_vec = block_path && a.pix_map == nullptr && (a.k % 4) == 0 && vec_ok;
  const bool block_masked = block_path && a.pix_map != nullptr;
  const size_t staged_lds = (size_t)(((a.k + 3) & ~3) + 4 * a.npix) * 4;
  const size_t lines_lds = (size_t)4 * a.kpad * 4 * (getenv("KPDI_PREP_NO_LINES") ? 1000 : 1);  // prep_wave_lines_kernel
  // float32 rows (dictionaries): LDS-DMA, double-buffered (prep_wave_masked_dma_kernel)
  const size_t dma_lds = (size_t)(((a.k + 3) & ~3) + 8 * ((a.npix + 255) & ~255)) * 4;
  const bool staged_dma = staged && a.dtype == KPDI_F32 && dma_lds <= 160 * 1024 && !getenv("KPDI_PREP_NO_DMA");
  // float32 rows whose mask is a set of runs: gathered straight from global memory (prep_wave_gather_kernel); the
  // plane-major forms (2: float16, 3: wide float32) are written as whole lines through LDS, forms 0 / 1 as 16-byte slots
  const bool gather = wave_path && a.pix_map != nullptr && a.quad_desc != nullptr && a.dtype == KPDI_F32 &&
                      ((uintptr_t)a.raw % 4) == 0 && (a.operand_form < 2 || lines_lds <= 64 * 1024) &&
                      !getenv("KPDI_PREP_NO_GATHER");
  static const bool prep16_affine = getenv("KPDI_PREP16") && !strcmp(getenv("KPDI_PREP16"), "block");
  static const int prep16_np = getenv("KPDI_PREP16") && !strcmp(getenv("KPDI_PREP16"), "block4") ? 4 : 2;
  dim3 block(PREP_THREADS);
  dim3 grid(wave_path ? (a.n_out + 3) / 4 : a.n_out);
  if (staged && !gather) grid = dim3(std::min((a.n_out + 3) / 4, 2048));
#define KPDI_PREP_H(T, H)                                                                                \
  if (H && vec4 && lines_lds <= 64 * 1024)                                                               \
    hipLaunchKernelGGL((prep_wave_lines_kernel<T>), grid, block, lines_lds, s, (const T *)a.raw, a.npix, a.row_map, a.k, \
                       a.kpad, a.metric, a.n_out, a.out, form);                                          \
  else if (vec4)                                                                                       \
    hipLaunchKernelGGL((prep_wave_kernel<T, 4, H>), grid, block, 0, s, (const T *)a.raw, a.npix,        \
                       a.row_map, a.pix_map, a.k, a.kpad, a.metric, a.n_out, a.out, form);               \
  else if (staged_dma) {                                                                                 \
    auto kd = prep_wave_masked_dma_kernel<H>;                                                            \
    if (dma_lds > 64 * 1024) {                                                                           \
      hipError_t e = hipFuncSetAttribute((const void *)kd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dma_lds); \
      if (e != hipSuccess) return e;                                                                     \
    }                                                                                                    \
    hipLaunchKernelGGL(kd, dim3(std::min((a.n_out + 3) / 4, 1024)), block, dma_lds, s, (const float *)a.raw, a.npix, \
                       a.row_map, a.pix_map, a.k, a.kpad, a.metric, a.n_out, a.out, form);               \
  } else if (staged) {                                                                                   \
    if (staged_lds > 64 * 1024) {                                                                        \
      hipError_t e = hipFuncSetAttribute((const void *)prep_wave_masked_kernel<T, H>,                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)staged_lds);   \
      if (e != hipSuccess) return e;                                                                     \
    }                                                                                                    \
    hipLaunchKernelGGL((prep_wave_masked_kernel<T, H>), grid, block, staged_lds, s, (const T *)a.raw,   \
                       a.npix, a.row_map, a.pix_map, a.k, a.kpad, a.metric, a.n_out, a.out, form);       \
  } else if (wave_path)                                                                                  \
    hipLaunchKernelGGL((prep_wave_kernel<T, 1, H>), grid, block, 0, s, (const T *)a.raw, a.npix,        \
                       a.row_map, a.pix_map, a.k, a.kpad, a.metric, a.n_out, a.out, H ? form : 0);       \
  else if (block_vec)                                                                                    \
    hipLaunchKernelGGL((prep_block_kernel<T, false, H>), H ? dim3(round_up(a.n_out, 32)) : grid, block, 0, s,            \
                       (const T *)a.raw, a.npix, a.row_map, a.pix_map, a.k, a.kpad, a.metric, a.out, form, H ? a.n_out : 0); \
  else if (block_masked)                                                                                 \
    hipLaunchKernelGGL((prep_block_kernel<T, true, H>), H ? dim3(round_up(a.n_out, 32)) : grid, block, 0, s,             \
                       (const T *)a.raw, a.npix, a.row_map, a.pix_map, a.k, a.kpad, a.metric, a.out, form, H ? a.n_out : 0); \
  else                                                                                                   \
    hipLaunchKernelGGL((prep_kernel<T>), grid, block, 0, s, (const T *)a.raw, a.npix, a.row_map,        \
                       a.pix_map, a.k, a.kpad, a.metric, a.out, H ? form : 0);
#define KPDI_PREP(T)                  \
  if (gather) {                                                                                           \
    if (a.operand_form >= 2)                                                                              \
      hipLaunchKernelGGL((prep_wave_gather_kernel<true>), grid, block, lines_lds, s, (const float *)a.raw, a.npix,       \
                         a.row_map, a.quad_desc, a.k, a.kpad, a.metric, a.n_out, a.out, form);            \
    else                                                                                                  \
      hipLaunchKernelGGL((prep_wave_gather_kernel<false>), grid, block, 0, s, (const float *)a.raw, a.npix,              \
                         a.row_map, a.quad_desc, a.k, a.kpad, a.metric, a.n_out, a.out, form);            \
  } else if (a.operand_form == 3 && (block_vec || block_masked)) {                                              \
    const size_t lds32 = (size_t)4 * (8 * ((a.kpad / 8 + 1) / 2) + 4) * 4;                               \
    auto k32 = block_masked ? prep32_block4_kernel<T, true> : prep32_block4_kernel<T, false>;            \
    if (lds32 > 64 * 1024) {                                                                             \
      hipError_t e = hipFuncSetAttribute((const void *)k32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32); \
      if (e != hipSuccess) return e;                                                                     \
    }                                                                                                    \
    hipLaunchKernelGGL(k32, dim3((a.n_out + 3) / 4), dim3(PREP16_THREADS), lds32, s, (const T *)a.raw, a.npix, \
                       a.row_map, a.pix_map, a.k, a.kpad, a.metric, a.n_out, a.out, form);               \
  } else if (a.operand_form == 2 && (block_vec || block_masked) && !prep16_affine) {                     \
    const int np = prep16_np;                                                                            \
    const size_t lds16 = (size_t)np * (2 * a.kpad + 8) * 2;                                              \
    auto k16 = np == 2 ? (block_masked ? prep16_block4_kernel<T, true, 2> : prep16_block4_kernel<T, false, 2>)           \
                       : (block_masked ? prep16_block4_kernel<T, true, 4> : prep16_block4_kernel<T, false, 4>);          \
    if (lds16 > 64 * 1024) {                                                                             \
      hipError_t e = hipFuncSetAttribute((const void *)k16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16); \
      if (e != hipSuccess) return e;                                                                     \
    }                                                                                                    \
    /* NP = 2: the affine mapping walks row groups of 4 in blocks of 16 units */                         \
    hipLaunchKernelGGL(k16, dim3(np == 2 ? round_up((a.n_out + 1) / 2, 16) : (a.n_out + 3) / 4), dim3(256 * np), lds16, s, \
                       (const T *)a.raw, a.npix, a.row_map, a.pix_map, a.k, a.kpad, a.metric, a.n_out, a.out, form);     \
  } else if (a.operand_form >= 2) {   \
    KPDI_PREP_H(T, true)              \
  } else {                            \
    KPDI_PREP_H(T, false)             \
  }                                   \
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
    case KPDI_F16: KPDI_PREP(_Float16)
    default: return hipErrorInvalidValue;
  }
#undef KPDI_PREP
#undef KPDI_PREP_H
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // paths that store whole float4 slots write the split-f16 form themselves; the others are
  // converted in place afterwards (rows beyond n_out are zero in either form).  The float16
  // form is written directly by every path.
  if (a.operand_form == 1 && !(vec4 || staged || gather || block_vec || block_masked))
    return launch_split_f16(a.out, round_up(a.n_out, TILE_DICT), a.kpad, s);
  return hipSuccess;
}

}  // namespace kpdi
