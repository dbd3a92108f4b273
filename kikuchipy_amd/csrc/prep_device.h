// prep_device.h - device helpers shared by prep.hip (pattern preparation) and preproc.hip (the fused
// background-removal + preparation kernel): reductions, the float16 slot address and the
// normalise-and-store tail that writes a pattern's prepared row into the tiled layout of kernels.h.
#pragma once
#include "kernels.h"
#include "../../include/kpdi.h"

namespace kpdi {

constexpr int NORM_NDP_CENTRED = 2;  // internal value of the `metric` argument: `ndp` in its centred form (prep.hip)

// ---- degenerate patterns (include/kpdi.h, "Degenerate patterns") ---------------------------------------------------
// A pattern whose normalisation is undefined is DEGENERATE: `ncc` - a CONSTANT pattern (dead or saturated detector
// frame): all kept pixels equal, tested EXACTLY (minimum == maximum of the pixels as read - no tolerance: one pixel
// of 3600 off by one count at 60 000 counts is an ordinary pattern and correlates as in the reference); `ndp` - an
// all-zero pattern; either metric - NaN or inf among the kept pixels, or a sum of squares that is not a positive
// finite number.  The reference divides 0 by 0 there (similarity_metrics/_normalized_cross_correlation.py:228-233,
// _normalized_dot_product.py:181-194) and ranks the resulting NaN FIRST (dask/array/chunk.py:167-258).  Here such a
// pattern is prepared as the all-zero row: its score against every pattern is exactly +0 ("no correlation"), on the
// experimental and on the dictionary side, in every arithmetic.  `norm2` = sum of squares the row is divided by the
// root of; `lo` / `hi` = minimum / maximum of the kept pixels before the mean is removed; `ncc` = the metric removes
// the mean and divides by the centred norm.
template <typename F>
__host__ __device__ inline bool degenerate_pattern(F norm2, F lo, F hi, bool ncc) {
  return !(norm2 > (F)0 && norm2 < (F)__builtin_inff()) || (ncc && lo == hi);  // (NaN fails both comparisons)
}
constexpr int PREP_THREADS = 256;
constexpr int WAVE_VALUES = 64;  // values per lane of the wave-per-pattern kernels (K <= 4096)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// minimum and maximum over a workgroup of NT threads; `red`: NT / 64 floats (used twice)
template <int NT>
__device__ __forceinline__ void block_minmax_n(float &lo, float &hi, float *red) {
  lo = wave_min(lo);
  hi = wave_max(hi);
  const int w = threadIdx.x >> 6;
  __syncthreads();  // protect `red` from the previous use
  if ((threadIdx.x & 63) == 0) red[w] = lo;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) t = fminf(t, red[i]);
  lo = t;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = hi;
  __syncthreads();
  t = red[0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) t = fmaxf(t, red[i]);
  hi = t;
}

// sum over a workgroup of NT threads; `red`: NT / 64 floats
template <int NT>
__device__ __forceinline__ float block_sum_n(float v, float *red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();  // protect `red` from the previous use
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_sum(float v, float *red) { return block_sum_n<PREP_THREADS>(v, red); }

template <typename T>
struct alignas(sizeof(T) * 4) Quad {
  T v[4];
};

// Four consecutive values of a raw pattern.  Raw patterns are read exactly once per preparation: the load is marked
// non-temporal (KPDI_PREP_NT, default on) so that the stream does not push the prepared matrices the match kernel is
// about to read out of the L2 / Infinity Cache.
#ifndef KPDI_PREP_NT
#define KPDI_PREP_NT 1
#endif
template <typename T>
__device__ __forceinline__ Quad<T> load_quad(const T *p) {
#if KPDI_PREP_NT
  Quad<T> q;
  if constexpr (sizeof(T) == 4) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    __builtin_memcpy(&q, &x, sizeof q);
    return q;
  } else if constexpr (sizeof(T) == 2) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 x = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(p));
    __builtin_memcpy(&q, &x, sizeof q);
    return q;
  } else if constexpr (sizeof(T) == 1) {
    const unsigned x = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(p));
    __builtin_memcpy(&q, &x, sizeof q);
    return q;
  } else {
    return *reinterpret_cast<const Quad<T> *>(p);
  }
#else
  return *reinterpret_cast<const Quad<T> *>(p);
#endif
}
// 8 / 16 bytes of a prepared row on their way out (KPDI_PREP_NT_STORE: non-temporal as well)
template <typename V>
__device__ __forceinline__ void store_out(V *dst, const V &v) {
#if defined(KPDI_PREP_NT_STORE) && KPDI_PREP_NT_STORE
  __builtin_nontemporal_store(v, dst);
#else
  *dst = v;
#endif
}

// ---- float16 form (KPDI_COMPUTE_F16): a row holds 2 * kpad f16, value * 2^12, in the layout of
// match16.hip: patterns in tiles of R rows (256, or 128 for the dictionary of the 4-wave variant),
// pixels in steps of B (48 or 32); one (tile, step) block is contiguous and stored PLANE-major:
// [B / 16 planes][R rows][16 pixels] - plane p holds the 16 pixels of k-step p of the step for all rows,
// 32 bytes per row, its two 16-byte halves (pixels 0-7 / 8-15 = what the lanes 0-31 / 32-63 of an MFMA
// operand read) swapped for rows with bit 3 set.  That swap makes every 16-lane group of a ds_read_b128
// (rows r .. of one plane) hit 16 distinct bank quads; 32 contiguous bytes per pattern and plane - a whole
// 128-byte line for 4 consecutive patterns - is what the preparation kernels can write efficiently (with
// 16-byte planes the 8 patterns sharing a line were written ~10 us apart: 2.4 TB/s at 120 x 120).
// `form` carries the geometry: bits 0-7 = 2, bits 8-15 = log2(R), bits 16-23 = B (f16_form below);
// `kpad` = floats per row (= B / 2 per step).
__host__ __device__ inline int f16_form(int tile_rows, int step) {
  return 2 | ((tile_rows == 128 ? 7 : 8) << 8) | (step << 16);
}
// Form 3 = the same plane-major blocks holding FLOAT32 values (the f32 form of match16.hip: tiles of 256 patterns,
// steps of 24 pixels, a row's 32 bytes of a plane = 8 pixels): byte for byte the float16 layout with every float32
// in the place of two float16, so pixel c lives where float16 index 2 c would (half_slot(out, r, 2 c, kpad, form)).
__host__ __device__ inline int wide32_form() { return 3 | (8 << 8) | (F16_STEP << 16); }
__device__ __forceinline__ char *half_slot(float *out, int r, int c, int kpad, int form) {
  const int lr = (form >> 8) & 0xff, bk = (form >> 16) & 0xff;
  // c / bk as ONE v_mul_hi_u32 (exact for c < 10^7): a runtime divisor costs ~25 instructions per stored slot,
  // and that code sits - unrolled 16 times - in every preparation kernel, whatever the form (it slowed the
  // masked f32 preparation from 0.79 to 1.33 ms)
  const unsigned magic = bk == 48 ? 89478486u : 134217728u;  // ceil(2^32 / 48), 2^32 / 32
  const int nsteps = (int)__umulhi(2u * (unsigned)kpad, magic);
  const int step = (int)__umulhi((unsigned)c, magic), cs = c - step * bk;
  const size_t block = (size_t)(r >> lr) * nsteps + step;
  const int row = r & ((1 << lr) - 1);
  return (char *)out + ((block * bk) << (lr + 1)) + (((cs >> 4) << lr) + row) * 32 + ((((cs >> 3) ^ (row >> 3)) & 1) << 4) +
         (cs & 7) * 2;
}

// same, v[4*i + e] holds kept pixel 4*(lane + 64*i) + e: float4 stores (full 16-byte slots)
// `split` = the operand form: 1 stores the split-f16 form directly (see split_f16_kernel below): the
// lane's four pixels are half of an 8-pixel slot, i.e. 8 bytes of the high-half slot and 8 of the
// low-half slot; 2 stores the float16 form (half_slot above)
// NT threads share the pattern: 64 = one wave (`lane` = lane id), PREP_THREADS = the whole
// workgroup (`lane` = thread id, sums through `red` in LDS)
template <int NT>
__device__ __forceinline__ float group_sum(float v, float *red) {
  if (NT == 64) return wave_sum(v);
  return block_sum_n<NT>(v, red);
}

template <int NT>
__device__ __forceinline__ void group_minmax(float &lo, float &hi, float *red) {
  if (NT == 64) {
    lo = wave_min(lo);
    hi = wave_max(hi);
  } else {
    block_minmax_n<NT>(lo, hi, red);
  }
}

// H16: the kernel was instantiated FOR the float16 form (split & 0xff == 2) / for the other two forms: with one
// body for all three, the float16 stores' address arithmetic - unrolled 16 times - raised the f32 kernels
// from 170-204 to 256 registers and the masked f32 preparation from 0.79 to 1.33 ms.
// `stage` (plane-major forms only): the row goes to this LDS buffer (kpad floats: logical order, no swizzle) instead of
// global memory; write_lines4() then writes four consecutive rows as whole 128-byte lines.
template <int NT = 64, int NV = WAVE_VALUES, bool H16 = false>
__device__ __forceinline__ void normalise_and_store_quads(float (&v)[NV], float s, int lane, int r, int k,
                                                          int kpad, int metric, float *out, int split,
                                                          float *red = nullptr, float *stage = nullptr) {
  const int nslab = kpad / TILE_K;
  float mean = 0.f;
  if (metric != KPDI_METRIC_NDP) mean = group_sum<NT>(s, red) / (float)k;
  float q2 = 0.f, lo = INFINITY, hi = -INFINITY;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (lane + NT * (i / 4)) + (i & 3);
    if (c < k) {
      lo = fminf(lo, v[i]);
      hi = fmaxf(hi, v[i]);
      v[i] -= mean;
      q2 += v[i] * v[i];
    } else {
      v[i] = 0.f;
    }
  }
  q2 = group_sum<NT>(q2, red);
  if (metric == KPDI_METRIC_NCC) group_minmax<NT>(lo, hi, red);
  const bool centred = metric == NORM_NDP_CENTRED;
  const float norm2 = centred ? q2 + (float)k * mean * mean : q2;
  const bool degenerate = degenerate_pattern(norm2, lo, hi, metric == KPDI_METRIC_NCC);  // (uniform over the NT threads of the row)
  const float inv = degenerate ? 0.f : 1.f / sqrtf(norm2);
  const float cval = degenerate ? 0.f : sqrtf((float)k) * mean * inv;
  if (degenerate) {  // 0 x NaN is NaN: the row is written as zeros
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < NV / 4; ++i) {
    const int c = 4 * (lane + NT * i);
    if (c < ((H16 && (split & 0xff) == 2) ? 2 * kpad : kpad)) {
      float4 w;
      w.x = (centred && c == k) ? cval : v[4 * i] * inv;
      w.y = (centred && c + 1 == k) ? cval : v[4 * i + 1] * inv;
      w.z = (centred && c + 2 == k) ? cval : v[4 * i + 2] * inv;
      w.w = (centred && c + 3 == k) ? cval : v[4 * i + 3] * inv;
      if (!H16 && !split) {
        *reinterpret_cast<float4 *>(out + prepared_offset(r, c, nslab)) = w;
      } else if (H16 && (split & 0xff) == 3) {
        if (stage)
          *reinterpret_cast<float4 *>(stage + c) = w;
        else
          *reinterpret_cast<float4 *>(half_slot(out, r, 2 * c, kpad, split)) = w;  // half of a row's 32 bytes of a plane
      } else if (H16) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 h;
        h[0] = (_Float16)(w.x * 4096.f);
        h[1] = (_Float16)(w.y * 4096.f);
        h[2] = (_Float16)(w.z * 4096.f);
        h[3] = (_Float16)(w.w * 4096.f);
        if (stage)
          *reinterpret_cast<h4 *>((_Float16 *)stage + c) = h;
        else
          *reinterpret_cast<h4 *>(half_slot(out, r, c, kpad, split)) = h;  // half of a slot: 8 bytes
      } else {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const float x[4] = {w.x * 4096.f, w.y * 4096.f, w.z * 4096.f, w.w * 4096.f};
        h4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          hi[e] = (_Float16)x[e];
          lo[e] = (_Float16)(x[e] - (float)hi[e]);
        }
        const int g8 = (c & 31) >> 3, half = (c & 7) >> 2;  // 8-pixel group of the slab; first / second 4 pixels
        const int slab_first = c & ~31;                      // any pixel of slot q lies at slab_first + 4 q
        char *hi_slot = (char *)(out + prepared_offset(r, slab_first + 4 * g8, nslab));
        char *lo_slot = (char *)(out + prepared_offset(r, slab_first + 4 * (4 + g8), nslab));
        *reinterpret_cast<h4 *>(hi_slot + 8 * half) = hi;
        *reinterpret_cast<h4 *>(lo_slot + 8 * half) = lo;
      }
    }
  }
}


// Four consecutive rows r0 .. r0 + 3 (r0 % 4 == 0) of a plane-major form, staged in LDS as rows of `row_floats` floats
// (kpad * 4 bytes of payload each, logical order), written by 256 threads as whole 128-byte lines: a plane's 32 bytes of
// the four rows are contiguous, the two 16-byte halves of a row swapped when its bit 3 is set - the same for all four.
__device__ __forceinline__ void write_lines4(float *out, const float *stage, int row_floats, int r0, int kpad, int form,
                                             int tid) {
  const int lr = (form >> 8) & 0xff, bk = (form >> 16) & 0xff;
  const unsigned magic = bk == 48 ? 89478486u : 134217728u;
  const int planes = kpad / 8;                       // 32 bytes per row and plane
  const int planes_per_step = bk / 16;
  const int nsteps = (int)__umulhi(2u * (unsigned)kpad, magic);
  const int row0 = r0 & ((1 << lr) - 1);
  const int swz = (row0 >> 3) & 1;
  const int j = tid & 7, row = j >> 1, half = j & 1;
  for (int P = tid >> 3; P < planes; P += 32) {
    const int step = (int)__umulhi((unsigned)(16 * P), magic);
    const int pl = P - step * planes_per_step;
    const size_t block = (size_t)(r0 >> lr) * nsteps + step;
    char *line = (char *)out + ((block * bk) << (lr + 1)) + (((size_t)pl << lr) + row0) * 32;
    const float4 q = *reinterpret_cast<const float4 *>(stage + (size_t)row * row_floats + 8 * P + 4 * (half ^ swz));
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const f32x4v qv = {q.x, q.y, q.z, q.w};
    store_out(reinterpret_cast<f32x4v *>(line + 32 * row + 16 * half), qv);
  }
}

}  // namespace kpdi
