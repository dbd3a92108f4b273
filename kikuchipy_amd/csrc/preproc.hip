// preproc.hip - static / dynamic background removal of the resident experimental
// patterns, in place (output in the input dtype), FUSED with the pattern
// preparation of the metric when the patterns are about to be matched.
//
// Reference (paths under /root/reference/src/kikuchipy):
//   _remove_static_background_subtract/_divide   pattern/_pattern.py:392-435
//   _remove_dynamic_background                   pattern/_pattern.py:438-481
//   _remove_background_subtract/_divide          pattern/_pattern.py:484-509
//   _rescale_with_min_max                        pattern/_pattern.py:96-111
//   _fft_filter (Barnes)                         filters/fft_barnes.py:155-177
//   prepare_experimental (ncc / ndp)             indexing/similarity_metrics/_normalized_cross_correlation.py:88-128
// Arithmetic follows the reference's NumPy evaluation (its `.py_func`): every
// step in float32, in the same order, IEEE division, no FMA contraction (the
// library is built with -ffp-contract=off), then `.astype(dtype_out)` =
// truncation toward zero.  Both quantisations of the static -> dynamic chain are
// kept (the dynamic step sees the truncated output of the static step), so the
// fused kernel is bit-identical to running the steps one after the other.
//
// The Barnes FFT filter with the reference's edge-replicating pad equals a
// correlation with the (separable, normalised) Gaussian window
// (tests/test_filters/test_fft_barnes.py:135-173); it is evaluated as two 1-D
// passes with float64 accumulation.  The result differs from the reference's
// float32 FFT by its FFT round-off (~6e-5 on values ~100), which can flip the
// final truncation on isolated pixels.
//
// preproc_fused_kernel: ONE WORKGROUP PER PATTERN, one pass over HBM:
//   raw pattern (vector loads) -> LDS f32 -> static background (two block reductions) -> truncate
//   -> Gaussian along the rows, written TRANSPOSED so that both 1-D passes read LDS with unit
//   stride across lanes -> Gaussian along the columns -> remove -> min/max -> rescale ->
//   truncate -> pattern written back (vector stores) -> [fused prepare_experimental: gather the
//   kept pixels through the signal mask's pixel map -> mean / norm -> the tiled, swizzled row of
//   kernels.h].  The 1-D correlations are register tiled: a thread produces CONV_R consecutive
//   outputs from CONV_R + n - 1 LDS reads (n taps), the taps are wave-uniform (scalar loads from
//   a zero-padded tap array).
//   LDS: 8 bytes per pixel -> detectors up to 19 200 pixels (138 x 138).
// Larger detectors (raw 480 x 480 patterns ...) take the streaming kernels below: the same
// arithmetic with the pattern re-read from L2 and the intermediate in a global scratch.
//
// Algorithmic bytes per pattern = 2 * npix * sizeof(dtype) (+ kpad * 4 for the prepared row when fused), and the
// kernel moves just that (traffic ratio 1.0) - but it is NOT HBM-bound: the counters (profiles/r03_prekernel_pmc.txt)
// put it at 0.07-0.11 of the HBM peak with SQ_INSTS_VALU x 4 cycles / SIMD = the kernel's duration: VALU-ISSUE-bound
// (960 float64 FMAs of ~2900 vector instructions per pattern at 60 x 60), with 34-39 % of its LDS cycles in bank
// conflicts.
#include "prep_device.h"
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

namespace kpdi {

constexpr int PP_THREADS = PREP_THREADS;

// min / max over a workgroup of THREADS threads; `red`: 2 * THREADS / 64 floats
template <int THREADS>
__device__ __forceinline__ void block_minmax(float &mn, float &mx, float *red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o, 64));
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  }
  constexpr int NW = THREADS / 64;
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    red[w] = mn;
    red[NW + w] = mx;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    mn = fminf(mn, red[i]);
    mx = fmaxf(mx, red[NW + i]);
  }
}

// (pattern - imin) / float(imax - imin) * (omax - omin) + omin, float32 (pattern/_pattern.py:110-111)
__device__ __forceinline__ float rescale(float v, float imin, float irange, float orange, float omin) {
  return ((v - imin) / irange) * orange + omin;
}

// The same with the division by a WORKGROUP-UNIFORM divisor taken apart: an IEEE float32 division is, on this chip, a
// reciprocal refined once, a quotient estimate and two residual corrections (what the compiler emits around
// v_div_scale / v_div_fixup, which only matter for operands near the ends of the exponent range).  The refined reciprocal
// depends on the divisor alone, so it is computed once per pass and each pixel pays five instructions instead of eleven;
// the quotient is the correctly rounded one - bit for bit the result of `/` - whenever no scaling is needed, which
// `usable()` checks on the divisor (pattern values are at most 65 535 in magnitude).
struct UniformDivisor {
  float d, rcp;
  bool fast;
  __device__ __forceinline__ explicit UniformDivisor(float divisor) : d(divisor) {
    const float a = fabsf(divisor);
    fast = a > 0x1p-40f && a < 0x1p40f;
    float r = __builtin_amdgcn_rcpf(divisor);
    const float e = __builtin_fmaf(-divisor, r, 1.f);
    rcp = __builtin_fmaf(e, r, r);
  }
  __device__ __forceinline__ float divide(float n) const {
    if (!fast) return n / d;  // (uniform branch)
    float q = n * rcp;
    float r = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(r, rcp, q);
    r = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(r, rcp, q);
  }
};
__device__ __forceinline__ float rescale(float v, float imin, const UniformDivisor &irange, float orange, float omin) {
  return irange.divide(v - imin) * orange + omin;
}

template <typename T>
__device__ __forceinline__ T cast_out(float v) {
  return (T)v;  // C truncation == ndarray.astype for in-range values
}

// a * ld + o as ONE v_mad_i32_i24 (all three far below 2^23; the compiler cannot know and emits a quarter-rate
// v_mul_lo_u32 or a multiply and an add)
__device__ __forceinline__ int index24(int a, int ld, int o) {
  int r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(ld), "v"(o));
  return r;
}
template <bool REFLECT>
__device__ __forceinline__ int wrap_index(int i, int n) {
  if (!REFLECT) {  // clamp(i, 0, n - 1) as ONE v_med3_i32 (the compiler, not knowing that n >= 1, emits max + min)
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(i), "s"(n - 1));
    return r;
  }
  // scipy.ndimage 'reflect': (d c b a | a b c d | d c b a)
  const int period = 2 * n;
  i %= period;
  if (i < 0) i += period;
  return i < n ? i : period - 1 - i;
}

// 1-D correlation along the SLOW axis of a [len][other] array (element (a, o) at a * ld + o, ld >= other);
// consecutive threads take consecutive `o`, so the loads are unit-stride across lanes.  A job is R consecutive outputs
// (a0 .. a0 + R - 1, o): out(a) = sum_u taps[u] * in(wrap(a + u - centre)), accumulated in float64 (fused multiply-add:
// the reference's FFT fixes no operation order here) in ascending u.  `tp` is the tap array padded with CONV_R - 1 zeros
// on both sides (tp[u + CONV_R - 1] = taps[u]): its index below is wave-uniform.
//
// What bounds it (rocprofv3 counters, profiles/r03_prekernel_pmc.txt): VALU ISSUE.  Round 2's kernel executed 3900 vector
// instructions per wave and pattern at 60 x 60 of which 960 were the float64 FMAs the filter needs (13.9 of the chip's
// 78.6 TFLOP/s float64; 0.07 of the HBM peak - it is NOT memory-bound, and the FMAs were a quarter of its work): per
// input value the loop spent a clamp, a 32-bit multiply (quarter rate), a branch on the boundary mode, an LDS read, a
// conversion and eight scalar tap loads on eight FMAs, two of which multiplied a padding zero; at 120 x 120 the 115 KB of
// LDS left one 256-thread workgroup per CU, one wave per SIMD, to wait for every one of those loads, and the transposed
// intermediate was written with a 16-way bank conflict.
// NT = the number of taps when it is one of the window sizes the reference's defaults produce (30 for 60 x 60 patterns,
// 60 for 120 x 120: int(truncate * std) with std = width / 8), 0 = any (the loop at the end).  With NT known: 16 outputs
// per job (half the inputs per output), chunks of 30 taps held in scalar registers, every (input, output) pair whose tap
// is a padding zero dropped at compile time, all LDS reads of a chunk in flight before the first FMA needs one, a 24-bit
// multiply-add for the address.  The products and their order per output are those of the generic loop (adding 0 * v
// changes nothing), so the results are bit-identical to it.
constexpr int CONV_CHUNK = 30;
constexpr int CONV_RU = 16;  // outputs per job of the unrolled form
template <int NT, bool REFLECT, int THREADS, typename Load, typename Store>
__device__ __forceinline__ void correlate_slow_axis(int len, int other, int ld, const double *__restrict__ tp, int n,
                                                    int centre, Load load, Store store) {
  if (NT > 0) {
    static_assert(NT == 0 || NT % CONV_CHUNK == 0, "window sizes are handled in chunks of 30 taps");
    const int jobs = ((len + CONV_RU - 1) / CONV_RU) * other;
    for (int job = threadIdx.x; job < jobs; job += THREADS) {
      const int ablk = job / other, o = job - ablk * other, a0 = ablk * CONV_RU;
      double acc[CONV_RU];
#pragma unroll
      for (int i = 0; i < CONV_RU; ++i) acc[i] = 0.0;
#pragma unroll 1
      for (int u0 = 0; u0 < NT; u0 += CONV_CHUNK) {
        const double *__restrict__ tc = tp + (CONV_R - 1) + u0;  // taps u0 .. u0 + 29 (wave-uniform: scalar loads)
        double w[CONV_CHUNK];
#pragma unroll
        for (int u = 0; u < CONV_CHUNK; ++u) w[u] = tc[u];
        float in[CONV_CHUNK + CONV_RU - 1];
#pragma unroll
        for (int jj = 0; jj < CONV_CHUNK + CONV_RU - 1; ++jj)
          in[jj] = load(index24(wrap_index<REFLECT>(a0 + u0 + jj - centre, len), ld, o));
#pragma unroll
        for (int jj = 0; jj < CONV_CHUNK + CONV_RU - 1; ++jj) {
          const double v = (double)in[jj];
#pragma unroll
          for (int i = 0; i < CONV_RU; ++i)
            if (jj - i >= 0 && jj - i < CONV_CHUNK) acc[i] = __builtin_fma(w[jj - i], v, acc[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < CONV_RU; ++i)
        if (a0 + i < len) store(a0 + i, o, (float)acc[i]);
    }
    return;
  }
  const int jobs = ((len + CONV_R - 1) / CONV_R) * other;
  for (int job = threadIdx.x; job < jobs; job += THREADS) {
    const int o = job % other, a0 = (job / other) * CONV_R;
    double acc[CONV_R];
#pragma unroll
    for (int i = 0; i < CONV_R; ++i) acc[i] = 0.0;
    for (int j = 0; j < CONV_R + n - 1; ++j) {
      const double v = (double)load(wrap_index<REFLECT>(a0 + j - centre, len) * ld + o);
#pragma unroll
      for (int i = 0; i < CONV_R; ++i) acc[i] = __builtin_fma(tp[j - i + CONV_R - 1], v, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < CONV_R; ++i)
      if (a0 + i < len) store(a0 + i, o, (float)acc[i]);
  }
}

struct PreArgs {
  void *patterns;
  int64_t n;
  int sy, sx;
  int do_static, st_operation, scale_bg;
  const float *bg;
  float bg_min, bg_max;
  int do_dynamic, dy_operation, ntaps, centre, reflect;
  const double *tp;
  float omin, omax;
  int do_prep;
  const int *out_row, *pix_map;
  int k, kpad, metric, form;
  float *out;
  float *scratch;  // streaming kernels: 2 * npix floats per workgroup
};

// ---- detectors up to 19 200 pixels: everything in one pass ---------------------------------
// `tp` (the padded taps) is a kernel parameter of its own: only a `const __restrict__` kernel argument
// is known to be invariant, which is what lets the wave-uniform tap reads become scalar loads.
// THREADS: 256, or 1024 when the pattern's LDS footprint leaves room for one workgroup per CU anyway (120 x 120: 116 KB) -
// four waves per SIMD instead of one, so that the loads of one wave wait in the shadow of another's FMAs.
// The row pass writes its result TRANSPOSED with a row pitch of sy | 1 floats: an odd pitch spreads the 64 lanes of a store
// (consecutive columns) over all banks (pitch 60 / 120: 8- / 16-way conflicts, 39 - 48 % of the LDS cycles of round 2's kernel).
template <typename T, int NV, bool H16, int NT, int THREADS>
__global__ __launch_bounds__(THREADS) void preproc_fused_kernel(PreArgs a, const double *__restrict__ tp) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ float red[2 * THREADS / 64];
  const int sy = a.sy, sx = a.sx, npix = sy * sx;
  const int npix4 = (npix + 3) & ~3;
  float *x = (float *)smem_raw;  // the pattern: raw -> after static -> minus background -> final
  float *tt = x + npix4;         // after the row pass, transposed: tt[c * ldt + r]
  const int ldt = sy | 1;
  const int tid = threadIdx.x;
  T *p = (T *)a.patterns + (size_t)blockIdx.x * npix;
  const bool vec = (npix & 3) == 0 && (((uintptr_t)a.patterns) % (4 * sizeof(T))) == 0;
  const int nquad = npix4 >> 2;
  const float orange = a.omax - a.omin;

  // ---- load (+ min / max of the raw pattern for scale_bg)
  float mn = INFINITY, mx = -INFINITY;
  for (int q = tid; q < nquad; q += THREADS) {
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec) {
      const Quad<T> u = *reinterpret_cast<const Quad<T> *>(p + 4 * q);
      w = make_float4((float)u.v[0], (float)u.v[1], (float)u.v[2], (float)u.v[3]);
    } else {
      float *wf = &w.x;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * q + e < npix) wf[e] = (float)p[4 * q + e];
    }
    *reinterpret_cast<float4 *>(x + 4 * q) = w;
    const float *wf = &w.x;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * q + e < npix) {
        mn = fminf(mn, wf[e]);
        mx = fmaxf(mx, wf[e]);
      }
  }
  // final values of a quad -> LDS (as float) and, when this was the last step, back to memory
  auto finish = [&](float imin, float irange_value, bool store) {
    const UniformDivisor irange(irange_value);
    for (int q = tid; q < nquad; q += THREADS) {
      float4 w = *reinterpret_cast<const float4 *>(x + 4 * q);
      float *wf = &w.x;
      Quad<T> u;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        u.v[e] = cast_out<T>(rescale(wf[e], imin, irange, orange, a.omin));
        wf[e] = (float)u.v[e];
      }
      *reinterpret_cast<float4 *>(x + 4 * q) = w;
      if (store) {
        if (vec) {
          *reinterpret_cast<Quad<T> *>(p + 4 * q) = u;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (4 * q + e < npix) p[4 * q + e] = u.v[e];
        }
      }
    }
  };

  // ---- static background (pattern/_pattern.py:392-435)
  if (a.do_static) {
    float pmin = 0.f, prange = 0.f;
    const float bgrange = a.bg_max - a.bg_min;
    if (a.scale_bg) {
      block_minmax<THREADS>(mn, mx, red);
      pmin = mn;
      prange = mx - mn;
    }
    mn = INFINITY;
    mx = -INFINITY;
    for (int q = tid; q < nquad; q += THREADS) {
      float4 w = *reinterpret_cast<const float4 *>(x + 4 * q);
      float *wf = &w.x;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * q + e < npix) {
          float b = a.bg[4 * q + e];
          if (a.scale_bg) b = rescale(b, a.bg_min, bgrange, prange, pmin);
          const float y = a.st_operation == KPDI_OP_SUBTRACT ? wf[e] - b : wf[e] / b;
          wf[e] = y;
          mn = fminf(mn, y);
          mx = fmaxf(mx, y);
        }
      *reinterpret_cast<float4 *>(x + 4 * q) = w;
    }
    block_minmax<THREADS>(mn, mx, red);
    finish(mn, mx - mn, !a.do_dynamic);
  }

  // ---- dynamic background (pattern/_pattern.py:438-481)
  if (a.do_dynamic) {
    __syncthreads();
    const int dy_op = a.dy_operation;
    auto rows = [&](auto reflect) {
      correlate_slow_axis<NT, decltype(reflect)::value, THREADS>(
          sy, sx, sx, tp, a.ntaps, a.centre, [&](int i) { return x[i]; },
          [&](int r, int c, float v) { tt[c * ldt + r] = v; });
    };
    auto cols = [&](auto reflect) {
      correlate_slow_axis<NT, decltype(reflect)::value, THREADS>(
          sx, sy, ldt, tp, a.ntaps, a.centre, [&](int i) { return tt[i]; },
          [&](int c, int r, float b) {
#ifdef KPDI_PRE_TIMING_NO_CONFLICT  // timing-only ablation (WRONG results): the pass's read-modify-write of x with unit stride
            const int i = c * sx + r;  // across the lanes instead of stride sx (60: 8-way, 120: 16-way bank conflicts)
#else
            const int i = r * sx + c;
#endif
            const float y = dy_op == KPDI_OP_SUBTRACT ? x[i] - b : x[i] / b;
            x[i] = y;  // only this thread touches x[i] during the pass
            mn = fminf(mn, y);
            mx = fmaxf(mx, y);
          });
    };
    if (a.reflect) rows(std::true_type{}); else rows(std::false_type{});
    __syncthreads();
    mn = INFINITY;
    mx = -INFINITY;
    if (a.reflect) cols(std::true_type{}); else cols(std::false_type{});
    block_minmax<THREADS>(mn, mx, red);
    finish(mn, mx - mn, true);
  }

  // ---- fused preparation: what launch_prep would do with the pattern just written
  if (a.do_prep) {
    const int r = a.out_row ? a.out_row[blockIdx.x] : (int)blockIdx.x;
    if (r >= 0) {  // workgroup-uniform
      __syncthreads();
      const int k = a.k;
      float v[NV];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NV / 4; ++i) {
        const int c = 4 * (tid + THREADS * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float val = 0.f;
          if (c + e < k) val = x[a.pix_map ? a.pix_map[c + e] : c + e];
          v[4 * i + e] = val;
        }
        s += (v[4 * i] + v[4 * i + 1]) + (v[4 * i + 2] + v[4 * i + 3]);
      }
      normalise_and_store_quads<THREADS, NV, H16>(v, s, tid, r, k, a.kpad, a.metric, a.out, a.form, red);
    }
  }
}

// ---- any detector size: the pattern is re-read from memory (L2 after its first touch) ---------
template <typename T>
__global__ __launch_bounds__(PP_THREADS) void static_stream_kernel(PreArgs a) {
  __shared__ float red[8];
  const int npix = a.sy * a.sx;
  T *p = (T *)a.patterns + (size_t)blockIdx.x * npix;
  const int tid = threadIdx.x;
  const float orange = a.omax - a.omin, bgrange = a.bg_max - a.bg_min;
  float pmin = 0.f, prange = 0.f;
  float mn = INFINITY, mx = -INFINITY;
  if (a.scale_bg) {
    for (int i = tid; i < npix; i += PP_THREADS) {
      const float v = (float)p[i];
      mn = fminf(mn, v);
      mx = fmaxf(mx, v);
    }
    block_minmax<PP_THREADS>(mn, mx, red);
    pmin = mn;
    prange = mx - mn;
  }
  auto value = [&](int i) {
    float b = a.bg[i];
    if (a.scale_bg) b = rescale(b, a.bg_min, bgrange, prange, pmin);
    const float v = (float)p[i];
    return a.st_operation == KPDI_OP_SUBTRACT ? v - b : v / b;
  };
  mn = INFINITY;
  mx = -INFINITY;
  for (int i = tid; i < npix; i += PP_THREADS) {
    const float y = value(i);
    mn = fminf(mn, y);
    mx = fmaxf(mx, y);
  }
  block_minmax<PP_THREADS>(mn, mx, red);
  const float irange = mx - mn;
  for (int i = tid; i < npix; i += PP_THREADS) p[i] = cast_out<T>(rescale(value(i), mn, irange, orange, a.omin));
}

// persistent workgroups; scratch = [gridDim.x][2][npix] floats (row pass transposed | pattern - background)
template <typename T, int NT>
__global__ __launch_bounds__(PP_THREADS) void dynamic_stream_kernel(PreArgs a, const double *__restrict__ tp) {
  __shared__ float red[8];
  const int sy = a.sy, sx = a.sx, npix = sy * sx;
  float *tt = a.scratch + (size_t)blockIdx.x * 2 * npix;
  float *yb = tt + npix;
  const int tid = threadIdx.x;
  const float orange = a.omax - a.omin;
  const int dy_op = a.dy_operation;
  for (int64_t pat = blockIdx.x; pat < a.n; pat += gridDim.x) {
    T *p = (T *)a.patterns + (size_t)pat * npix;
    auto rows = [&](auto reflect) {
      correlate_slow_axis<NT, decltype(reflect)::value, PP_THREADS>(
          sy, sx, sx, tp, a.ntaps, a.centre, [&](int i) { return (float)p[i]; },
          [&](int r, int c, float v) { tt[c * sy + r] = v; });
    };
    float mn = INFINITY, mx = -INFINITY;
    auto cols = [&](auto reflect) {
      correlate_slow_axis<NT, decltype(reflect)::value, PP_THREADS>(
          sx, sy, sy, tp, a.ntaps, a.centre, [&](int i) { return tt[i]; },
          [&](int c, int r, float b) {
            const int i = r * sx + c;
            const float v = (float)p[i];
            const float y = dy_op == KPDI_OP_SUBTRACT ? v - b : v / b;
            yb[i] = y;
            mn = fminf(mn, y);
            mx = fmaxf(mx, y);
          });
    };
    if (a.reflect) rows(std::true_type{}); else rows(std::false_type{});
    __syncthreads();
    if (a.reflect) cols(std::true_type{}); else cols(std::false_type{});
    block_minmax<PP_THREADS>(mn, mx, red);  // (its barriers also order the yb writes before the reads below)
    const float irange = mx - mn;
    for (int i = tid; i < npix; i += PP_THREADS) p[i] = cast_out<T>(rescale(yb[i], mn, irange, orange, a.omin));
    __syncthreads();  // the scratch is reused by the next pattern
  }
}

size_t preprocess_scratch_floats(int sy, int sx, int64_t n, int *grid_out) {
  const int grid = (int)std::min<int64_t>(n, 512);
  if (grid_out) *grid_out = grid;
  return (size_t)grid * 2 * (size_t)sy * sx;
}

static size_t fused_lds_bytes(int sy, int sx) {
  const size_t npix4 = ((size_t)sy * sx + 3) & ~(size_t)3;
  return (npix4 + (size_t)sx * (sy | 1)) * 4;  // the pattern + the transposed intermediate (odd row pitch)
}
// threads of the fused kernel's workgroup: 1024 once a CU holds one pattern's LDS anyway
static int fused_threads(int sy, int sx) { return fused_lds_bytes(sy, sx) > 80 * 1024 ? 1024 : PP_THREADS; }

bool preprocess_fits_fused(int sy, int sx, int prepared_cols) {
  const int threads = fused_threads(sy, sx);
  return fused_lds_bytes(sy, sx) <= 150 * 1024 && prepared_cols <= threads * (threads == 1024 ? 32 : WAVE_VALUES);
}

template <typename T>
static hipError_t launch_pre_t(const PreLaunch &l, const PreArgs &a, bool *prep_done, hipStream_t s) {
  const int cols = l.k + (l.metric == NORM_NDP_CENTRED ? 1 : 0);
  const int cols_pad = l.operand_form == 2 ? 2 * l.kpad : l.kpad;
  if (preprocess_fits_fused(l.sy, l.sx, 0)) {
    // the preparation joins the kernel when a prepared row fits the 64 registers per thread
    const bool prep = l.do_prep && preprocess_fits_fused(l.sy, l.sx, std::max(cols, cols_pad));
    PreArgs b = a;
    b.do_prep = prep;
    const size_t lds = fused_lds_bytes(l.sy, l.sx);
    const int threads = fused_threads(l.sy, l.sx);
    const bool small = !prep || std::max(cols, cols_pad) <= threads * 16;
    const bool h16 = prep && l.operand_form >= 2;
    // the unrolled correlation for the window sizes of the reference's defaults (KPDI_PRE_GENERIC=1: always the loop)
    static const bool generic_only = getenv("KPDI_PRE_GENERIC") != nullptr;
    const int nt = (!l.do_dynamic || generic_only) ? 0 : (l.ntaps == 30 ? 30 : (l.ntaps == 60 ? 60 : 0));
    // (values per thread of the fused preparation: 16, else 64 - 32 with 1024 threads, whose 128 registers 64 do not fit)
#define KPDI_PRE_PICK(NT, TH)                                                                                           \
  (small ? (h16 ? preproc_fused_kernel<T, 16, true, NT, TH> : preproc_fused_kernel<T, 16, false, NT, TH>)               \
         : (h16 ? preproc_fused_kernel<T, (TH == 1024 ? 32 : WAVE_VALUES), true, NT, TH>                                \
                : preproc_fused_kernel<T, (TH == 1024 ? 32 : WAVE_VALUES), false, NT, TH>))
    auto kernel = threads == 1024 ? (nt == 60 ? KPDI_PRE_PICK(60, 1024) : KPDI_PRE_PICK(0, 1024))
                                  : (nt == 30 ? KPDI_PRE_PICK(30, PP_THREADS) : KPDI_PRE_PICK(0, PP_THREADS));
#undef KPDI_PRE_PICK
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)l.n), dim3(threads), lds, s, b, b.tp);
    *prep_done = prep;
    return hipGetLastError();
  }
  *prep_done = false;
  if (l.do_static) {
    hipLaunchKernelGGL((static_stream_kernel<T>), dim3((unsigned)l.n), dim3(PP_THREADS), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (l.do_dynamic) {
    int grid = 0;
    const size_t need = preprocess_scratch_floats(l.sy, l.sx, l.n, &grid);
    if (!l.scratch || l.scratch_floats < need) return hipErrorInvalidValue;
    if (l.ntaps == 60 && !getenv("KPDI_PRE_GENERIC"))
      hipLaunchKernelGGL((dynamic_stream_kernel<T, 60>), dim3((unsigned)grid), dim3(PP_THREADS), 0, s, a, a.tp);
    else
      hipLaunchKernelGGL((dynamic_stream_kernel<T, 0>), dim3((unsigned)grid), dim3(PP_THREADS), 0, s, a, a.tp);
    return hipGetLastError();
  }
  return hipSuccess;
}

hipError_t launch_preprocess(const PreLaunch &l, bool *prep_done, hipStream_t s) {
  *prep_done = false;
  if (l.n <= 0 || (!l.do_static && !l.do_dynamic)) return hipSuccess;
  if (l.n >= (int64_t)INT32_MAX) return hipErrorInvalidValue;
  PreArgs a;
  a.patterns = l.patterns;
  a.n = l.n;
  a.sy = l.sy;
  a.sx = l.sx;
  a.do_static = l.do_static;
  a.st_operation = l.st_operation;
  a.scale_bg = l.scale_bg;
  a.bg = l.bg;
  a.bg_min = l.bg_min;
  a.bg_max = l.bg_max;
  a.do_dynamic = l.do_dynamic;
  a.dy_operation = l.dy_operation;
  a.ntaps = l.ntaps;
  a.centre = l.centre;
  a.reflect = l.reflect;
  a.tp = l.taps_padded;
  a.omin = l.omin;
  a.omax = l.omax;
  a.do_prep = l.do_prep;
  a.out_row = l.out_row;
  a.pix_map = l.pix_map;
  a.k = l.k;
  a.kpad = l.kpad;
  a.metric = l.metric;
  a.form = l.operand_form == 2 ? f16_form(F16_TILE, l.f16_step) : (l.operand_form == 3 ? wide32_form() : l.operand_form);
  a.out = l.out;
  a.scratch = l.scratch;
  switch (l.dtype) {
    case KPDI_U8: return launch_pre_t<uint8_t>(l, a, prep_done, s);
    case KPDI_I8: return launch_pre_t<int8_t>(l, a, prep_done, s);
    case KPDI_U16: return launch_pre_t<uint16_t>(l, a, prep_done, s);
    case KPDI_I16: return launch_pre_t<int16_t>(l, a, prep_done, s);
    case KPDI_F32: return launch_pre_t<float>(l, a, prep_done, s);
    case KPDI_F64: return launch_pre_t<double>(l, a, prep_done, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace kpdi
