// preproc.hip - static / dynamic background removal of the resident
// experimental patterns, in place, output in the input dtype.
//
// Reference (paths under /root/reference/src/kikuchipy):
//   _remove_static_background_subtract/_divide   pattern/_pattern.py:392-435
//   _remove_dynamic_background                   pattern/_pattern.py:438-481
//   _remove_background_subtract/_divide          pattern/_pattern.py:484-509
//   _rescale_with_min_max                        pattern/_pattern.py:96-111
//   _fft_filter (Barnes)                         filters/fft_barnes.py:155-177
// Arithmetic follows the reference's NumPy evaluation (its `.py_func`): every
// step in float32, in the same order, IEEE division, no FMA contraction (the
// library is built with -ffp-contract=off), then `.astype(dtype_out)` =
// truncation toward zero.
//
// The Barnes FFT filter with the reference's edge-replicating pad equals a
// correlation with the (separable, normalised) Gaussian window
// (tests/test_filters/test_fft_barnes.py:135-173); it is evaluated here as two
// 1-D passes in LDS with float64 accumulation.  The result differs from the
// reference's float32 FFT by its FFT round-off (~6e-5 on values ~100), which
// can flip the final truncation on isolated pixels.
//
// One workgroup per pattern; pattern + intermediate live in LDS.  HBM-bound:
// algorithmic bytes = 2 * npix * sizeof(dtype) per pattern.
#include "kernels.h"
#include "../../include/kpdi.h"
#include <math.h>

namespace kpdi {

constexpr int PP_THREADS = 256;

__device__ __forceinline__ void block_minmax(float &mn, float &mx, float *red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o, 64));
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  }
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    red[w] = mn;
    red[4 + w] = mx;
  }
  __syncthreads();
  mn = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
  mx = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
}

// (pattern - imin) / float(imax - imin) * (omax - omin) + omin, float32 (pattern/_pattern.py:110-111)
__device__ __forceinline__ float rescale(float v, float imin, float irange, float orange, float omin) {
  return ((v - imin) / irange) * orange + omin;
}

template <typename T>
__device__ __forceinline__ T cast_out(float v) {
  return (T)v;  // C truncation == ndarray.astype for in-range values
}

template <typename T>
__global__ __launch_bounds__(PP_THREADS) void static_bg_kernel(T *pats, int npix, const float *bg,
                                                               float bgmin, float bgmax, int operation,
                                                               int scale_bg, float omin, float omax) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *x = (float *)smem_raw;
  __shared__ float red[8];
  T *p = pats + (size_t)blockIdx.x * npix;
  const int tid = threadIdx.x;
  float mn = INFINITY, mx = -INFINITY;
  for (int i = tid; i < npix; i += PP_THREADS) {
    const float v = (float)p[i];
    x[i] = v;
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
  float pmin = 0.f, prange = 0.f;
  const float bgrange = bgmax - bgmin;
  if (scale_bg) {
    block_minmax(mn, mx, red);
    pmin = mn;
    prange = mx - mn;
  }
  mn = INFINITY;
  mx = -INFINITY;
  for (int i = tid; i < npix; i += PP_THREADS) {
    float b = bg[i];
    if (scale_bg) b = rescale(b, bgmin, bgrange, prange, pmin);
    const float y = operation == KPDI_OP_SUBTRACT ? x[i] - b : x[i] / b;
    x[i] = y;
    mn = fminf(mn, y);
    mx = fmaxf(mx, y);
  }
  block_minmax(mn, mx, red);
  const float irange = mx - mn;
  const float orange = omax - omin;
  for (int i = tid; i < npix; i += PP_THREADS) p[i] = cast_out<T>(rescale(x[i], mn, irange, orange, omin));
}

__device__ __forceinline__ int wrap_index(int i, int n, int reflect) {
  if (!reflect) return i < 0 ? 0 : (i >= n ? n - 1 : i);
  // scipy.ndimage 'reflect': (d c b a | a b c d | d c b a)
  const int period = 2 * n;
  i %= period;
  if (i < 0) i += period;
  return i < n ? i : period - 1 - i;
}

template <typename T>
__global__ __launch_bounds__(PP_THREADS) void dynamic_bg_kernel(T *pats, int sy, int sx, const double *ty,
                                                                int nty, int cy, const double *tx, int ntx,
                                                                int cx, int reflect, int operation,
                                                                float omin, float omax) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int npix = sy * sx;
  float *x = (float *)smem_raw;  // pattern, later pattern - background
  float *t = x + npix;           // after the axis-0 pass
  __shared__ float red[8];
  T *p = pats + (size_t)blockIdx.x * npix;
  const int tid = threadIdx.x;
  for (int i = tid; i < npix; i += PP_THREADS) x[i] = (float)p[i];
  __syncthreads();
  // axis 0 (rows)
  for (int i = tid; i < npix; i += PP_THREADS) {
    const int r = i / sx, c = i - r * sx;
    double acc = 0.0;
    for (int u = 0; u < nty; ++u) acc += ty[u] * (double)x[wrap_index(r + u - cy, sy, reflect) * sx + c];
    t[i] = (float)acc;
  }
  __syncthreads();
  // axis 1 (columns), then remove the background
  float mn = INFINITY, mx = -INFINITY;
  for (int i = tid; i < npix; i += PP_THREADS) {
    const int r = i / sx, c = i - r * sx;
    double acc = 0.0;
    for (int v = 0; v < ntx; ++v) acc += tx[v] * (double)t[r * sx + wrap_index(c + v - cx, sx, reflect)];
    const float b = (float)acc;
    const float y = operation == KPDI_OP_SUBTRACT ? x[i] - b : x[i] / b;
    x[i] = y;  // only this thread touches x[i] from here on
    mn = fminf(mn, y);
    mx = fmaxf(mx, y);
  }
  block_minmax(mn, mx, red);
  const float irange = mx - mn;
  const float orange = omax - omin;
  for (int i = tid; i < npix; i += PP_THREADS) p[i] = cast_out<T>(rescale(x[i], mn, irange, orange, omin));
}

template <typename K>
static hipError_t set_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024)
    return hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return hipSuccess;
}

hipError_t launch_static_bg(const StaticBgLaunch &a, hipStream_t s) {
  if (a.n <= 0) return hipSuccess;
  const int npix = a.sy * a.sx;
  const size_t lds = (size_t)npix * 4;
  if (lds > 150 * 1024) return hipErrorInvalidValue;
#define KPDI_ST(T)                                                                                     \
  {                                                                                                    \
    hipError_t e = set_lds(static_bg_kernel<T>, lds);                                                  \
    if (e != hipSuccess) return e;                                                                     \
    hipLaunchKernelGGL((static_bg_kernel<T>), dim3((unsigned)a.n), dim3(PP_THREADS), lds, s,          \
                       (T *)a.patterns, npix, a.bg, a.bg_min, a.bg_max, a.operation, a.scale_bg,       \
                       a.omin, a.omax);                                                                \
    break;                                                                                             \
  }
  switch (a.dtype) {
    case KPDI_U8: KPDI_ST(uint8_t)
    case KPDI_I8: KPDI_ST(int8_t)
    case KPDI_U16: KPDI_ST(uint16_t)
    case KPDI_I16: KPDI_ST(int16_t)
    case KPDI_F32: KPDI_ST(float)
    case KPDI_F64: KPDI_ST(double)
    default: return hipErrorInvalidValue;
  }
#undef KPDI_ST
  return hipGetLastError();
}

hipError_t launch_dynamic_bg(const DynamicBgLaunch &a, hipStream_t s) {
  if (a.n <= 0) return hipSuccess;
  const int npix = a.sy * a.sx;
  const size_t lds = (size_t)npix * 8;
  if (lds > 150 * 1024) return hipErrorInvalidValue;
#define KPDI_DY(T)                                                                                     \
  {                                                                                                    \
    hipError_t e = set_lds(dynamic_bg_kernel<T>, lds);                                                 \
    if (e != hipSuccess) return e;                                                                     \
    hipLaunchKernelGGL((dynamic_bg_kernel<T>), dim3((unsigned)a.n), dim3(PP_THREADS), lds, s,         \
                       (T *)a.patterns, a.sy, a.sx, a.taps_y, a.ntaps_y, a.centre_y,   \
                       a.taps_x, a.ntaps_x, a.centre_x, a.reflect, a.operation,        \
                       a.omin, a.omax);                                                                \
    break;                                                                                             \
  }
  switch (a.dtype) {
    case KPDI_U8: KPDI_DY(uint8_t)
    case KPDI_I8: KPDI_DY(int8_t)
    case KPDI_U16: KPDI_DY(uint16_t)
    case KPDI_I16: KPDI_DY(int16_t)
    case KPDI_F32: KPDI_DY(float)
    case KPDI_F64: KPDI_DY(double)
    default: return hipErrorInvalidValue;
  }
#undef KPDI_DY
  return hipGetLastError();
}

}  // namespace kpdi
