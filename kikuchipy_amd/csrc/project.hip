// project.hip - dictionary generation on the device: projection of a square-Lambert
// master pattern onto the detector, one simulated pattern per rotation.
// What EBSDMasterPattern.get_patterns computes per dask chunk:
//   signals/util/_master_pattern.py:299-370  _project_patterns_from_master_pattern_with_fixed_pc
//   signals/util/_master_pattern.py:449-527  _project_single_pattern_from_master_pattern
//   signals/util/_master_pattern.py:530-568  _vector2lambert
//   signals/util/_master_pattern.py:580-678  _get_lambert_interpolation_parameters
//   signals/util/_master_pattern.py:682-708  _get_pixel_from_master_pattern
//   _utils/numba.py:59-81                    rotate_vector
//   pattern/_pattern.py:97-111               _rescale_with_min_max
//
// All arithmetic in f64 with the reference's formulas; its divisions, square roots and
// arctan are evaluated with division-free helpers (hardware reciprocal seeds + one
// correction step, a polynomial arctan) that agree with libm to the last bit or two -
// far below the rounding of the float32 output.  The master pattern is held as f32 (exact
// for the uint8 / uint16 / float32 master patterns kikuchipy loads) and widened on use.
//
// One workgroup per simulated pattern.  The detector's direction cosines (npix x 3 f64)
// and the master pattern (2 x npx x npy f32, L2/MALL resident) are shared by every
// pattern; per pixel the kernel reads 24 B of direction cosines, gathers 4 x 4 B and
// writes one output value.  Detectors up to 64 x 64 keep the f64 intensities of the
// pattern in registers between the min/max reduction and the rescale; larger ones
// recompute them.
#include "kernels.h"
#include "../../include/kpdi.h"

// this file does not follow an f32 NumPy operation order (see above): let mul+add fuse
#pragma clang fp contract(fast)
#include "projection.h"

namespace kpdi {

#ifndef PROJ_MIN_BLOCKS
#define PROJ_MIN_BLOCKS 2
#endif
#ifndef PROJ_FENCE
#define PROJ_FENCE 4
#endif
constexpr int PROJ_THREADS = 256;
constexpr int PROJ_VALUES = 16;  // register-resident pixels per thread (<= 4096 per pattern)

template <typename T>
__device__ __forceinline__ T cast_out(double v);
template <> __device__ __forceinline__ float cast_out<float>(double v) { return (float)v; }
template <> __device__ __forceinline__ double cast_out<double>(double v) { return v; }
// ndarray.astype(integer): truncation toward zero
template <> __device__ __forceinline__ uint8_t cast_out<uint8_t>(double v) { return (uint8_t)(int)v; }
template <> __device__ __forceinline__ uint16_t cast_out<uint16_t>(double v) { return (uint16_t)(int)v; }

__device__ __forceinline__ void block_minmax(double &lo, double &hi, double *red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    lo = fmin(lo, __shfl_xor(lo, o, 64));
    hi = fmax(hi, __shfl_xor(hi, o, 64));
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[2 * w] = lo;
    red[2 * w + 1] = hi;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PROJ_THREADS / 64; ++i) {
    lo = fmin(lo, red[2 * i]);
    hi = fmax(hi, red[2 * i + 1]);
  }
}

template <typename T, bool IN_REGS>
__global__ __launch_bounds__(PROJ_THREADS, PROJ_MIN_BLOCKS) void project_kernel(const double *rotations, const double *dc, int npix,
                                                               MasterView mp, int rescale, double omin,
                                                               double omax, T *out) {
  __shared__ double red[2 * PROJ_THREADS / 64];
  const int64_t n = blockIdx.x;
  const RotCoeff r = rot_coeff(rotations + 4 * n);
  T *o = out + n * (int64_t)npix;
  const int tid = threadIdx.x;
  if (IN_REGS) {
    // straight-line code: pixel indices are clamped instead of branching around the tail
    // (a divergent branch per slot makes the compiler copy the whole value array around)
    double v[PROJ_VALUES];
    double lo = INFINITY, hi = -INFINITY;
    const int last = npix - 1;
#pragma unroll
    for (int i = 0; i < PROJ_VALUES; ++i) {
      const int c = tid + PROJ_THREADS * i;
      const int cc = min(c, last);
      v[i] = project_pixel(r, dc[3 * cc], dc[3 * cc + 1], dc[3 * cc + 2], mp);
      lo = fmin(lo, v[i]);  // a clamped slot repeats the last pixel: min/max unchanged
      hi = fmax(hi, v[i]);
      // two pixels in flight per thread; without the fence the scheduler interleaves all 16
      // and the kernel needs > 400 VGPRs (one wave per SIMD)
      if ((i % PROJ_FENCE) == PROJ_FENCE - 1) __builtin_amdgcn_sched_barrier(0);
    }
    double gain = 1.0, offs = 0.0, base = 0.0;
    if (rescale) {
      block_minmax(lo, hi, red);
      gain = (omax - omin) / (hi - lo);
      base = lo;
      offs = omin;
    }
#pragma unroll
    for (int i = 0; i < PROJ_VALUES; ++i) {
      const int c = tid + PROJ_THREADS * i;
      const double w = rescale ? (v[i] - base) * gain + offs : v[i];
      if (c < npix) o[c] = cast_out<T>(w);
    }
  } else {
    double lo = INFINITY, hi = -INFINITY;
    if (rescale) {
      for (int c = tid; c < npix; c += PROJ_THREADS) {
        const double v = project_pixel(r, dc[3 * c], dc[3 * c + 1], dc[3 * c + 2], mp);
        lo = fmin(lo, v);
        hi = fmax(hi, v);
      }
      block_minmax(lo, hi, red);
    }
    const double gain = (omax - omin) / (hi - lo);
    for (int c = tid; c < npix; c += PROJ_THREADS) {
      double v = project_pixel(r, dc[3 * c], dc[3 * c + 1], dc[3 * c + 2], mp);
      if (rescale) v = (v - lo) * gain + omin;
      o[c] = cast_out<T>(v);
    }
  }
}

size_t packed_master_floats(int npx, int npy) { return (size_t)2 * npy * (npx + 1) * 2; }

void pack_master_pattern(const float *upper, const float *lower, int npx, int npy, float *out) {
  for (int h = 0; h < 2; ++h) {
    const float *m = h ? lower : upper;
    for (int r = 0; r < npy; ++r) {
      const int r1 = r + 1 < npy ? r + 1 : r;
      float *o = out + ((size_t)(h * npy + r) * (npx + 1)) * 2;
      for (int c = 0; c <= npx; ++c) {
        const int cc = c < npx ? c : npx - 1;
        o[2 * c] = m[(size_t)r * npx + cc];
        o[2 * c + 1] = m[(size_t)r1 * npx + cc];
      }
    }
  }
}

hipError_t launch_project(const ProjectLaunch &a, hipStream_t s) {
  if (a.n <= 0) return hipSuccess;
  MasterView mp;
  mp.packed = (const float2 *)a.master_packed;
  mp.npx = a.npx;
  mp.npy = a.npy;
  mp.scale = (double)(a.npx - 1) / 2.0;
  mp.lam2px = mp.scale / 1.2533141373155002512;  // sqrt(pi / 2)
  const bool regs = a.npix <= PROJ_THREADS * PROJ_VALUES;
  dim3 grid((unsigned)a.n), block(PROJ_THREADS);
#define KPDI_PROJECT(T)                                                                                   \
  if (regs)                                                                                               \
    hipLaunchKernelGGL((project_kernel<T, true>), grid, block, 0, s, a.rotations, a.direction_cosines,   \
                       a.npix, mp, a.rescale, a.out_min, a.out_max, (T *)a.out);                          \
  else                                                                                                    \
    hipLaunchKernelGGL((project_kernel<T, false>), grid, block, 0, s, a.rotations, a.direction_cosines,  \
                       a.npix, mp, a.rescale, a.out_min, a.out_max, (T *)a.out);                          \
  break;
  switch (a.dtype_out) {
    case KPDI_F32: KPDI_PROJECT(float)
    case KPDI_F64: KPDI_PROJECT(double)
    case KPDI_U8: KPDI_PROJECT(uint8_t)
    case KPDI_U16: KPDI_PROJECT(uint16_t)
    default: return hipErrorInvalidValue;
  }
#undef KPDI_PROJECT
  return hipGetLastError();
}

}  // namespace kpdi
