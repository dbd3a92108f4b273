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
// pattern in registers (8 per thread, 512 threads: 116 VGPRs, four waves per SIMD) between
// the min/max reduction and the rescale; larger ones recompute them.
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
#define PROJ_FENCE 2
#endif
#ifndef PROJ_THREADS_N
#define PROJ_THREADS_N 512
#endif
constexpr int PROJ_THREADS = PROJ_THREADS_N;
constexpr int PROJ_VALUES = 4096 / PROJ_THREADS;  // register-resident pixels per thread (<= 4096 per pattern)

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

// One PC per pattern (_get_direction_cosines_for_varying_pc, signals/util/_master_pattern.py:216-295):
// the direction cosine of a pixel is formed on the fly from the pattern's PC instead of being read
struct DetectorGeom {
  const double *pcs;  // [n][3] (PCx, PCy, PCz) Bruker convention, or nullptr: read `dc`
  int nrows, ncols;
  double om[9];       // detector -> sample, row-major
};

struct PixelRay {
  double x_min, y_max, x_scale, y_scale, x_half, y_half, pcz;
  int ncols;
  const double *om;
  __device__ __forceinline__ void init(const DetectorGeom &g, const double *pc) {
    const double aspect = (double)g.ncols / (double)g.nrows;
    pcz = pc[2];
    x_min = -aspect * (pc[0] / pcz);
    const double x_max = aspect * (1.0 - pc[0]) / pcz, y_min = -(1.0 - pc[1]) / pcz;
    y_max = pc[1] / pcz;
    x_scale = (x_max - x_min) / (double)g.ncols;
    y_scale = (y_max - y_min) / (double)g.nrows;
    x_half = x_scale / 2.0;
    y_half = y_scale / 2.0;
    ncols = g.ncols;
    om = g.om;
  }
  __device__ __forceinline__ void at(int c, double &x, double &y, double &z) const {
    const int row = c / ncols, col = c - row * ncols;
    const double v0 = (x_min + (double)col * x_scale + x_half) * pcz;
    const double v1 = (y_max + (double)row * (-y_scale) - y_half) * pcz;
    const double w0 = v0 * om[0] + v1 * om[1] + pcz * om[2];
    const double w1 = v0 * om[3] + v1 * om[4] + pcz * om[5];
    const double w2 = v0 * om[6] + v1 * om[7] + pcz * om[8];
    const double rn = rsq_fast(w0 * w0 + w1 * w1 + w2 * w2);
    x = w0 * rn;
    y = w1 * rn;
    z = w2 * rn;
  }
};

template <typename T, bool IN_REGS, bool VARPC>
__global__ __launch_bounds__(PROJ_THREADS, PROJ_MIN_BLOCKS) void project_kernel(const double *rotations, const double *dc, int npix,
                                                               MasterView mp, DetectorGeom geom, int rescale, double omin,
                                                               double omax, T *out) {
  __shared__ double red[2 * PROJ_THREADS / 64];
  const int64_t n = blockIdx.x;
  const RotCoeff r = rot_coeff(rotations + 4 * n);
  PixelRay ray;
  if (VARPC) ray.init(geom, geom.pcs + 3 * n);
  auto pixel = [&](int c) {
    double x, y, z;
    if (VARPC) {
      ray.at(c, x, y, z);
    } else {
      x = dc[3 * c];
      y = dc[3 * c + 1];
      z = dc[3 * c + 2];
    }
    return project_pixel(r, x, y, z, mp);
  };
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
      v[i] = pixel(cc);
      lo = fmin(lo, v[i]);  // a clamped slot repeats the last pixel: min/max unchanged
      hi = fmax(hi, v[i]);
      // two pixels in flight per thread; without the fence the scheduler interleaves all the
      // slots and the kernel needs > 400 VGPRs (one wave per SIMD)
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
        const double v = pixel(c);
        lo = fmin(lo, v);
        hi = fmax(hi, v);
      }
      block_minmax(lo, hi, red);
    }
    const double gain = (omax - omin) / (hi - lo);
    for (int c = tid; c < npix; c += PROJ_THREADS) {
      double v = pixel(c);
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
  DetectorGeom geom;
  geom.pcs = a.pcs;
  geom.nrows = a.nrows;
  geom.ncols = a.ncols;
  for (int i = 0; i < 9; ++i) geom.om[i] = a.pcs ? a.om[i] : 0.0;
  dim3 grid((unsigned)a.n), block(PROJ_THREADS);
#define KPDI_PROJECT_V(T, R, V)                                                                           \
  hipLaunchKernelGGL((project_kernel<T, R, V>), grid, block, 0, s, a.rotations, a.direction_cosines,     \
                     a.npix, mp, geom, a.rescale, a.out_min, a.out_max, (T *)a.out)
#define KPDI_PROJECT(T)                                                                                   \
  if (a.pcs) {                                                                                            \
    if (regs) KPDI_PROJECT_V(T, true, true); else KPDI_PROJECT_V(T, false, true);                         \
  } else {                                                                                                \
    if (regs) KPDI_PROJECT_V(T, true, false); else KPDI_PROJECT_V(T, false, false);                       \
  }                                                                                                       \
  break;
  switch (a.dtype_out) {
    case KPDI_F32: KPDI_PROJECT(float)
    case KPDI_F64: KPDI_PROJECT(double)
    case KPDI_U8: KPDI_PROJECT(uint8_t)
    case KPDI_U16: KPDI_PROJECT(uint16_t)
    default: return hipErrorInvalidValue;
  }
#undef KPDI_PROJECT
#undef KPDI_PROJECT_V
  return hipGetLastError();
}

}  // namespace kpdi
