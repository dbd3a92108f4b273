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

namespace kpdi {

#ifndef PROJ_MIN_BLOCKS
#define PROJ_MIN_BLOCKS 2
#endif
#ifndef PROJ_FENCE
#define PROJ_FENCE 4
#endif
constexpr int PROJ_THREADS = 256;
constexpr int PROJ_VALUES = 16;  // register-resident pixels per thread (<= 4096 per pattern)

struct RotCoeff {
  double xx, xy, xz, yy, yx, yz, zz, zy, zx;
};

// coefficients of rotate_vector (_utils/numba.py:62-81), formed exactly as written there
__device__ __forceinline__ RotCoeff rot_coeff(const double *q) {
  const double a = q[0], b = q[1], c = q[2], d = q[3];
  const double aa = a * a, bb = b * b, cc = c * c, dd = d * d;
  const double ac = a * c, ab = a * b, ad = a * d, bc = b * c, bd = b * d, cd = c * d;
  RotCoeff r;
  r.xx = aa + bb - cc - dd; r.xz = ac + bd; r.xy = bc - ad;
  r.yy = aa - bb + cc - dd; r.yx = ad + bc; r.yz = cd - ab;
  r.zz = aa - bb - cc + dd; r.zy = ab + cd; r.zx = bd - ac;
  return r;
}

// Master pattern in HBM (built by kpdi_set_master_pattern, `pack_master_pattern`): per
// hemisphere npy rows x (npx + 1) columns of float2 {m[r][c], m[r+1][c]} with the last row
// and column repeated, so the 2 x 2 bilinear footprint {m00, m10, m01, m11} of a pixel is
// ONE 16-byte gather (two neighbouring float2) and the reference's edge rule
// (`niip = nii` / `nijp = nij` beyond the last row / column) falls out of the padding.
struct MasterView {
  const float2 *packed;  // [2 hemispheres][npy][npx + 1]
  int npx, npy;
  double scale;     // (npx - 1) / 2
  double lam2px;    // scale / sqrt(pi/2): square-Lambert coordinate -> master-pattern pixels
};

// 1/d and 1/sqrt(x): hardware seed (~2^-23 relative) + one third-order correction step
// -> below 1 ulp of f64; no IEEE special-case handling (inputs are positive and normal)
__device__ __forceinline__ double rcp_fast(double d) {
  const double r = __builtin_amdgcn_rcp(d);
  const double e = fma(-d, r, 1.0);
  return fma(r, fma(e, e, e), r);
}
__device__ __forceinline__ double rsq_fast(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * y, y, 1.0);
  return fma(y, e * fma(0.375, e, 0.5), y);
}

// coefficients of atan(u)/u in u^2 (uniform loads: they live in SGPRs, which an f64 FMA
// can take as an operand; 64-bit literals would each cost a VGPR pair and a move)
__constant__ double ATAN_C[11] = {
    0x1.ffffffffffff8p-1,  -0x1.555555555329bp-2, 0x1.999999973269cp-3,  -0x1.24924889af9fep-3,
    0x1.c71c469a22141p-4,  -0x1.745968dbb8c55p-4, 0x1.3adfd52a966cfp-4,  -0x1.0f2d87b7f5b5cp-4,
    0x1.ca7e184710557p-5,  -0x1.50b33e5fd9dc6p-5, 0x1.2edf629854fb0p-6};

// atan(a / b) for 0 <= a <= b, b > 0, with ONE division: below tan(pi/8) the quotient is
// the argument, above it atan(t) = pi/4 + atan((t - 1) / (t + 1)) = pi/4 + atan((a - b) / (a + b)).
// The odd degree-21 polynomial is a Chebyshev fit of atan(u)/u on |u| <= tan(pi/8)
// (absolute error < 1e-16, i.e. rounding level).
__device__ __forceinline__ double atan_ratio(double a, double b) {
  const bool low = a <= 0.41421356237309503 * b;
  const double num = low ? a : a - b;
  const double den = low ? b : a + b;
  const double r = rcp_fast(den);
  double u = num * r;
  u = fma(fma(-den, u, num), r, u);  // one residual step: u = num / den to rounding level
  const double s = u * u;
  double p = ATAN_C[10];
#pragma unroll
  for (int i = 9; i >= 0; --i) p = fma(p, s, ATAN_C[i]);
  return fma(u, p, low ? 0.0 : 0.78539816339744831);
}

// intensity of one detector pixel: direction cosine (x, y, z) seen through rotation r.
// Same formulas as the reference; its f64 divisions / sqrt / arctan are evaluated with the
// helpers above (differences at the 1e-16 level, far below the float32 output's rounding).
__device__ __forceinline__ double project_pixel(const RotCoeff &r, double x, double y, double z,
                                                const MasterView &mp) {
  // rotate_vector
  const double vx = r.xx * x + 2.0 * (r.xz * z + r.xy * y);
  const double vy = r.yy * y + 2.0 * (r.yx * x + r.yz * z);
  const double vz = r.zz * z + 2.0 * (r.zy * y + r.zx * x);
  // _vector2lambert: normalise, then (X, Y) = sign * sqrt(2 (1 - |z|)) * (sqrt(pi)/2, 2/sqrt(pi) atan(minor/major))
  const double rn = rsq_fast(fma(vx, vx, fma(vy, vy, vz * vz)));
  const double ax = fabs(vx) * rn, ay = fabs(vy) * rn, az = fabs(vz) * rn;
  const double s2 = fmax(2.0 * (1.0 - az), 0.0);
  const double sqrt_z = s2 > 0.0 ? s2 * rsq_fast(s2) : 0.0;  // 0 at the poles: (X, Y) = (0, 0)
  constexpr double SQRT_PI = 1.7724538509055160273;
  constexpr double SQRT_PI_OVER_2 = SQRT_PI / 2.0;
  constexpr double TWO_OVER_SQRT_PI = 2.0 / SQRT_PI;
  const bool xdom = ay <= ax;
  const double major = xdom ? ax : ay, minor = xdom ? ay : ax;
  const double at = major > 0.0 ? atan_ratio(minor, major) : 0.0;
  // coordinate along the dominant axis, and across it (sign of minor/major = sign(x) sign(y))
  const double along = sqrt_z * SQRT_PI_OVER_2, across = sqrt_z * TWO_OVER_SQRT_PI * at;
  const double lx = copysign(xdom ? along : across, vx);
  const double ly = copysign(xdom ? across : along, vy);
  // _get_lambert_interpolation_parameters: row from Lambert Y, column from Lambert X
  const double i_this = ly * mp.lam2px, j_this = lx * mp.lam2px;
  const int nii = (int)(i_this + mp.scale);  // int32() truncation; the neighbours nii + 1 /
  const int nij = (int)(j_this + mp.scale);  // nij + 1 and their edge rule are in the layout
  const double di = i_this - (double)nii + mp.scale;
  const double dj = j_this - (double)nij + mp.scale;
  const double dim = 1.0 - di, djm = 1.0 - dj;
  // out-of-contract input (NaN rotation / zero vector) must not fault: clamp the read
  const int r0 = min(max(nii, 0), mp.npy - 1), c0 = min(max(nij, 0), mp.npx - 1);
  const int hemi = (vz >= 0.0) ? 0 : mp.npy;
  // _get_pixel_from_master_pattern
  const float2 *q = mp.packed + (unsigned)((hemi + r0) * (mp.npx + 1) + c0);
  float4 f;
  __builtin_memcpy(&f, q, 16);  // 8-byte aligned: {m00, m10, m01, m11}
  return ((double)f.x * dim + (double)f.y * di) * djm + ((double)f.z * dim + (double)f.w * di) * dj;
}

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
