// projection.h - device code shared by project.hip (dictionary generation) and refine.hip
// (refinement objective): one detector pixel's intensity from a square-Lambert master
// pattern.  Reference: signals/util/_master_pattern.py:449-708, _utils/numba.py:59-81.
//
// f64 with the reference's formulas; its divisions, square roots and arctan are evaluated
// with division-free helpers (hardware reciprocal seeds + one correction step, a
// polynomial arctan) that agree with libm to the last bit or two - far below the rounding
// of the float32 value the reference keeps.  Translation units including this header use
// `#pragma clang fp contract(fast)` for this code (it follows no NumPy f32 operation order).
#pragma once
#include <hip/hip_runtime.h>

namespace kpdi {

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

// d = a * b + c with the uniform constant c read from a scalar register pair.  Written as
// asm because the compiler otherwise evaluates Horner's scheme with v_fmac_f64 (D = A*B + D),
// which needs every coefficient moved into a fresh VGPR pair first: two extra VALU moves per
// term, 22 per pixel.  Scalar moves issue beside the vector pipe.
__device__ __forceinline__ double fma_sconst(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
  return d;
}

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
  // atan(u)/u in s = u^2 (Chebyshev fit, see above), Horner with scalar-register coefficients
  const double s = u * u;
  double p = 0x1.2edf629854fb0p-6;
  p = fma_sconst(p, s, -0x1.50b33e5fd9dc6p-5);
  p = fma_sconst(p, s, 0x1.ca7e184710557p-5);
  p = fma_sconst(p, s, -0x1.0f2d87b7f5b5cp-4);
  p = fma_sconst(p, s, 0x1.3adfd52a966cfp-4);
  p = fma_sconst(p, s, -0x1.745968dbb8c55p-4);
  p = fma_sconst(p, s, 0x1.c71c469a22141p-4);
  p = fma_sconst(p, s, -0x1.24924889af9fep-3);
  p = fma_sconst(p, s, 0x1.999999973269cp-3);
  p = fma_sconst(p, s, -0x1.555555555329bp-2);
  p = fma_sconst(p, s, 0x1.ffffffffffff8p-1);
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

}  // namespace kpdi
