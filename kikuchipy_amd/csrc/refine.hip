// refine.hip - refinement of orientations and/or projection centres (PCs): the objective
// functions AND the optimiser on the device.
//   indexing/_refinement/_objective_functions.py:36-190   1 - NCC(exp, project(euler, pc))
//   indexing/_refinement/_solvers.py:51-74               _prepare_pattern
//   indexing/_refinement/_solvers.py:79-460              *_solver_scipy with method=minimize
//   similarity_metrics/_normalized_cross_correlation.py:200-225   NCC of a centred pattern
//   _utils/numba.py:43-57 (rotation_from_euler), _utils/_gnonomic_bounds.py:25-65,
//   signals/util/_master_pattern.py:133-204 (direction cosines for one PC)
// The optimiser the reference calls is scipy.optimize.minimize(method="Nelder-Mead"), a
// third-party dependency (pyproject: scipy >= 1.7); `NelderMead` below restates SciPy
// 1.15.3's `_minimize_neldermead` (scipy/optimize/_optimize.py) operation for operation in
// f64 without fused multiply-add, so that on the same objective values it walks the same
// simplex path (kpdi_nelder_mead_selftest pins that bit for bit against SciPy).
//
// One workgroup per (experimental pattern, start) runs the WHOLE optimisation: thread 0
// owns the simplex (LDS) and decides the next point; all 256 threads evaluate the
// objective there (each pixel: direction cosine for the current PC -> rotate -> Lambert ->
// one 16-byte master-pattern gather -> float32 value; three f64 block sums give the NCC).
// No host round trip and no kernel launch per evaluation: a refinement of M patterns is
// one launch.  The objective is f64-VALU-bound like project.hip; the centred pattern
// (k x 4 B) and the master pattern stay in L2.
#include "kernels.h"
#include "../../include/kpdi.h"

#pragma clang fp contract(fast)
#include "projection.h"
#pragma clang fp contract(off)

#include <climits>

namespace kpdi {

constexpr int REF_THREADS = 256;
constexpr int NV_MAX = 6;

__device__ __forceinline__ double block_sum3(double &a, double &b, double &c, double *red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    a += __shfl_xor(a, o, 64);
    b += __shfl_xor(b, o, 64);
    c += __shfl_xor(c, o, 64);
  }
  const int w = threadIdx.x >> 6;
  __syncthreads();  // `red` may still be read from the previous evaluation
  if ((threadIdx.x & 63) == 0) {
    red[3 * w] = a;
    red[3 * w + 1] = b;
    red[3 * w + 2] = c;
  }
  __syncthreads();
  a = b = c = 0.0;
#pragma unroll
  for (int i = 0; i < REF_THREADS / 64; ++i) {
    a += red[3 * i];
    b += red[3 * i + 1];
    c += red[3 * i + 2];
  }
  return a;
}

// ---- _prepare_pattern: gather kept pixels, float32, optional rescale to [-1, 1] (float32
// arithmetic in the reference's order), centre, squared norm of the centred pattern
template <typename T>
__global__ __launch_bounds__(REF_THREADS) void refine_prep_kernel(const T *raw, int npix, const int *pix_map, int k,
                                                                  int rescale, float *out, double *sqnorm) {
  __shared__ double red[3 * REF_THREADS / 64];
  __shared__ float mm[2 * REF_THREADS / 64];
  const int64_t n = blockIdx.x;
  const T *p = raw + n * npix;
  float *o = out + n * k;
  const int tid = threadIdx.x;
  float lo = INFINITY, hi = -INFINITY;
  if (rescale) {
    for (int c = tid; c < k; c += REF_THREADS) {
      const float v = (float)p[pix_map ? pix_map[c] : c];
      lo = fminf(lo, v);
      hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
      lo = fminf(lo, __shfl_xor(lo, s, 64));
      hi = fmaxf(hi, __shfl_xor(hi, s, 64));
    }
    if ((tid & 63) == 0) {
      mm[2 * (tid >> 6)] = lo;
      mm[2 * (tid >> 6) + 1] = hi;
    }
    __syncthreads();
    for (int i = 0; i < REF_THREADS / 64; ++i) {
      lo = fminf(lo, mm[2 * i]);
      hi = fmaxf(hi, mm[2 * i + 1]);
    }
  }
  const float span = hi - lo;
  double s1 = 0.0, d0 = 0.0, d1 = 0.0;
  for (int c = tid; c < k; c += REF_THREADS) {
    float v = (float)p[pix_map ? pix_map[c] : c];
    if (rescale) v = (v - lo) / span * 2.0f + -1.0f;
    s1 += (double)v;
  }
  block_sum3(s1, d0, d1, red);
  const float mean = (float)(s1 / (double)k);
  double s2 = 0.0;
  d0 = d1 = 0.0;
  for (int c = tid; c < k; c += REF_THREADS) {
    float v = (float)p[pix_map ? pix_map[c] : c];
    if (rescale) v = (v - lo) / span * 2.0f + -1.0f;
    v -= mean;
    o[c] = v;
    s2 += (double)v * (double)v;
  }
  block_sum3(s2, d0, d1, red);
  if (tid == 0) sqnorm[n] = (double)(float)s2;  // the reference keeps it as a float32
}

// ---- geometry shared by every evaluation of a launch
struct RefineGeom {
  int nrows, ncols, k;
  const unsigned *rowcol;  // [k] (row << 16 | col) of every kept pixel
  double om[9];            // detector -> sample orientation matrix, row-major
  MasterView mp;
};

// _utils/numba.py:43-57
__device__ inline void rotation_from_euler(double phi1, double Phi, double phi2, double *q) {
  const double sigma = 0.5 * (phi1 + phi2), delta = 0.5 * (phi1 - phi2);
  double c, s, cs, ss, cd, sd;
  sincos(0.5 * Phi, &s, &c);
  sincos(sigma, &ss, &cs);
  sincos(delta, &sd, &cd);
  q[0] = c * cs;
  q[1] = -s * cd;
  q[2] = -s * sd;
  q[3] = -c * ss;
  if (q[0] < 0.0) {
    q[0] = -q[0];
    q[1] = -q[1];
    q[2] = -q[2];
    q[3] = -q[3];
  }
}

#pragma clang fp contract(fast)
// 1 - NCC between the centred experimental pattern `pat` and the pattern projected for
// quaternion `q` and PC `pc`; every thread returns the same value
__device__ inline double refine_objective(const double *q, const double *pc, const RefineGeom &g, const float *pat,
                                          double sqn, double *red) {
  const RotCoeff r = rot_coeff(q);
  // get_gnomonic_bounds + _get_direction_cosines_for_fixed_pc
  const double aspect = (double)g.ncols / (double)g.nrows;
  const double pcx = pc[0], pcy = pc[1], pcz = pc[2];
  const double x_min = -aspect * (pcx / pcz), x_max = aspect * (1.0 - pcx) / pcz;
  const double y_min = -(1.0 - pcy) / pcz, y_max = pcy / pcz;
  const double x_scale = (x_max - x_min) / (double)g.ncols, y_scale = (y_max - y_min) / (double)g.nrows;
  const double x_half = x_scale / 2.0, y_half = y_scale / 2.0;
  double s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int c = threadIdx.x; c < g.k; c += REF_THREADS) {
    const unsigned rc = g.rowcol[c];
    const double gx = x_min + (double)(rc & 0xffffu) * x_scale;
    const double gy = y_max + (double)(rc >> 16) * (-y_scale);
    const double v0 = (gx + x_half) * pcz, v1 = (gy - y_half) * pcz, v2 = pcz;
    const double w0 = v0 * g.om[0] + v1 * g.om[1] + v2 * g.om[2];
    const double w1 = v0 * g.om[3] + v1 * g.om[4] + v2 * g.om[5];
    const double w2 = v0 * g.om[6] + v1 * g.om[7] + v2 * g.om[8];
    const double rn = rsq_fast(w0 * w0 + w1 * w1 + w2 * w2);
    const double sim = (double)(float)project_pixel(r, w0 * rn, w1 * rn, w2 * rn, g.mp);  // dtype_out=float32
    s1 += sim;
    s2 += sim * sim;
    s3 += (double)pat[c] * sim;
  }
  block_sum3(s1, s2, s3, red);
  // sum(exp * (sim - mean)) = sum(exp * sim) because exp is centred; sum((sim - mean)^2) = s2 - s1^2 / k
  const double var = s2 - s1 * s1 / (double)g.k;
  return 1.0 - s3 / sqrt(sqn * var);
}
#pragma clang fp contract(off)

// control variables -> (quaternion, PC) for the three refinement modes
__device__ inline void unpack_variables(int mode, const double *x, const double *fixed, double *q, double *pc) {
  if (mode == KPDI_REFINE_ORI) {  // x = Euler angles, fixed = PC
    rotation_from_euler(x[0], x[1], x[2], q);
    pc[0] = fixed[0]; pc[1] = fixed[1]; pc[2] = fixed[2];
  } else if (mode == KPDI_REFINE_PC) {  // x = PC, fixed = quaternion
    q[0] = fixed[0]; q[1] = fixed[1]; q[2] = fixed[2]; q[3] = fixed[3];
    pc[0] = x[0]; pc[1] = x[1]; pc[2] = x[2];
  } else {  // x = Euler angles + PC
    rotation_from_euler(x[0], x[1], x[2], q);
    pc[0] = x[3]; pc[1] = x[4]; pc[2] = x[5];
  }
}

// ---- SciPy's Nelder-Mead as a state machine: `begin` / `feed(f)` are called by ONE thread;
// after each call either `done` is set or `xeval` holds the next point to evaluate.
struct NelderMead {
  enum { S_INIT, S_REFLECT, S_EXPAND, S_CONTRACT_OUT, S_CONTRACT_IN, S_SHRINK, S_DONE };
  int n, state, j, fcalls, iterations, maxiter, maxfun, bounded;
  double xatol, fatol;
  double sim[NV_MAX + 1][NV_MAX], fsim[NV_MAX + 1];
  double lb[NV_MAX], ub[NV_MAX];
  double xbar[NV_MAX], xr[NV_MAX], xt[NV_MAX], fxr;
  double xeval[NV_MAX];

  __device__ void clip(double *x) const {
    if (!bounded) return;
    for (int i = 0; i < n; ++i) x[i] = fmin(fmax(x[i], lb[i]), ub[i]);  // np.clip
  }
  // the evaluation-count guard of _wrap_scalar_function_maxfun_validation: false = "raised"
  __device__ bool request(const double *x) {
    if (fcalls >= maxfun) return false;
    ++fcalls;
    for (int i = 0; i < n; ++i) xeval[i] = x[i];
    return true;
  }
  // np.argsort on <= 7 values = insertion sort: stable, NaN last
  __device__ void sort() {
    for (int a = 1; a <= n; ++a) {
      const double f = fsim[a];
      double x[NV_MAX];
      for (int i = 0; i < n; ++i) x[i] = sim[a][i];
      int b = a - 1;
      while (b >= 0 && (f < fsim[b] || (fsim[b] != fsim[b] && f == f))) {
        fsim[b + 1] = fsim[b];
        for (int i = 0; i < n; ++i) sim[b + 1][i] = sim[b][i];
        --b;
      }
      fsim[b + 1] = f;
      for (int i = 0; i < n; ++i) sim[b + 1][i] = x[i];
    }
  }
  __device__ void finish() { state = S_DONE; }
  __device__ void raised() {  // `except _MaxFuncCallError: pass` + `finally:` sort
    sort();
    loop_top();
  }
  __device__ void end_iteration() {
    ++iterations;
    sort();
    loop_top();
  }
  __device__ void accept(const double *x, double f) {
    for (int i = 0; i < n; ++i) sim[n][i] = x[i];
    fsim[n] = f;
  }
  __device__ void loop_top() {
    if (!(fcalls < maxfun && iterations < maxiter)) return finish();
    double dx = 0.0, df = 0.0;
    for (int a = 1; a <= n; ++a) {
      for (int i = 0; i < n; ++i) dx = fmax(dx, fabs(sim[a][i] - sim[0][i]));
      df = fmax(df, fabs(fsim[0] - fsim[a]));
    }
    if (dx <= xatol && df <= fatol) return finish();
    for (int i = 0; i < n; ++i) {
      double s = sim[0][i];
      for (int a = 1; a < n; ++a) s += sim[a][i];  // np.add.reduce(sim[:-1], 0)
      xbar[i] = s / (double)n;
      xr[i] = 2.0 * xbar[i] - 1.0 * sim[n][i];  // (1 + rho) * xbar - rho * sim[-1]
    }
    clip(xr);
    state = S_REFLECT;
    if (!request(xr)) raised();
  }
  __device__ void shrink_step() {
    for (int i = 0; i < n; ++i) sim[j][i] = sim[0][i] + 0.5 * (sim[j][i] - sim[0][i]);
    clip(sim[j]);
    state = S_SHRINK;
    if (!request(sim[j])) raised();
  }
  __device__ void begin(int nvar, const double *x0, const double *lower, const double *upper, double xa, double fa,
                        int max_iter, int max_fun) {
    n = nvar;
    xatol = xa;
    fatol = fa;
    maxiter = max_iter;
    maxfun = max_fun;
    bounded = lower != nullptr;
    double x[NV_MAX];
    for (int i = 0; i < n; ++i) {
      x[i] = x0[i];
      if (bounded) {
        lb[i] = lower[i];
        ub[i] = upper[i];
      }
    }
    clip(x);
    for (int i = 0; i < n; ++i) sim[0][i] = x[i];
    for (int k = 0; k < n; ++k) {
      for (int i = 0; i < n; ++i) sim[k + 1][i] = x[i];
      sim[k + 1][k] = x[k] != 0.0 ? (1.0 + 0.05) * x[k] : 0.00025;
    }
    if (bounded)  // reflect vertices above the upper bound into the interior, then clip
      for (int a = 0; a <= n; ++a) {
        for (int i = 0; i < n; ++i)
          if (sim[a][i] > ub[i]) sim[a][i] = 2.0 * ub[i] - sim[a][i];
        clip(sim[a]);
      }
    for (int a = 0; a <= n; ++a) fsim[a] = INFINITY;
    fcalls = 0;
    iterations = 1;
    state = S_INIT;
    j = 0;
    if (!request(sim[0])) {
      sort();
      loop_top();
    }
  }
  __device__ void feed(double f) {
    switch (state) {
      case S_INIT:
        fsim[j] = f;
        ++j;
        if (j <= n && request(sim[j])) return;
        sort();
        loop_top();
        return;
      case S_REFLECT:
        fxr = f;
        if (fxr < fsim[0]) {
          for (int i = 0; i < n; ++i) xt[i] = 3.0 * xbar[i] - 2.0 * sim[n][i];  // (1 + rho chi) xbar - rho chi sim[-1]
          clip(xt);
          state = S_EXPAND;
          if (!request(xt)) raised();
        } else if (fxr < fsim[n - 1]) {
          accept(xr, fxr);
          end_iteration();
        } else if (fxr < fsim[n]) {
          for (int i = 0; i < n; ++i) xt[i] = 1.5 * xbar[i] - 0.5 * sim[n][i];  // (1 + psi rho) xbar - psi rho sim[-1]
          clip(xt);
          state = S_CONTRACT_OUT;
          if (!request(xt)) raised();
        } else {
          for (int i = 0; i < n; ++i) xt[i] = 0.5 * xbar[i] + 0.5 * sim[n][i];  // (1 - psi) xbar + psi sim[-1]
          clip(xt);
          state = S_CONTRACT_IN;
          if (!request(xt)) raised();
        }
        return;
      case S_EXPAND:
        if (f < fxr) accept(xt, f); else accept(xr, fxr);
        end_iteration();
        return;
      case S_CONTRACT_OUT:
        if (f <= fxr) {
          accept(xt, f);
          end_iteration();
        } else {
          j = 1;
          shrink_step();
        }
        return;
      case S_CONTRACT_IN:
        if (f < fsim[n]) {
          accept(xt, f);
          end_iteration();
        } else {
          j = 1;
          shrink_step();
        }
        return;
      case S_SHRINK:
        fsim[j] = f;
        ++j;
        if (j <= n) shrink_step(); else end_iteration();
        return;
      default:
        return;
    }
  }
  // OptimizeResult: fun = np.min(fsim), x = sim[0]
  __device__ double fun() const {
    double m = fsim[0];
    for (int a = 1; a <= n; ++a) m = fsim[a] < m ? fsim[a] : m;
    return m;
  }
};

// result row: fun, nfev, nit, x[0..nvar)
constexpr int RESULT_STRIDE = REFINE_RESULT_STRIDE;
static_assert(RESULT_STRIDE == 3 + NV_MAX, "result row = fun, nfev, nit + control variables");

__global__ __launch_bounds__(REF_THREADS) void refine_solve_kernel(int mode, int nvar, int nfixed, int n_starts,
                                                                    const double *x0, const double *fixed,
                                                                    const double *lower, const double *upper,
                                                                    RefineGeom g, const float *patterns,
                                                                    const double *sqnorm, double xatol, double fatol,
                                                                    int maxiter, int maxfun, double *results) {
  __shared__ NelderMead nm;
  __shared__ double red[3 * REF_THREADS / 64];
  const int64_t job = blockIdx.x;  // (pattern, start)
  const int64_t pat_id = job / n_starts;
  const float *pat = patterns + pat_id * g.k;
  const double sqn = sqnorm[pat_id];
  const double *fx = fixed + job * nfixed;
  if (threadIdx.x == 0)
    nm.begin(nvar, x0 + job * nvar, lower ? lower + job * nvar : nullptr, upper ? upper + job * nvar : nullptr, xatol,
             fatol, maxiter, maxfun);
  __syncthreads();
  while (nm.state != NelderMead::S_DONE) {
    double q[4], pc[3];
    unpack_variables(mode, nm.xeval, fx, q, pc);
    const double f = refine_objective(q, pc, g, pat, sqn, red);
    __syncthreads();  // every thread has read nm.xeval / nm.state
    if (threadIdx.x == 0) nm.feed(f);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double *o = results + job * RESULT_STRIDE;
    o[0] = nm.fun();
    o[1] = (double)nm.fcalls;
    o[2] = (double)nm.iterations;
    for (int i = 0; i < nvar; ++i) o[3 + i] = nm.sim[0][i];
  }
}

// batched objective: one workgroup per evaluation
__global__ __launch_bounds__(REF_THREADS) void refine_objective_kernel(int mode, int nvar, int nfixed, const int *pattern_index,
                                                                        const double *x, const double *fixed, RefineGeom g,
                                                                        const float *patterns, const double *sqnorm,
                                                                        double *out) {
  __shared__ double red[3 * REF_THREADS / 64];
  const int64_t e = blockIdx.x;
  const int64_t pat_id = pattern_index[e];
  double q[4], pc[3];
  unpack_variables(mode, x + e * nvar, fixed + e * nfixed, q, pc);
  const double f = refine_objective(q, pc, g, patterns + pat_id * g.k, sqnorm[pat_id], red);
  if (threadIdx.x == 0) out[e] = f;
}

// ---- the optimiser alone on analytic f64 objectives (pins NelderMead against SciPy)
__device__ inline double selftest_objective(int kind, int n, const double *x) {
  double acc = 0.0;
  if (kind == 0) {  // Rosenbrock: sum(100 (x[i+1] - x[i]^2)^2 + (1 - x[i])^2)
    for (int i = 0; i + 1 < n; ++i) {
      const double d = x[i + 1] - x[i] * x[i], e = 1.0 - x[i];
      const double t = 100.0 * (d * d) + e * e;
      acc = i == 0 ? t : acc + t;
    }
  } else {  // weighted bowl: sum((i + 1) (x[i] - 0.3 (i + 1))^2)
    for (int i = 0; i < n; ++i) {
      const double d = x[i] - 0.3 * (double)(i + 1);
      const double t = (double)(i + 1) * (d * d);
      acc = i == 0 ? t : acc + t;
    }
  }
  return acc;
}

__global__ void nelder_mead_selftest_kernel(int kind, int nvar, const double *x0, const double *lower, const double *upper,
                                            double xatol, double fatol, int maxiter, int maxfun, double *result) {
  __shared__ NelderMead nm;
  if (threadIdx.x != 0) return;
  nm.begin(nvar, x0, lower, upper, xatol, fatol, maxiter, maxfun);
  while (nm.state != NelderMead::S_DONE) nm.feed(selftest_objective(kind, nvar, nm.xeval));
  result[0] = nm.fun();
  result[1] = (double)nm.fcalls;
  result[2] = (double)nm.iterations;
  for (int i = 0; i < nvar; ++i) result[3 + i] = nm.sim[0][i];
}

// ---- launchers
static RefineGeom make_geom(const RefineLaunch &a) {
  RefineGeom g;
  g.nrows = a.nrows;
  g.ncols = a.ncols;
  g.k = a.k;
  g.rowcol = a.rowcol;
  for (int i = 0; i < 9; ++i) g.om[i] = a.om[i];
  g.mp.packed = (const float2 *)a.master_packed;
  g.mp.npx = a.npx;
  g.mp.npy = a.npy;
  g.mp.scale = (double)(a.npx - 1) / 2.0;
  g.mp.lam2px = g.mp.scale / 1.2533141373155002512;
  return g;
}

hipError_t launch_refine_prep(const void *raw, int dtype, int64_t n, int npix, const int *pix_map, int k, int rescale,
                              float *out, double *sqnorm, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  dim3 grid((unsigned)n), block(REF_THREADS);
#define KPDI_RPREP(T)                                                                                            \
  hipLaunchKernelGGL((refine_prep_kernel<T>), grid, block, 0, s, (const T *)raw, npix, pix_map, k, rescale, out, \
                     sqnorm);                                                                                    \
  break;
  switch (dtype) {
    case KPDI_U8: KPDI_RPREP(uint8_t)
    case KPDI_I8: KPDI_RPREP(int8_t)
    case KPDI_U16: KPDI_RPREP(uint16_t)
    case KPDI_I16: KPDI_RPREP(int16_t)
    case KPDI_I32: KPDI_RPREP(int32_t)
    case KPDI_U32: KPDI_RPREP(uint32_t)
    case KPDI_F32: KPDI_RPREP(float)
    case KPDI_F64: KPDI_RPREP(double)
    case KPDI_F16: KPDI_RPREP(_Float16)
    default: return hipErrorInvalidValue;
  }
#undef KPDI_RPREP
  return hipGetLastError();
}

hipError_t launch_refine_solve(const RefineLaunch &a, hipStream_t s) {
  if (a.n_jobs <= 0) return hipSuccess;
  hipLaunchKernelGGL(refine_solve_kernel, dim3((unsigned)a.n_jobs), dim3(REF_THREADS), 0, s, a.mode, a.nvar, a.nfixed,
                     a.n_starts, a.x0, a.fixed, a.lower, a.upper, make_geom(a), a.patterns, a.sqnorm, a.xatol, a.fatol,
                     a.maxiter, a.maxfun, a.results);
  return hipGetLastError();
}

hipError_t launch_refine_objective(const RefineLaunch &a, const int *pattern_index, double *out, hipStream_t s) {
  if (a.n_jobs <= 0) return hipSuccess;
  hipLaunchKernelGGL(refine_objective_kernel, dim3((unsigned)a.n_jobs), dim3(REF_THREADS), 0, s, a.mode, a.nvar,
                     a.nfixed, pattern_index, a.x0, a.fixed, make_geom(a), a.patterns, a.sqnorm, out);
  return hipGetLastError();
}

hipError_t launch_nelder_mead_selftest(int kind, int nvar, const double *x0, const double *lower, const double *upper,
                                       double xatol, double fatol, int maxiter, int maxfun, double *result,
                                       hipStream_t s) {
  hipLaunchKernelGGL(nelder_mead_selftest_kernel, dim3(1), dim3(64), 0, s, kind, nvar, x0, lower, upper, xatol, fatol,
                     maxiter, maxfun, result);
  return hipGetLastError();
}

}  // namespace kpdi
