// extras.hip - the rows either side of the sweep: dictionary generation on the device (master
// pattern, detector, projection - project.hip), refinement (refine.hip), the orientation similarity map (osm.hip).
// (one of the host translation units api.hip was split into in round 5: context.h holds what they share)
#include "context.h"

using namespace kpdi;

namespace kpdi {

int project_to_device(kpdi_ctx *c, const double *rotations, int64_t n, int rescale, double out_min, double out_max,
                      int dtype_out, void *d_out, const VarPc *var, bool stay_async) {
  if (!c->have_master) return fail(KPDI_EINVAL, "kpdi_set_master_pattern has not been called");
  if (!var && !c->have_dc) return fail(KPDI_EINVAL, "kpdi_set_detector has not been called");
  if (!rotations) return fail(KPDI_EINVAL, "rotations pointer is NULL");
  if (n <= 0 || n >= (int64_t)INT_MAX) return fail(KPDI_EINVAL, "need between 1 and 2^31-1 rotations per call");
  if (rescale && !(out_max > out_min)) return fail(KPDI_EINVAL, "rescale needs out_max > out_min");
  HIPCHK(c->rot.reserve((size_t)n * 7 * sizeof(double)));
  const size_t rot_bytes = (size_t)n * 4 * sizeof(double);
  if (stay_async && !var) {
    kpdi_ctx::RotStage &st = c->rot_stage[c->rot_next];
    if (st.copied) HIPCHK(hipEventSynchronize(st.copied));  // (four pushes ago: long done)
    if (st.pin.reserve(rot_bytes) == hipSuccess) {
      c->rot_next = (c->rot_next + 1) % 4;
      if (!st.copied) HIPCHK(hipEventCreateWithFlags(&st.copied, hipEventDisableTiming));
      memcpy(st.pin.p, rotations, rot_bytes);
      HIPCHK(hipMemcpyAsync(c->rot.p, st.pin.p, rot_bytes, hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipEventRecord(st.copied, c->stream));
    } else {  // no page-locked memory to be had: the synchronous way
      (void)hipGetLastError();
      stay_async = false;
    }
  }
  if (!stay_async || var) HIPCHK(hipMemcpyAsync(c->rot.p, rotations, rot_bytes, hipMemcpyHostToDevice, c->stream));
  c->cnt.h2d_bytes += (double)n * 4 * sizeof(double);
  kpdi::ProjectLaunch p{};
  p.rotations = c->rot.as<double>();
  p.n = n;
  if (var) {
    double *d_pcs = c->rot.as<double>() + (size_t)n * 4;
    HIPCHK(hipMemcpyAsync(d_pcs, var->pcs, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    p.pcs = d_pcs;
    p.nrows = var->nrows;
    p.ncols = var->ncols;
    for (int i = 0; i < 9; ++i) p.om[i] = var->om[i];
    p.direction_cosines = nullptr;
    p.npix = var->nrows * var->ncols;
  } else {
    p.direction_cosines = c->dcos.as<double>();
    p.npix = (int)c->dc_npix;
  }
  p.master_packed = c->mp_packed.as<float>();
  p.npx = c->mp_npx;
  p.npy = c->mp_npy;
  p.rescale = rescale;
  p.out_min = out_min;
  p.out_max = out_max;
  p.dtype_out = dtype_out;
  p.out = d_out;
  {
    ScopedTimer t(c, &c->ev_proj);
    HIPCHK(kpdi::launch_project(p, c->stream));
  }
  // the rotations buffer may be a temporary of the caller's binding: it must have been read
  // before we return (pageable memory is staged synchronously, pinned memory is not)
  if (!stay_async || var) HIPCHK(hipStreamSynchronize(c->stream));
  return KPDI_OK;
}

}  // namespace kpdi

extern "C" {

int kpdi_set_master_pattern(kpdi_ctx *c, const void *upper, const void *lower, int dtype, int npx, int npy) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!upper) return fail(KPDI_EINVAL, "upper hemisphere pointer is NULL");
  if (npx < 2 || npy < 2) return fail(KPDI_EINVAL, "master pattern must be at least 2 x 2 pixels");
  if (dtype != KPDI_U8 && dtype != KPDI_U16 && dtype != KPDI_F32 && dtype != KPDI_F64)
    return fail(KPDI_EINVAL, "master pattern dtype must be uint8, uint16, float32 or float64");
  int rc = use_device(c);
  if (rc) return rc;
  const size_t n = (size_t)npx * npy;
  std::vector<float> up(n), lo;
  auto convert = [&](const void *src, std::vector<float> &dst) {
    switch (dtype) {
      case KPDI_U8: for (size_t i = 0; i < n; ++i) dst[i] = (float)((const uint8_t *)src)[i]; break;
      case KPDI_U16: for (size_t i = 0; i < n; ++i) dst[i] = (float)((const uint16_t *)src)[i]; break;
      case KPDI_F32: for (size_t i = 0; i < n; ++i) dst[i] = ((const float *)src)[i]; break;
      default: for (size_t i = 0; i < n; ++i) dst[i] = (float)((const double *)src)[i]; break;
    }
  };
  convert(upper, up);
  if (lower && lower != upper) {
    lo.resize(n);
    convert(lower, lo);
  }
  std::vector<float> packed(kpdi::packed_master_floats(npx, npy));
  kpdi::pack_master_pattern(up.data(), lo.empty() ? up.data() : lo.data(), npx, npy, packed.data());
  HIPCHK(c->mp_packed.reserve(packed.size() * sizeof(float)));
  HIPCHK(hipMemcpyAsync(c->mp_packed.p, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice,
                        c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->cnt.h2d_bytes += (double)(packed.size() * sizeof(float));
  c->mp_npx = npx;
  c->mp_npy = npy;
  c->have_master = true;
  return KPDI_OK;
}

int kpdi_set_direction_cosines(kpdi_ctx *c, const double *dc, int64_t npix) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!dc) return fail(KPDI_EINVAL, "direction cosines pointer is NULL");
  if (npix <= 0 || npix >= (int64_t)INT_MAX / 3) return fail(KPDI_EINVAL, "bad number of detector pixels");
  int rc = use_device(c);
  if (rc) return rc;
  const size_t bytes = (size_t)npix * 3 * sizeof(double);
  HIPCHK(c->dcos.reserve(bytes));
  HIPCHK(hipMemcpyAsync(c->dcos.p, dc, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->dc_npix = npix;
  c->have_dc = true;
  return KPDI_OK;
}

int kpdi_set_detector(kpdi_ctx *c, const double *gb, double pcz, int nrows, int ncols, const double *om) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!gb || !om) return fail(KPDI_EINVAL, "gnomonic bounds / orientation matrix pointer is NULL");
  if (nrows <= 0 || ncols <= 0) return fail(KPDI_EINVAL, "detector must have at least one pixel");
  // _get_direction_cosines_for_fixed_pc (signals/util/_master_pattern.py:175-203)
  const double x_scale = (gb[1] - gb[0]) / ncols;
  const double y_scale = (gb[3] - gb[2]) / nrows;
  const double x_half = x_scale / 2, y_half = y_scale / 2;
  std::vector<double> dc((size_t)nrows * ncols * 3);
  for (int r = 0; r < nrows; ++r) {
    const double gy = gb[3] + r * (-y_scale);  // np.arange(y_max, y_min, -y_scale)[r]
    for (int col = 0; col < ncols; ++col) {
      const double gx = gb[0] + col * x_scale;  // np.arange(x_min, x_max, x_scale)[col]
      const double v[3] = {(gx + x_half) * pcz, (gy - y_half) * pcz, pcz};
      double w[3];
      for (int a = 0; a < 3; ++a) w[a] = v[0] * om[3 * a] + v[1] * om[3 * a + 1] + v[2] * om[3 * a + 2];
      const double norm = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      double *o = &dc[((size_t)r * ncols + col) * 3];
      o[0] = w[0] / norm;
      o[1] = w[1] / norm;
      o[2] = w[2] / norm;
    }
  }
  return kpdi_set_direction_cosines(c, dc.data(), (int64_t)nrows * ncols);
}

int kpdi_get_direction_cosines(kpdi_ctx *c, double *out) {
  if (!c || !out) return fail(KPDI_EINVAL, "NULL argument");
  if (!c->have_dc) return fail(KPDI_EINVAL, "no detector set");
  int rc = use_device(c);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(out, c->dcos.p, (size_t)c->dc_npix * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return KPDI_OK;
}

int kpdi_project_patterns(kpdi_ctx *c, const double *rotations, int64_t n, int rescale, double out_min,
                          double out_max, int dtype_out, void *out) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!out) return fail(KPDI_EINVAL, "output pointer is NULL");
  if (dtype_out != KPDI_F32 && dtype_out != KPDI_F64 && dtype_out != KPDI_U8 && dtype_out != KPDI_U16)
    return fail(KPDI_EINVAL, "dtype_out must be float32, float64, uint8 or uint16");
  int rc = use_device(c);
  if (rc) return rc;
  if (!c->have_dc) return fail(KPDI_EINVAL, "kpdi_set_detector has not been called");
  const size_t bytes = (size_t)n * c->dc_npix * kpdi::dtype_size(dtype_out);
  if (n > 0) HIPCHK(c->proj_out.reserve(bytes));
  rc = project_to_device(c, rotations, n, rescale, out_min, out_max, dtype_out, c->proj_out.p);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(out, c->proj_out.p, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return KPDI_OK;
}

int kpdi_project_patterns_varying_pc(kpdi_ctx *c, const double *rotations, const double *pcs, int64_t n, int nrows,
                                     int ncols, const double *om, int rescale, double out_min, double out_max,
                                     int dtype_out, void *out) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!out || !pcs || !om) return fail(KPDI_EINVAL, "NULL argument");
  if (nrows <= 0 || ncols <= 0) return fail(KPDI_EINVAL, "detector must have at least one pixel");
  if (dtype_out != KPDI_F32 && dtype_out != KPDI_F64 && dtype_out != KPDI_U8 && dtype_out != KPDI_U16)
    return fail(KPDI_EINVAL, "dtype_out must be float32, float64, uint8 or uint16");
  int rc = use_device(c);
  if (rc) return rc;
  const size_t bytes = (size_t)n * nrows * ncols * kpdi::dtype_size(dtype_out);
  if (n > 0) HIPCHK(c->proj_out.reserve(bytes));
  const VarPc var{pcs, nrows, ncols, om};
  rc = project_to_device(c, rotations, n, rescale, out_min, out_max, dtype_out, c->proj_out.p, &var);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(out, c->proj_out.p, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return KPDI_OK;
}

}  // extern "C"

// ---- refinement ---------------------------------------------------------------
namespace {
int refine_mode_sizes(int mode, int *nvar, int *nfixed) {
  switch (mode) {
    case KPDI_REFINE_ORI: *nvar = 3; *nfixed = 3; return KPDI_OK;
    case KPDI_REFINE_PC: *nvar = 3; *nfixed = 4; return KPDI_OK;
    case KPDI_REFINE_ORI_PC: *nvar = 6; *nfixed = 0; return KPDI_OK;
  }
  return fail(KPDI_EINVAL, "unknown refinement mode %d", mode);
}

int refine_fill_launch(kpdi_ctx *c, int mode, kpdi::RefineLaunch *a) {
  if (!c->have_ref) return fail(KPDI_EINVAL, "kpdi_refine_set_patterns has not been called");
  if (!c->have_master) return fail(KPDI_EINVAL, "kpdi_set_master_pattern has not been called");
  int rc = refine_mode_sizes(mode, &a->nvar, &a->nfixed);
  if (rc) return rc;
  a->mode = mode;
  a->nrows = c->ref_nrows;
  a->ncols = c->ref_ncols;
  a->k = c->ref_k;
  a->rowcol = c->ref_rowcol.as<unsigned>();
  for (int i = 0; i < 9; ++i) a->om[i] = c->ref_om[i];
  a->master_packed = c->mp_packed.as<float>();
  a->npx = c->mp_npx;
  a->npy = c->mp_npy;
  a->patterns = c->ref_pat.as<float>();
  a->sqnorm = c->ref_sqn.as<double>();
  return KPDI_OK;
}
}  // namespace

extern "C" {

int kpdi_refine_set_patterns(kpdi_ctx *c, const void *patterns, int dtype, int64_t n, int nrows, int ncols,
                             const uint8_t *signal_mask, int rescale, const double *om) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!patterns || !om) return fail(KPDI_EINVAL, "patterns / orientation matrix pointer is NULL");
  const size_t es = kpdi::dtype_size(dtype);
  if (es == 0) return fail(KPDI_EINVAL, "unknown dtype %d", dtype);
  if (n <= 0 || n >= (int64_t)INT_MAX) return fail(KPDI_EINVAL, "need between 1 and 2^31-1 patterns");
  if (nrows <= 0 || ncols <= 0 || nrows > 65535 || ncols > 65535)
    return fail(KPDI_EINVAL, "detector shape must be within 1..65535 pixels per side");
  int rc = use_device(c);
  if (rc) return rc;
  const int npix = nrows * ncols;
  std::vector<int> map;
  std::vector<unsigned> rowcol;
  for (int i = 0; i < npix; ++i)
    if (!signal_mask || !signal_mask[i]) {
      map.push_back(i);
      rowcol.push_back(((unsigned)(i / ncols) << 16) | (unsigned)(i % ncols));
    }
  const int k = (int)map.size();
  if (k < 2) return fail(KPDI_EINVAL, "the signal mask must leave at least two pixels");
  const size_t bytes = (size_t)n * npix * es;
  HIPCHK(c->ref_raw.reserve(bytes));
  HIPCHK(c->ref_map.reserve((size_t)k * sizeof(int)));
  HIPCHK(c->ref_rowcol.reserve((size_t)k * sizeof(unsigned)));
  HIPCHK(c->ref_pat.reserve((size_t)n * k * sizeof(float)));
  HIPCHK(c->ref_sqn.reserve((size_t)n * sizeof(double)));
  HIPCHK(hipMemcpyAsync(c->ref_raw.p, patterns, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->ref_map.p, map.data(), (size_t)k * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->ref_rowcol.p, rowcol.data(), (size_t)k * sizeof(unsigned), hipMemcpyHostToDevice,
                        c->stream));
  c->cnt.h2d_bytes += (double)bytes;
  HIPCHK(kpdi::launch_refine_prep(c->ref_raw.p, dtype, n, npix, signal_mask ? c->ref_map.as<int>() : nullptr, k,
                                  rescale, c->ref_pat.as<float>(), c->ref_sqn.as<double>(), c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));  // `map`, `rowcol` and the caller's buffer have been consumed
  c->ref_nrows = nrows;
  c->ref_ncols = ncols;
  c->ref_k = k;
  c->ref_n = n;
  for (int i = 0; i < 9; ++i) c->ref_om[i] = om[i];
  c->have_ref = true;
  return KPDI_OK;
}

int kpdi_refine_get_prepared(kpdi_ctx *c, float *patterns_out, double *sqnorm_out) {
  if (!c || !patterns_out || !sqnorm_out) return fail(KPDI_EINVAL, "NULL argument");
  if (!c->have_ref) return fail(KPDI_EINVAL, "kpdi_refine_set_patterns has not been called");
  int rc = use_device(c);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(patterns_out, c->ref_pat.p, (size_t)c->ref_n * c->ref_k * sizeof(float), hipMemcpyDeviceToHost,
                        c->stream));
  HIPCHK(hipMemcpyAsync(sqnorm_out, c->ref_sqn.p, (size_t)c->ref_n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return KPDI_OK;
}

int kpdi_refine_objective(kpdi_ctx *c, int mode, int64_t n_eval, const int32_t *pattern_index, const double *x,
                          const double *fixed, double *out) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!pattern_index || !x || !out) return fail(KPDI_EINVAL, "NULL argument");
  if (n_eval <= 0 || n_eval >= (int64_t)INT_MAX) return fail(KPDI_EINVAL, "need between 1 and 2^31-1 evaluations");
  int rc = use_device(c);
  if (rc) return rc;
  kpdi::RefineLaunch a{};
  rc = refine_fill_launch(c, mode, &a);
  if (rc) return rc;
  if (a.nfixed > 0 && !fixed) return fail(KPDI_EINVAL, "this mode needs the `fixed` array");
  for (int64_t e = 0; e < n_eval; ++e)
    if (pattern_index[e] < 0 || pattern_index[e] >= c->ref_n)
      return fail(KPDI_EINVAL, "pattern index %d out of range at evaluation %lld", pattern_index[e], (long long)e);
  const size_t nx = (size_t)n_eval * a.nvar, nf = (size_t)n_eval * a.nfixed;
  HIPCHK(c->ref_in.reserve((nx + nf + 1) * sizeof(double)));
  HIPCHK(c->ref_idx.reserve((size_t)n_eval * sizeof(int)));
  HIPCHK(c->ref_out.reserve((size_t)n_eval * sizeof(double)));
  double *d_x = c->ref_in.as<double>(), *d_f = d_x + nx;
  HIPCHK(hipMemcpyAsync(d_x, x, nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (nf) HIPCHK(hipMemcpyAsync(d_f, fixed, nf * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->ref_idx.p, pattern_index, (size_t)n_eval * sizeof(int), hipMemcpyHostToDevice, c->stream));
  a.n_jobs = n_eval;
  a.x0 = d_x;
  a.fixed = d_f;
  HIPCHK(kpdi::launch_refine_objective(a, c->ref_idx.as<int>(), c->ref_out.as<double>(), c->stream));
  return results_to_host(c, out, c->ref_out.p, (size_t)n_eval * sizeof(double));
}

}  // extern "C"

namespace {
// SciPy's resolution of maxiter / maxfev (scipy/optimize/_optimize.py, _minimize_neldermead)
void resolve_budget(int nvar, int maxiter, int maxfev, int *it, int *fev) {
  const bool no_it = maxiter <= 0, no_fev = maxfev <= 0;
  if (no_it && no_fev) {
    *it = nvar * 200;
    *fev = nvar * 200;
  } else if (no_it) {
    *it = INT_MAX;
    *fev = maxfev;
  } else if (no_fev) {
    *it = maxiter;
    *fev = INT_MAX;
  } else {
    *it = maxiter;
    *fev = maxfev;
  }
}
}  // namespace

extern "C" {

int kpdi_refine_solve(kpdi_ctx *c, int mode, int64_t n_patterns, int n_starts, const double *x0, const double *fixed,
                      const double *lower, const double *upper, double xatol, double fatol, int maxiter, int maxfev,
                      double *results) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!x0 || !results) return fail(KPDI_EINVAL, "NULL argument");
  if ((lower == nullptr) != (upper == nullptr)) return fail(KPDI_EINVAL, "give both bounds or neither");
  if (n_starts <= 0) return fail(KPDI_EINVAL, "need at least one start per pattern");
  int rc = use_device(c);
  if (rc) return rc;
  kpdi::RefineLaunch a{};
  rc = refine_fill_launch(c, mode, &a);
  if (rc) return rc;
  if (n_patterns != c->ref_n)
    return fail(KPDI_EINVAL, "%lld patterns were set but starts for %lld were given", (long long)c->ref_n,
                (long long)n_patterns);
  if (a.nfixed > 0 && !fixed) return fail(KPDI_EINVAL, "this mode needs the `fixed` array");
  const int64_t jobs = n_patterns * n_starts;
  if (jobs >= (int64_t)INT_MAX) return fail(KPDI_EINVAL, "too many (pattern, start) pairs");
  const size_t nx = (size_t)jobs * a.nvar, nf = (size_t)jobs * a.nfixed;
  if (lower)
    for (size_t i = 0; i < nx; ++i)
      if (lower[i] > upper[i])
        return fail(KPDI_EINVAL, "Nelder Mead - one of the lower bounds is greater than an upper bound.");
  const size_t total = nx * (lower ? 3 : 1) + nf + 1;
  HIPCHK(c->ref_in.reserve(total * sizeof(double)));
  HIPCHK(c->ref_out.reserve((size_t)jobs * kpdi::REFINE_RESULT_STRIDE * sizeof(double)));
  double *d_x = c->ref_in.as<double>(), *d_f = d_x + nx, *d_lo = d_f + nf, *d_hi = d_lo + nx;
  HIPCHK(hipMemcpyAsync(d_x, x0, nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (nf) HIPCHK(hipMemcpyAsync(d_f, fixed, nf * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (lower) {
    HIPCHK(hipMemcpyAsync(d_lo, lower, nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_hi, upper, nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  HIPCHK(hipMemsetAsync(c->ref_out.p, 0, (size_t)jobs * kpdi::REFINE_RESULT_STRIDE * sizeof(double), c->stream));
  a.n_jobs = jobs;
  a.n_starts = n_starts;
  a.x0 = d_x;
  a.fixed = d_f;
  a.lower = lower ? d_lo : nullptr;
  a.upper = lower ? d_hi : nullptr;
  a.xatol = xatol;
  a.fatol = fatol;
  resolve_budget(a.nvar, maxiter, maxfev, &a.maxiter, &a.maxfun);
  a.results = c->ref_out.as<double>();
  hipEvent_t e0 = c->get_event(), e1 = c->get_event();
  HIPCHK(hipEventRecord(e0, c->stream));
  HIPCHK(kpdi::launch_refine_solve(a, c->stream));
  HIPCHK(hipEventRecord(e1, c->stream));
  rc = results_to_host(c, results, c->ref_out.p, (size_t)jobs * kpdi::REFINE_RESULT_STRIDE * sizeof(double));
  if (rc) return rc;
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  c->cnt.refine_ms += ms;
  c->ev_pool.push_back(e0);
  c->ev_pool.push_back(e1);
  return KPDI_OK;
}

int kpdi_nelder_mead_selftest(kpdi_ctx *c, int kind, int nvar, const double *x0, const double *lower,
                              const double *upper, double xatol, double fatol, int maxiter, int maxfev,
                              double *result) {
  if (!c || !x0 || !result) return fail(KPDI_EINVAL, "NULL argument");
  if (nvar < 1 || nvar > 6) return fail(KPDI_EINVAL, "nvar must be within 1..6");
  if ((lower == nullptr) != (upper == nullptr)) return fail(KPDI_EINVAL, "give both bounds or neither");
  int rc = use_device(c);
  if (rc) return rc;
  HIPCHK(c->ref_in.reserve((size_t)(3 * nvar + 1) * sizeof(double)));
  HIPCHK(c->ref_out.reserve((size_t)kpdi::REFINE_RESULT_STRIDE * sizeof(double)));
  double *d_x = c->ref_in.as<double>(), *d_lo = d_x + nvar, *d_hi = d_lo + nvar;
  HIPCHK(hipMemcpyAsync(d_x, x0, nvar * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (lower) {
    HIPCHK(hipMemcpyAsync(d_lo, lower, nvar * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_hi, upper, nvar * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  int it, fev;
  resolve_budget(nvar, maxiter, maxfev, &it, &fev);
  HIPCHK(kpdi::launch_nelder_mead_selftest(kind, nvar, d_x, lower ? d_lo : nullptr, lower ? d_hi : nullptr, xatol,
                                           fatol, it, fev, c->ref_out.as<double>(), c->stream));
  HIPCHK(hipMemcpyAsync(result, c->ref_out.p, (size_t)(3 + nvar) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return KPDI_OK;
}

// ---- orientation similarity map ---------------------------------------------------
int kpdi_orientation_similarity_map(kpdi_ctx *c, const int64_t *simulation_indices, int ny, int nx, int keep_n,
                                    int n_best, int from_n_best, const int32_t *footprint_offsets, int n_fp,
                                    int center_index, int normalize, float *out) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!footprint_offsets || !out) return fail(KPDI_EINVAL, "NULL argument");
  if (ny <= 0 || nx <= 0 || (int64_t)ny * nx >= (int64_t)INT_MAX) return fail(KPDI_EINVAL, "bad map shape");
  if (keep_n <= 0) return fail(KPDI_EINVAL, "keep_n must be positive");
  if (n_best > keep_n) return fail(KPDI_EINVAL, "n_best %d cannot be greater than keep_n %d", n_best, keep_n);
  if (from_n_best < 1 || from_n_best > n_best) return fail(KPDI_EINVAL, "from_n_best must be within 1..n_best");
  if (n_fp < 1 || n_fp > 64) return fail(KPDI_EINVAL, "the footprint must have between 1 and 64 points");
  if (center_index < 0 || center_index >= n_fp) return fail(KPDI_EINVAL, "center_index outside the footprint");
  int rc = use_device(c);
  if (rc) return rc;
  const size_t n_points = (size_t)ny * nx, n = n_points * keep_n;
  const int *d_idx = nullptr;
  if (simulation_indices) {
    std::vector<int> tmp(n);
    for (size_t i = 0; i < n; ++i) {
      if (simulation_indices[i] < INT_MIN || simulation_indices[i] > INT_MAX)
        return fail(KPDI_EINVAL, "simulation index %lld does not fit 32 bits", (long long)simulation_indices[i]);
      tmp[i] = (int)simulation_indices[i];
    }
    HIPCHK(c->osm_idx.reserve(n * sizeof(int)));
    HIPCHK(hipMemcpyAsync(c->osm_idx.p, tmp.data(), n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    d_idx = c->osm_idx.as<int>();
  } else {
    if (!c->final_valid) return fail(KPDI_EINVAL, "no resident result: call kpdi_finalize first");
    if ((size_t)c->m != n_points || c->keep_n != keep_n)
      return fail(KPDI_EINVAL, "the resident result is %d x %d but a %d x %d map with keep_n %d was asked for", c->m,
                  c->keep_n, ny, nx, keep_n);
    d_idx = c->final_idx;
  }
  const int n_layers = n_best - from_n_best + 1;
  HIPCHK(c->osm_out.reserve(n_points * n_layers * sizeof(float)));
  HIPCHK(kpdi::launch_osm(d_idx, ny, nx, keep_n, n_best, from_n_best, footprint_offsets, n_fp, center_index,
                          normalize != 0, c->osm_out.as<float>(), c->stream));
  return results_to_host(c, out, c->osm_out.p, n_points * n_layers * sizeof(float));
}

}  // extern "C"
