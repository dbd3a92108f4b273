// api.hip - host side of libkpdi.so, core: the C ABI of include/kpdi.h for a context's life cycle, the problem and the
// experimental set, the background-removal calls, device buffers and counters - on top of the kernels in prep.hip /
// preproc.hip.  The sweep itself lives in sweep.hip, float64 arithmetic in exact64.hip, the hand-over of the result and
// the communicators in finalize.hip, dictionary generation / refinement / OSM in extras.hip (split in round 5; they
// share context.h).
#include "context.h"

using namespace kpdi;

namespace {
thread_local std::string g_err;
}

namespace kpdi {

int fail(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

// for the translation units that do not include context.h (group.hip, h5ebsd.hip)
int fail_msg(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  return fail(code, "%s", buf);
}

const char *thread_error() { return g_err.c_str(); }

int drain_events(kpdi_ctx *c, std::vector<std::pair<hipEvent_t, hipEvent_t>> &list, double *ms_sum) {
  for (auto &pr : list) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, pr.first, pr.second));
    *ms_sum += ms;
    c->ev_pool.push_back(pr.first);
    c->ev_pool.push_back(pr.second);
  }
  list.clear();
  return KPDI_OK;
}

// the copies of the last result (finalize_enqueue) read the running lists: whoever writes those next waits for them
int wait_result_copy(kpdi_ctx *c) {
  if (c->result_copy) {
    HIPCHK(hipStreamWaitEvent(c->stream, c->result_copy, 0));
    c->result_copy = nullptr;
  }
  return KPDI_OK;
}

// ---- queued initialisations: one launch (kernels.h: FillSegments) instead of one per buffer
int flush_fills(kpdi_ctx *c) {
  if (c->fills.n == 0) return KPDI_OK;
  {
    ScopedTimer t(c, &c->ev_fixed);
    HIPCHK(kpdi::launch_fill_segments(c->fills, c->stream));
  }
  c->fills.n = 0;
  return KPDI_OK;
}
int queue_fill(kpdi_ctx *c, void *p, size_t words, unsigned value, int bound_used) {
  if (words == 0) return KPDI_OK;
  if (c->fills.n == kpdi::FILL_SEGMENTS) {
    int rc = flush_fills(c);
    if (rc) return rc;
  }
  const int i = c->fills.n++;
  c->fills.p[i] = (unsigned *)p;
  c->fills.words[i] = words;
  c->fills.value[i] = value;
  c->fills.bound_used[i] = bound_used;
  return KPDI_OK;
}
constexpr unsigned BITS_NEG_INF = 0xff800000u, BITS_INT_MAX = 0x7fffffffu;
int queue_fill_topk(kpdi_ctx *c, float *scores, int *idx, size_t n) {
  int rc = queue_fill(c, scores, n, BITS_NEG_INF);
  return rc ? rc : queue_fill(c, idx, n, BITS_INT_MAX);
}

void dtype_range(int dtype, float *omin, float *omax) {
  // skimage.util.dtype.dtype_range, as used at signals/ebsd.py:523 and :676
  switch (dtype) {
    case KPDI_U8: *omin = 0.f; *omax = 255.f; break;
    case KPDI_U16: *omin = 0.f; *omax = 65535.f; break;
    case KPDI_I8: *omin = -128.f; *omax = 127.f; break;
    case KPDI_I16: *omin = -32768.f; *omax = 32767.f; break;
    default: *omin = -1.f; *omax = 1.f; break;  // float32 / float64
  }
}

// what the prep kernels are told: `ndp` is evaluated in its centred form (prep.hip) except in the
// first thing every entry point does.  `keep_pending`: the one caller (kpdi_push_dictionary_chunk) that starts its upload
// BEFORE it looks at the float64 certification of the previous chunk
int use_device(kpdi_ctx *c, bool keep_pending) {
  HIPCHK(hipSetDevice(c->device));
  if (c->pend64.active && !keep_pending) return resolve_exact64(c);
  return KPDI_OK;
}

static int set_experimental_common(kpdi_ctx *c, const void *src, bool src_on_device, int dtype, int64_t m_all,
                            const uint8_t *nav_mask) {
  if (!c->have_problem) return fail(KPDI_EINVAL, "kpdi_set_problem must be called before kpdi_set_experimental");
  const size_t es = kpdi::dtype_size(dtype);
  if (es == 0) return fail(KPDI_EINVAL, "unknown dtype %d", dtype);
  if (m_all <= 0) return fail(KPDI_EINVAL, "need at least one experimental pattern");
  if (!src) return fail(KPDI_EINVAL, "patterns pointer is NULL");
  const size_t bytes = (size_t)m_all * c->npix * es;
  HIPCHK(c->exp_raw.reserve(bytes));
  HIPCHK(hipMemcpyAsync(c->exp_raw.p, src, bytes, src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                        c->stream));
  if (!src_on_device) c->cnt.h2d_bytes += (double)bytes;
  c->exp_dtype = dtype;
  c->m_all = m_all;
  c->have_nav_mask = nav_mask != nullptr;
  c->pend = kpdi_ctx::PendingPre{};  // recorded steps belonged to the previous set
  if (nav_mask) {
    if (m_all >= (int64_t)INT_MAX) return fail(KPDI_EINVAL, "too many experimental patterns");
    std::vector<int> rows, inv((size_t)m_all, -1);  // kept pattern -> source row; source row -> kept pattern or -1
    rows.reserve((size_t)m_all);
    for (int64_t i = 0; i < m_all; ++i)
      if (!nav_mask[i]) {
        inv[(size_t)i] = (int)rows.size();
        rows.push_back((int)i);
      }
    c->m = (int)rows.size();
    HIPCHK(c->row_map.reserve(std::max<size_t>(rows.size(), 1) * sizeof(int)));
    HIPCHK(c->inv_map.reserve(inv.size() * sizeof(int)));
    if (!rows.empty())
      HIPCHK(hipMemcpyAsync(c->row_map.p, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->inv_map.p, inv.data(), inv.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));  // `rows` / `inv` die at scope exit
  } else {
    if (m_all >= (int64_t)INT_MAX) return fail(KPDI_EINVAL, "too many experimental patterns");
    c->m = (int)m_all;
  }
  c->m_pad = kpdi::round_up(std::max(c->m, 1), kpdi::TILE_EXP);
  c->have_exp = true;
  c->exp_prepared = false;
  c->run_valid = false;
  discard_pending(c);
  c->final_valid = false;
  return KPDI_OK;
}

// device -> caller's (pageable) buffer through the page-locked staging buffer, then synchronise
int results_to_host(kpdi_ctx *c, void *dst, const void *d_src, size_t bytes) {
  if (bytes == 0) {
    HIPCHK(hipStreamSynchronize(c->stream));
    return KPDI_OK;
  }
  if (c->pin_out.reserve(bytes) != hipSuccess) {  // no page-locked memory to be had: copy directly
    (void)hipGetLastError();
    HIPCHK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return KPDI_OK;
  }
  HIPCHK(hipMemcpyAsync(c->pin_out.p, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  memcpy(dst, c->pin_out.p, bytes);
  return KPDI_OK;
}

}  // namespace kpdi

extern "C" {

const char *kpdi_version(void) { return "kpdi 0.2.0 (gfx950)"; }

size_t kpdi_counters_size(void) { return sizeof(kpdi_counters); }

const char *kpdi_last_error(void) { return g_err.c_str(); }

int kpdi_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int kpdi_create(int device_id, kpdi_ctx **out) {
  if (!out) return fail(KPDI_EINVAL, "out is NULL");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(KPDI_ENODEV, "no HIP device visible: libkpdi has no CPU fallback");
  if (device_id < 0 || device_id >= n) return fail(KPDI_EINVAL, "device %d out of range [0, %d)", device_id, n);
  HIPCHK(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(KPDI_ENODEV, "device %d is %s; libkpdi is built for gfx950 (MI355X) only", device_id, prop.gcnArchName);
  kpdi_ctx *c = new kpdi_ctx();
  c->device = device_id;
  c->n_cu = prop.multiProcessorCount;
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete c;
    return fail(KPDI_EHIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
  }
  *out = c;
  return KPDI_OK;
}

int kpdi_destroy(kpdi_ctx *c) {
  if (!c) return KPDI_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  release_held(c);
  c->pin_out.release();
  for (auto &st : c->rot_stage) {
    st.pin.release();
    if (st.copied) (void)hipEventDestroy(st.copied);
  }
  c->pend64.flag.release();
  if (c->pend64.ready) (void)hipEventDestroy(c->pend64.ready);
  for (auto &rs : c->slots) {
    rs.pin.release();
    if (rs.ready) (void)hipEventDestroy(rs.ready);
  }
  if (c->result_done) (void)hipEventDestroy(c->result_done);
  if (c->result_stream) (void)hipStreamDestroy(c->result_stream);
  if (c->lists_final) (void)hipEventDestroy(c->lists_final);
  if (c->peer_read) (void)hipEventDestroy(c->peer_read);
  for (DevBuf *b : {&c->pix_map, &c->quad_desc, &c->exp_raw, &c->row_map, &c->exp_x, &c->dict_raw, &c->dict_y, &c->part_s,
                    &c->part_i, &c->part_cnt, &c->tail_s, &c->tail_i, &c->tail_scores, &c->list16, &c->run_s[0], &c->run_s[1], &c->run_i[0], &c->run_i[1], &c->loc_s, &c->loc_i,
                    &c->bound_s, &c->bound_i, &c->gthr, &c->tile_ctr, &c->gather_s, &c->gather_i, &c->bg, &c->taps, &c->inv_map, &c->pre_scratch,
                    &c->mp_packed, &c->dcos, &c->rot, &c->proj_out,
                    &c->ref_raw, &c->ref_map, &c->ref_rowcol, &c->ref_pat, &c->ref_sqn, &c->ref_in, &c->ref_out,
                    &c->ref_idx, &c->osm_idx, &c->osm_out, &c->stage[0], &c->stage[1], &c->pending.raw, &c->pending.raw_b, &c->pending_hold.raw})
    b->release();
  for (auto *l : {&c->ev_match, &c->ev_prep, &c->ev_merge, &c->ev_proj, &c->ev_pre, &c->ev_rescore})
    for (auto &pr : *l) {
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  for (int b = 0; b < 2; ++b) {
    if (c->stage_filled[b]) (void)hipEventDestroy(c->stage_filled[b]);
    if (c->stage_free[b]) (void)hipEventDestroy(c->stage_free[b]);
    if (c->pending.consumed[b]) (void)hipEventDestroy(c->pending.consumed[b]);
  }
  if (c->pending.filled) (void)hipEventDestroy(c->pending.filled);
  if (c->copy_stream) {
    (void)hipStreamSynchronize(c->copy_stream);
    (void)hipStreamDestroy(c->copy_stream);
  }
  if (c->stream2) {
    (void)hipStreamSynchronize(c->stream2);
    (void)hipStreamDestroy(c->stream2);
    (void)hipEventDestroy(c->ev_fork);
    (void)hipEventDestroy(c->ev_join);
  }
  (void)hipStreamDestroy(c->stream);
  delete c;
  return KPDI_OK;
}

int kpdi_synchronize(kpdi_ctx *c) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  int rc = use_device(c);
  if (rc) return rc;
  if (c->have_exp && c->have_problem) {
    rc = flush_pending(c);  // "everything pushed has been swept"
    if (rc) return rc;
  }
  if (c->have_problem) {
    rc = flush_pending(c, true);  // ... and everything handed over to be held is prepared
    if (rc) return rc;
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->result_stream) HIPCHK(hipStreamSynchronize(c->result_stream));
  return KPDI_OK;
}

int kpdi_set_problem(kpdi_ctx *c, int sy, int sx, const uint8_t *signal_mask, int metric, int compute_dtype,
                     int keep_n) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (sy <= 0 || sx <= 0) return fail(KPDI_EINVAL, "detector shape (%d, %d) must be positive", sy, sx);
  if (metric != KPDI_METRIC_NCC && metric != KPDI_METRIC_NDP) return fail(KPDI_EINVAL, "unknown metric %d", metric);
  if (compute_dtype != KPDI_COMPUTE_F32 && compute_dtype != KPDI_COMPUTE_F16X2 && compute_dtype != KPDI_COMPUTE_F16 &&
      compute_dtype != KPDI_COMPUTE_F64)
    return fail(KPDI_EINVAL, "unknown compute dtype %d", compute_dtype);
  // float64 arithmetic = the f32 path as the screen + rescoring in double (rescore.hip)
  const bool exact64 = compute_dtype == KPDI_COMPUTE_F64;
  if (exact64) compute_dtype = KPDI_COMPUTE_F32;
  if (keep_n <= 0) return fail(KPDI_EINVAL, "keep_n must be >= 1");
  int rc = use_device(c);
  if (rc) return rc;
  c->sw.read();
  const int npix = sy * sx;
  std::vector<int> keep;
  if (signal_mask) {
    for (int i = 0; i < npix; ++i)
      if (!signal_mask[i]) keep.push_back(i);
    if (keep.empty()) return fail(KPDI_EINVAL, "the signal mask excludes every pixel");
    HIPCHK(c->pix_map.reserve(keep.size() * sizeof(int)));
    HIPCHK(hipMemcpyAsync(c->pix_map.p, keep.data(), keep.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    std::vector<unsigned> desc;
    c->have_quad_desc = kpdi::gather_descriptors(keep.data(), (int)keep.size(), npix, &desc);
    if (c->have_quad_desc) {
      HIPCHK(c->quad_desc.reserve(desc.size() * sizeof(unsigned)));
      HIPCHK(hipMemcpyAsync(c->quad_desc.p, desc.data(), desc.size() * sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
  } else {
    c->have_quad_desc = false;
  }
  if ((c->pend.st || c->pend.dy) && c->have_exp && (sy != c->sy || sx != c->sx)) {
    bool dummy = false;  // recorded background-removal steps belong to the old detector shape
    rc = flush_preprocess(c, false, &dummy);
    if (rc) return rc;
  }
  if (npix != c->npix) c->have_exp = false;  // resident patterns belong to another detector shape
  // the prepared layout of held chunks depends on shape, mask, metric and arithmetic
  int waves = 8;
  if (const char *e = getenv("KPDI_F16_WAVES")) waves = atoi(e) == 4 ? 4 : 8;
  // the f32 kernel in use (match.hip / the one-wave form of match16.hip) is chosen when the first chunk of a sweep arrives
  // (decide_form); until then - and whenever it cannot change any more - the previous choice stands
  const int wide_mode = getenv("KPDI_F32_WIDE") ? (atoi(getenv("KPDI_F32_WIDE")) != 0) : -1;
  bool wide32 = compute_dtype == KPDI_COMPUTE_F32 && (wide_mode == 1 || (wide_mode < 0 && c->have_problem && c->wide32));
  if (!c->have_problem || sy != c->sy || sx != c->sx || metric != c->metric || compute_dtype != c->compute ||
      (signal_mask != nullptr) != c->have_sig_mask || keep != c->kept_pixels || wide32 != c->wide32)
    release_held(c);
  c->f16_waves = waves;
  c->wide32 = wide32;
  c->wide_mode = wide_mode;
  c->kept_pixels = keep;
  c->sy = sy;
  c->sx = sx;
  c->npix = npix;
  c->have_sig_mask = signal_mask != nullptr;
  c->k_kept = signal_mask ? (int)keep.size() : npix;
  // floats per prepared row; the float16 form packs two pixels into one float: steps of 48 pixels
  // (match16.hip).  `ndp` rows carry one extra column (prep.hip: centred evaluation), except in the float16 form
  c->kpad = compute_dtype == KPDI_COMPUTE_F16
                ? kpdi::round_up(c->k_kept, kpdi::f16_geometry(c->f16_waves).step) / 2
                : kpdi::round_up(c->k_kept + (metric == KPDI_METRIC_NDP ? 1 : 0), wide32 ? kpdi::F16_STEP / 2 : kpdi::TILE_K);
  c->metric = metric;
  c->compute = compute_dtype;
  c->exact64 = exact64;
  c->keep_n = keep_n;
  c->have_problem = true;
  c->exp_prepared = false;
  c->run_valid = false;
  discard_pending(c);
  c->final_valid = false;
  c->cnt.kpad = c->kpad;
  c->cnt.k_kept = c->k_kept;
  return KPDI_OK;
}

int kpdi_set_keep_n(kpdi_ctx *c, int keep_n) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!c->have_problem) return fail(KPDI_EINVAL, "kpdi_set_problem has not been called");
  if (keep_n <= 0) return fail(KPDI_EINVAL, "keep_n must be >= 1");
  c->keep_n = keep_n;
  c->run_valid = false;
  discard_pending(c);
  c->final_valid = false;
  return KPDI_OK;
}

int kpdi_set_experimental(kpdi_ctx *c, const void *patterns, int dtype, int64_t m_all, const uint8_t *nav_mask) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  int rc = use_device(c);
  if (rc) return rc;
  return set_experimental_common(c, patterns, false, dtype, m_all, nav_mask);
}

int kpdi_set_experimental_dev(kpdi_ctx *c, const void *d_patterns, int dtype, int64_t m_all,
                              const uint8_t *nav_mask) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  int rc = use_device(c);
  if (rc) return rc;
  return set_experimental_common(c, d_patterns, true, dtype, m_all, nav_mask);
}

int64_t kpdi_n_experimental(kpdi_ctx *c) { return c && c->have_exp ? c->m : 0; }

int kpdi_remove_static_background(kpdi_ctx *c, const float *static_bg, int operation, int scale_bg) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!c->have_exp) return fail(KPDI_EINVAL, "kpdi_set_experimental has not been called");
  if (!static_bg) return fail(KPDI_EINVAL, "static_bg is NULL");
  if (c->exp_dtype == KPDI_F16 || c->exp_dtype == KPDI_I32 || c->exp_dtype == KPDI_U32)
    return fail(KPDI_EINVAL, "background removal takes uint8/int8/uint16/int16/float32/float64 patterns");
  if (operation != KPDI_OP_SUBTRACT && operation != KPDI_OP_DIVIDE) return fail(KPDI_EINVAL, "unknown operation");
  int rc = use_device(c);
  if (rc) return rc;
  // one static step followed by one dynamic step fuse into a single kernel; anything recorded that
  // this step cannot follow runs now
  bool dummy = false;
  if (c->pend.st || c->pend.dy) {
    rc = flush_preprocess(c, false, &dummy);
    if (rc) return rc;
  }
  HIPCHK(c->bg.reserve((size_t)c->npix * sizeof(float)));
  HIPCHK(hipMemcpyAsync(c->bg.p, static_bg, (size_t)c->npix * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));  // static_bg may be freed by the caller after return
  c->pend.st = true;
  c->pend.st_op = operation;
  c->pend.st_scale = scale_bg ? 1 : 0;
  c->pend.bg_min = *std::min_element(static_bg, static_bg + c->npix);
  c->pend.bg_max = *std::max_element(static_bg, static_bg + c->npix);
  c->exp_prepared = false;
  c->run_valid = false;
  discard_pending(c);
  c->final_valid = false;
  return KPDI_OK;
}

int kpdi_remove_dynamic_background(kpdi_ctx *c, int operation, int filter_domain, double std, double truncate) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!c->have_exp) return fail(KPDI_EINVAL, "kpdi_set_experimental has not been called");
  if (operation != KPDI_OP_SUBTRACT && operation != KPDI_OP_DIVIDE) return fail(KPDI_EINVAL, "unknown operation");
  if (c->exp_dtype == KPDI_F16 || c->exp_dtype == KPDI_I32 || c->exp_dtype == KPDI_U32)
    return fail(KPDI_EINVAL, "background removal takes uint8/int8/uint16/int16/float32/float64 patterns");
  int rc = use_device(c);
  if (rc) return rc;
  if (std <= 0) std = c->sx / 8.0;  // signals/ebsd.py:648-649
  std::vector<double> taps;
  int n, centre, reflect;
  if (filter_domain == KPDI_DOMAIN_FREQUENCY) {
    // pattern/_pattern.py:604-613: n = int(truncate*std) samples of
    // scipy.signal.windows.gaussian, normalised; centre from filters/fft_barnes.py:106-117
    n = (int)(truncate * std);
    if (n < 1) return fail(KPDI_EINVAL, "Gaussian window of int(truncate*std) = %d samples", n);
    taps.resize(n);
    double sum = 0;
    for (int i = 0; i < n; ++i) {
      const double x = i - (n - 1) / 2.0;
      taps[i] = exp(-0.5 * (x / std) * (x / std));
      sum += taps[i];
    }
    for (double &t : taps) t /= sum;
    centre = n - 1 - (n - 1) / 2;
    reflect = 0;
  } else if (filter_domain == KPDI_DOMAIN_SPATIAL) {
    // scipy.ndimage.gaussian_filter(sigma=std, truncate=truncate), mode='reflect'
    const int r = (int)(truncate * std + 0.5);
    n = 2 * r + 1;
    taps.resize(n);
    double sum = 0;
    for (int i = 0; i < n; ++i) {
      const double x = i - r;
      taps[i] = exp(-0.5 / (std * std) * x * x);
      sum += taps[i];
    }
    for (double &t : taps) t /= sum;
    centre = r;
    reflect = 1;
  } else {
    return fail(KPDI_EINVAL, "unknown filter domain %d", filter_domain);
  }
  bool dummy = false;
  if (c->pend.dy) {  // a second dynamic step cannot join the recorded one
    rc = flush_preprocess(c, false, &dummy);
    if (rc) return rc;
  }
  // the kernels read the taps through a window of CONV_R outputs: zero padding on both sides
  std::vector<double> padded(taps.size() + 2 * (kpdi::CONV_R - 1), 0.0);
  std::copy(taps.begin(), taps.end(), padded.begin() + (kpdi::CONV_R - 1));
  HIPCHK(c->taps.reserve(padded.size() * sizeof(double)));
  HIPCHK(hipMemcpyAsync(c->taps.p, padded.data(), padded.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));  // `padded` dies at scope exit
  c->pend.dy = true;
  c->pend.dy_op = operation;
  c->pend.reflect = reflect;
  c->pend.ntaps = n;
  c->pend.centre = centre;
  c->exp_prepared = false;
  c->run_valid = false;
  discard_pending(c);
  c->final_valid = false;
  return KPDI_OK;
}

int kpdi_get_experimental(kpdi_ctx *c, void *out) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!c->have_exp) return fail(KPDI_EINVAL, "kpdi_set_experimental has not been called");
  int rc = use_device(c);
  if (rc) return rc;
  bool dummy = false;
  rc = flush_preprocess(c, false, &dummy);  // recorded background-removal steps run now
  if (rc) return rc;
  const size_t bytes = (size_t)c->m_all * c->npix * kpdi::dtype_size(c->exp_dtype);
  HIPCHK(hipMemcpyAsync(out, c->exp_raw.p, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return KPDI_OK;
}

size_t kpdi_dtype_size(int dtype) { return kpdi::dtype_size(dtype); }

int kpdi_reset_topk(kpdi_ctx *c) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  int rc = use_device(c);
  if (rc) return rc;
  c->run_valid = false;
  discard_pending(c);
  c->final_valid = false;
  return KPDI_OK;
}

int kpdi_dev_alloc(kpdi_ctx *c, size_t bytes, void **d_out) {
  if (!c || !d_out) return fail(KPDI_EINVAL, "NULL argument");
  int rc = use_device(c);
  if (rc) return rc;
  HIPCHK(hipMalloc(d_out, bytes));
  return KPDI_OK;
}

int kpdi_dev_free(kpdi_ctx *c, void *d_ptr) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  int rc = use_device(c);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipFree(d_ptr));
  return KPDI_OK;
}

int kpdi_h2d(kpdi_ctx *c, void *d_dst, const void *src, size_t bytes) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  int rc = use_device(c);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return KPDI_OK;
}

int kpdi_d2h(kpdi_ctx *c, void *dst, const void *d_src, size_t bytes) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  int rc = use_device(c);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return KPDI_OK;
}

int kpdi_set_profiling(kpdi_ctx *c, int on) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (on < 0 || on > 3)
    return fail(KPDI_EINVAL, "profiling level %d (0 off, 1 every phase, 2 match launches only, 3 every phase + the epilogue counters)", on);
  c->profiling = on;
  return KPDI_OK;
}

int kpdi_get_counters(kpdi_ctx *c, kpdi_counters *out) {
  if (!c || !out) return fail(KPDI_EINVAL, "NULL argument");
  int rc = use_device(c);
  if (rc) return rc;
  if (c->have_exp && c->have_problem) {
    rc = flush_pending(c);  // (the counters cover everything pushed)
    if (rc) return rc;
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  rc = drain_events(c, c->ev_match, &c->cnt.match_ms);
  if (rc) return rc;
  rc = drain_events(c, c->ev_prep, &c->cnt.prep_ms);
  if (rc) return rc;
  rc = drain_events(c, c->ev_merge, &c->cnt.merge_ms);
  if (rc) return rc;
  rc = drain_events(c, c->ev_proj, &c->cnt.project_ms);
  if (rc) return rc;
  rc = drain_events(c, c->ev_pre, &c->cnt.preproc_ms);
  if (rc) return rc;
  rc = drain_events(c, c->ev_rescore, &c->cnt.rescore_ms);
  if (rc) return rc;
  rc = drain_events(c, c->ev_comm, &c->cnt.comm_ms);
  if (rc) return rc;
  rc = drain_events(c, c->ev_fixed, &c->cnt.fixed_ms);
  if (rc) return rc;
  if (c->epi_stats.p) {
    unsigned long long st[4];
    HIPCHK(hipMemcpy(st, c->epi_stats.p, sizeof st, hipMemcpyDeviceToHost));
    c->cnt.epi_lists = (int64_t)st[0];
    c->cnt.epi_appended = (int64_t)st[1];
    c->cnt.epi_overflows = (int64_t)st[2];
    c->cnt.epi_direct_first = (int64_t)st[3];
  }
  c->cnt.f64_certificate = c->exact64 ? (c->sw.f64_statistical ? 1 : 2) : 0;
  c->cnt.comm_ranks = 0;
  if (c->comm) {
    int count = 0;
    if (g_rccl.CommCount(c->comm, &count) == ncclSuccess) c->cnt.comm_ranks = count;
  }
  *out = c->cnt;
  return KPDI_OK;
}

int kpdi_reset_counters(kpdi_ctx *c) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  kpdi_counters tmp;
  int rc = kpdi_get_counters(c, &tmp);  // recycles pending events
  if (rc) return rc;
  const int kpad = c->cnt.kpad, kk = c->cnt.k_kept, gr = c->cnt.gather_ranks;
  c->cnt = kpdi_counters{};
  if (c->epi_stats.p) HIPCHK(hipMemsetAsync(c->epi_stats.p, 0, 4 * sizeof(unsigned long long), c->stream));
  c->cnt.kpad = kpad;
  c->cnt.k_kept = kk;
  c->cnt.gather_ranks = gr;
  return KPDI_OK;
}

}  // extern "C"
