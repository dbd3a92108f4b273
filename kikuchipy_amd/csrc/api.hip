// api.hip - host side of libkpdi.so: the C ABI of include/kpdi.h on top of the
// kernels in prep.hip / match.hip / merge.hip / preproc.hip.
//
// What one context holds (all in the HBM of ONE MI355X):
//   raw experimental patterns (m_all x npix, caller's dtype)      - pre-processed in place
//   prepared experimental matrix X (m_pad x kpad f32)             - built once per set
//   raw + prepared dictionary chunk Y (n_pad x kpad f32)          - rebuilt per chunk
//   per-lane partial lists of the match kernel                    - [m_pad][2*nsplit][len]
//   running best-k (m x k: f32 score, i32 dictionary index)       - ping-pong pair
// The running best-k is the whole state of the sweep, exactly as in the
// reference's loop (indexing/_dictionary_indexing.py:97-98).
#include "../../include/kpdi.h"
#include "kernels.h"
#include "form_model.h"
#include "group_hooks.h"

#include <dlfcn.h>
#include <limits.h>
#include <math.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess)                                                                \
      return fail(KPDI_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) {
      hipError_t e = hipFree(p);
      p = nullptr;
      cap = 0;
      if (e != hipSuccess) return e;
    }
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T>
  T *as() const { return (T *)p; }
};

// page-locked host staging (results come back through it: a device-to-host copy into pageable
// memory goes through the runtime's pin-on-the-fly path, measured at several ms per call and a
// slower following sweep for a 40 000 x 20 result)
struct PinBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    release();
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct Rccl {
  void *lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  std::string why;  // why the last load() failed
  bool load() {
    if (lib) return true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
      h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) {
      const char *e = dlerror();
      why = e ? e : "librccl.so not found";
      return false;
    }
    // resolve everything into locals; `lib` (= "loaded") is only published on full success
    Rccl t;
#define KPDI_SYM(field, name)                                         \
  t.field = (decltype(t.field))dlsym(h, name);                         \
  if (!t.field) {                                                     \
    const char *e = dlerror();                                        \
    why = std::string("symbol ") + name + ": " + (e ? e : "missing"); \
    dlclose(h);                                                       \
    return false;                                                     \
  }
    KPDI_SYM(GetUniqueId, "ncclGetUniqueId")
    KPDI_SYM(CommInitRank, "ncclCommInitRank")
    KPDI_SYM(CommInitAll, "ncclCommInitAll")
    KPDI_SYM(CommDestroy, "ncclCommDestroy")
    KPDI_SYM(AllGather, "ncclAllGather")
    KPDI_SYM(GroupStart, "ncclGroupStart")
    KPDI_SYM(GroupEnd, "ncclGroupEnd")
    KPDI_SYM(GetErrorString, "ncclGetErrorString")
    KPDI_SYM(CommCount, "ncclCommCount")
#undef KPDI_SYM
    GetUniqueId = t.GetUniqueId;
    CommInitRank = t.CommInitRank;
    CommInitAll = t.CommInitAll;
    CommDestroy = t.CommDestroy;
    AllGather = t.AllGather;
    GroupStart = t.GroupStart;
    GroupEnd = t.GroupEnd;
    GetErrorString = t.GetErrorString;
    CommCount = t.CommCount;
    lib = h;
    return true;
  }
};
Rccl g_rccl;

}  // namespace

namespace kpdi {
// for the other translation units with extern "C" entry points (h5ebsd.hip)
int fail_msg(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  return fail(code, "%s", buf);
}
}  // namespace kpdi

// Developer / A-B switches: ONE convention - read from the environment by kpdi_set_problem and
// fixed for the context until the next kpdi_set_problem (a variable changed in between is seen then, never mid-sweep).
struct Switches {
  double odd_wide = kpdi::FORM_ODD_SPLIT_WIDE, odd_classic = kpdi::FORM_ODD_SPLIT_CLASSIC;
  double wide_launch = kpdi::FORM_WIDE_LAUNCH, fixed_frac = 0.8;
  bool xcd_grid = true, xcd_pad = true, no_tail = false, one_stream = false, tail_stream2 = false;
  bool f64_worstcase = false, f64_sync = false;
  long upload_tiles = 0;
  void read() {
    *this = Switches{};
    auto num = [](const char *name, double dflt) {
      const char *e = getenv(name);
      return e ? atof(e) : dflt;
    };
    odd_wide = num("KPDI_NS_ODD_WIDE", odd_wide);
    odd_classic = num("KPDI_NS_ODD_CLASSIC", odd_classic);
    wide_launch = num("KPDI_FORM_WIDE_LAUNCH", wide_launch);
    fixed_frac = num("KPDI_FIXED_FRAC", fixed_frac);
    if (const char *e = getenv("KPDI_XCD_GRID")) xcd_grid = atoi(e) != 0;
    if (const char *e = getenv("KPDI_XCD_PAD")) xcd_pad = atoi(e) != 0;
    no_tail = getenv("KPDI_NO_TAIL") != nullptr;
    one_stream = getenv("KPDI_ONE_STREAM") != nullptr;
    tail_stream2 = getenv("KPDI_TAIL_STREAM2") != nullptr;
    if (const char *e = getenv("KPDI_F64_EPS")) f64_worstcase = !strcmp(e, "worstcase");
    f64_sync = getenv("KPDI_F64_SYNC") != nullptr;
    if (const char *e = getenv("KPDI_UPLOAD_TILES")) upload_tiles = atol(e);
  }
};

struct kpdi_ctx {
  int device = 0;
  int n_cu = 256;
  hipStream_t stream = nullptr;
  Switches sw;

  // problem
  bool have_problem = false;
  int sy = 0, sx = 0, npix = 0;
  int k_kept = 0, kpad = 0;
  bool have_sig_mask = false;
  DevBuf pix_map;  // int[k_kept]
  // signal mask as gather descriptors, one per 4 kept pixels (prep.hip: prep_wave_gather_kernel), when every such
  // quad lies in at most two runs of consecutive detector pixels (a circular mask: one run per detector row)
  DevBuf quad_desc;
  bool have_quad_desc = false;
  int metric = KPDI_METRIC_NCC;
  int compute = KPDI_COMPUTE_F32;
  int f16_waves = 8;  // variant of the float16 kernel (match16.hip), fixed per problem: KPDI_F16_WAVES = 8 | 4
  // KPDI_COMPUTE_F32 on match16.hip's one-wave-per-SIMD kernel (256 x 256 tiles, lists out of the registers, exact f32
  // MFMAs; operand form 3): fixed per problem, KPDI_F32_WIDE = 1 | 0
  bool wide32 = false;
  int wide_mode = -1;  // KPDI_F32_WIDE: 1 / 0 force the form, unset (-1): decided per sweep (decide_form)
  int keep_n = 0;

  // experimental
  bool have_exp = false, exp_prepared = false;
  int exp_dtype = KPDI_U8;
  int64_t m_all = 0;
  int m = 0, m_pad = 0;
  bool have_nav_mask = false;
  DevBuf exp_raw, row_map, exp_x;

  // dictionary chunk
  DevBuf dict_raw, dict_y;
  // prepared chunks kept resident for sweeps against several experimental sets
  struct HeldChunk {
    DevBuf y;
    int64_t n = 0, start = 0;
  };
  std::vector<HeldChunk> held;
  std::vector<int> kept_pixels;  // host copy of pix_map: tells whether a new problem keeps the layout
  // host-pointer pushes are cut into pieces whose upload (copy stream) overlaps the sweep of
  // the previous piece (compute stream): two staging buffers, events for hand-over
  DevBuf stage[2];
  hipStream_t stream2 = nullptr;  // second compute stream of multi-launch sweeps
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipStream_t copy_stream = nullptr;
  int stage_next = 0;
  hipEvent_t stage_filled[2] = {nullptr, nullptr}, stage_free[2] = {nullptr, nullptr};

  // top-k state
  DevBuf part_s, part_i;       // partial lists of one match launch
  DevBuf tail_s, tail_i;       // partial lists of the quarter-tile tail launch (match.hip: ROWT = 1)
  DevBuf list16;               // float16 form: home of the per-lane lists during a launch (match16.hip)
  int tail_nsplit = 0;         // lists per pattern / 2 of the last run_match's tail launch, 0 = none
  DevBuf run_s[2], run_i[2];   // running best-k ping-pong
  int run_cur = 0;
  bool run_valid = false;
  bool run_empty = true;          // no chunk merged yet: the running lists hold nothing (and are not initialised)
  kpdi::FillSegments fills;       // small initialisations queued for ONE launch (queue_fill / flush_fills)
  const float *tail_queued = nullptr;  // prepared chunk whose partial last tile is already queued for zeroing
  // the match launch's bookkeeping (bound / counters) queued ahead of the preparation kernels by push_chunk_dev
  struct MatchSetup {
    bool valid = false;
    int n_chunk = 0, n_tiles = 0, nsplit = 0, rows_per_launch = 0, list_len = 0;
  } presetup;
  struct MatchPlan {
    int tail_tiles = 0, n_main = 0, fixed_draws = 3, bound_rank = 1, bound_grouped = 0, tail_units = 0, tail_nsplit = 0;
  } preplan;
  bool final_valid = false;       // `final_idx` points at the lists kpdi_finalize handed out last
  const int *final_idx = nullptr;
  DevBuf osm_idx, osm_out;
  DevBuf gthr;                            // shared rejection bound of the match kernel
  int bound_key = -1;                     // plan the bound array was initialised for (-1: none)
  DevBuf tile_ctr;                        // dynamic tile counters of the match kernel
  DevBuf loc_s, loc_i, bound_s, bound_i;  // multi-pass (keep_n > 32)
  DevBuf gather_s, gather_i;              // RCCL all-gather target
  // float64 arithmetic (KPDI_COMPUTE_F64): the f32 path screens, rescore.hip rescores and keeps the best-k in double
  bool exact64 = false;
  DevBuf run64_s, run64_i;                // running float64 best-k [m][keep_n]
  DevBuf cand64;                          // float64 scores of the screened candidates [m][columns]
  DevBuf cert64;                          // [0]: bits of max |f32 - f64| over the sweep; [1]: uncertified patterns of a merge
  DevBuf gather64_s, gather64_i, final64_s, final64_i;
  PinBuf pin_out;                         // float64 results on their way to the caller
  // the certification read-back of the last float64 chunk, not yet looked at (sweep_exact64 / resolve_exact64)
  struct Pending64 {
    bool active = false, defer = false;
    const float *y = nullptr;
    const void *raw = nullptr;
    int raw_dtype = 0, n_tiles = 0, nsplit = 0, rows_per_launch = 0, cap = 0, done = 0, extra = 0;
    int64_t n_chunk = 0, global_start = 0;
    hipEvent_t ready = nullptr;
    PinBuf flag;  // int: patterns the last merge could not certify
  } pend64;
  // kpdi_finalize[_async]: two page-locked slots (scores + indices on their way to the caller) with an event each
  struct ResultSlot {
    PinBuf pin;
    hipEvent_t ready = nullptr;
    size_t n = 0;
    bool pending = false;
  } slots[2];
  int next_slot = 0;
  // the copies of a result run on a stream of their own (the next map's kernels need not queue behind them); whoever
  // next WRITES the lists they read (the merge into the running best-k) waits for `result_copy` first
  hipStream_t result_stream = nullptr;
  hipEvent_t result_done = nullptr;   // compute stream: the lists of the result are final
  hipEvent_t result_copy = nullptr;   // = slots[].ready of the copy still to be waited for, or nullptr
  const int32_t *result_i32 = nullptr;    // the indices of the last kpdi_finalize in that buffer (kpdi_result_indices_i32)
  int64_t result_n = 0;

  // pre-processing: kpdi_remove_*_background only RECORD the step; the kernels run (fused with the
  // preparation of the patterns when those are about to be matched) in flush_preprocess
  struct PendingPre {
    bool st = false, dy = false;
    int st_op = 0, st_scale = 0;
    float bg_min = 0.f, bg_max = 0.f;
    int dy_op = 0, reflect = 0, ntaps = 0, centre = 0;
  } pend;
  DevBuf bg, taps, inv_map, pre_scratch;

  // dictionary generation (project.hip)
  bool have_master = false, have_dc = false;
  int mp_npx = 0, mp_npy = 0;
  int64_t dc_npix = 0;
  DevBuf mp_packed, dcos, rot, proj_out;

  // refinement (refine.hip)
  bool have_ref = false;
  int ref_nrows = 0, ref_ncols = 0, ref_k = 0;
  int64_t ref_n = 0;
  double ref_om[9] = {};
  DevBuf ref_raw, ref_map, ref_rowcol, ref_pat, ref_sqn, ref_in, ref_out, ref_idx;

  // comm
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  // in-process groups (group.hip): the members' lists peer-copied into gather_s / gather_i (gather64_*) of the ROOT
  // member instead of an RCCL all-gather; `p2p_ranks` > 0 = that many lists are waiting there for the next finalize
  int p2p_ranks = 0;
  hipEvent_t lists_final = nullptr;  // this member's running lists are final (recorded on `stream`)
  hipEvent_t peer_read = nullptr;    // root: the peer copies of the members' lists have run

  // measurement: 0 off; 1 every phase bracketed by HIP events; 2 the match launches (and the all-gather) only - an event
  // record between two kernels costs ~6 us of idle GPU (profiles/r04_share_timeline.txt: 71 us per 3 ms step with level 1)
  int profiling = 0;
  bool timed(const void *list) const { return profiling == 1 || (profiling == 2 && (list == &ev_match || list == &ev_comm)); }
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_match, ev_prep, ev_merge, ev_proj, ev_pre, ev_rescore, ev_comm, ev_fixed;
  std::vector<hipEvent_t> ev_pool;
  kpdi_counters cnt{};

  hipEvent_t get_event() {
    if (!ev_pool.empty()) {
      hipEvent_t e = ev_pool.back();
      ev_pool.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
};

namespace {

struct ScopedTimer {
  kpdi_ctx *c;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> *list;
  hipEvent_t a = nullptr, b = nullptr;
  ScopedTimer(kpdi_ctx *ctx, std::vector<std::pair<hipEvent_t, hipEvent_t>> *l) : c(ctx), list(l) {
    if (c->timed(list)) {
      a = c->get_event();
      b = c->get_event();
      (void)hipEventRecord(a, c->stream);
    }
  }
  ~ScopedTimer() {
    if (a) {
      (void)hipEventRecord(b, c->stream);
      list->push_back({a, b});
    }
  }
};

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
int queue_fill(kpdi_ctx *c, void *p, size_t words, unsigned value, int bound_used = -1) {
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
// float16 form
int prep_metric(const kpdi_ctx *c) {
  return c->metric == KPDI_METRIC_NDP && c->compute != KPDI_COMPUTE_F16 ? 2 : c->metric;
}

// the match kernel in use is match16.hip's (plane-major operand blocks, tiles of 256, lists in scratch)
bool uses16(const kpdi_ctx *c) { return c->compute == KPDI_COMPUTE_F16 || c->wide32; }
// operand form of the prepared matrices: 0 f32 tiled, 1 split f16, 2 float16, 3 f32 plane-major (kernels.h)
int operand_form(const kpdi_ctx *c) { return c->wide32 ? 3 : c->compute; }

// patterns per dictionary tile of the match kernel in use (the float16 form has its own kernel)
int dict_tile(const kpdi_ctx *c) {
  return uses16(c) ? kpdi::f16_geometry(c->f16_waves).dict_tile : kpdi::TILE_DICT;
}
// lists per pattern and dictionary split the match kernel writes
int lists_per_split(const kpdi_ctx *c) { return uses16(c) ? 4 : 2; }
// entries ranked per pass when keep_n needs several (bounded) passes
int pass_entries(const kpdi_ctx *c) { return c->wide32 ? 20 : kpdi::KMAX_LIMIT; }

int resolve_exact64(kpdi_ctx *c);
// first thing every entry point does.  `keep_pending`: the one caller (kpdi_push_dictionary_chunk) that starts its upload
// BEFORE it looks at the float64 certification of the previous chunk
int use_device(kpdi_ctx *c, bool keep_pending = false) {
  HIPCHK(hipSetDevice(c->device));
  if (c->pend64.active && !keep_pending) return resolve_exact64(c);
  return KPDI_OK;
}

// How a sweep of `row_blocks` x `n_tiles` tile pairs is laid on the CUs (one persistent
// workgroup per CU): `nsplit` workgroups share the dictionary tiles of a row block through
// its dynamic tile counter, and a launch covers as many row blocks as fit the chip; larger
// experimental sets take several launches.  The plan minimises the makespan counted in
// tiles: launches * ceil(n_tiles / nsplit), plus a small per-launch cost.
int choose_nsplit(const kpdi_ctx *c, int row_blocks, int n_tiles, int *rows_per_launch, int wide = -1) {
  // (the 4-wave float16 variant runs two workgroups per CU)
  const int cap = c->n_cu * kpdi::match_blocks_per_cu();
  // splits that are not a multiple of 8 leave the launch without an XCD grid (plan_xcd_grid); the kernels of match16.hip
  // (static hand-out) pay more for that than match.hip does (profiles/r03_form_choice.json)
  if (wide < 0) wide = uses16(c) ? 1 : 0;
  const double odd = wide ? c->sw.odd_wide : c->sw.odd_classic;
  int best_ns = 1, best_rpl = std::max(1, std::min(row_blocks, cap));
  double best_cost = 1e30;
  for (int ns = 1; ns <= std::min(cap, n_tiles); ++ns) {
    const int rpl = std::max(1, std::min(row_blocks, cap / ns));
    const int launches = (row_blocks + rpl - 1) / rpl;
    // multiples of 8 keep the workgroups of one XCD (block id % 8) on the same row block
    const double cost = launches * ((n_tiles + ns - 1) / ns + 0.5) * (ns % 8 == 0 ? 1.0 : odd);
    if (cost < best_cost - 1e-9) {
      best_cost = cost;
      best_ns = ns;
      best_rpl = rpl;
    }
  }
  *rows_per_launch = best_rpl;
  return best_ns;
}

// Runs the recorded background-removal steps on the resident patterns (in place).  With
// `with_prep` the metric's preparation of the patterns is fused into the same kernel when the
// detector allows it (preproc.hip); *prep_done reports whether it was.
int flush_preprocess(kpdi_ctx *c, bool with_prep, bool *prep_done) {
  *prep_done = false;
  if (!c->pend.st && !c->pend.dy) return KPDI_OK;
  kpdi::PreLaunch a{};
  a.patterns = c->exp_raw.p;
  a.dtype = c->exp_dtype;
  a.n = c->m_all;
  a.sy = c->sy;
  a.sx = c->sx;
  a.do_static = c->pend.st;
  a.bg = c->bg.as<float>();
  a.bg_min = c->pend.bg_min;
  a.bg_max = c->pend.bg_max;
  a.st_operation = c->pend.st_op;
  a.scale_bg = c->pend.st_scale;
  a.do_dynamic = c->pend.dy;
  a.taps_padded = c->taps.as<double>();
  a.ntaps = c->pend.ntaps;
  a.centre = c->pend.centre;
  a.reflect = c->pend.reflect;
  a.dy_operation = c->pend.dy_op;
  dtype_range(c->exp_dtype, &a.omin, &a.omax);
  a.do_prep = with_prep;
  if (with_prep) {
    a.out_row = c->have_nav_mask ? c->inv_map.as<int>() : nullptr;
    a.pix_map = c->have_sig_mask ? c->pix_map.as<int>() : nullptr;
    a.k = c->k_kept;
    a.kpad = c->kpad;
    a.metric = prep_metric(c);
    a.operand_form = operand_form(c);
    a.f16_step = kpdi::f16_geometry(c->f16_waves).step;
    a.out = c->exp_x.as<float>();
  }
  if (c->pend.dy && !kpdi::preprocess_fits_fused(c->sy, c->sx, 0)) {
    a.scratch_floats = kpdi::preprocess_scratch_floats(c->sy, c->sx, c->m_all, nullptr);
    HIPCHK(c->pre_scratch.reserve(a.scratch_floats * sizeof(float)));
    a.scratch = c->pre_scratch.as<float>();
  }
  {
    ScopedTimer t(c, &c->ev_pre);
    hipError_t e = kpdi::launch_preprocess(a, prep_done, c->stream);
    if (e != hipSuccess)
      return fail(KPDI_EHIP, "background-removal kernel: %s (dtype %d, %dx%d)", hipGetErrorString(e), c->exp_dtype,
                  c->sy, c->sx);
  }
  c->cnt.preproc_launches += 1;
  c->pend = kpdi_ctx::PendingPre{};
  return KPDI_OK;
}

int prepare_experimental(kpdi_ctx *c) {
  if (c->exp_prepared) return KPDI_OK;
  if (!c->have_exp) return fail(KPDI_EINVAL, "no experimental patterns set");
  if (!c->have_problem) return fail(KPDI_EINVAL, "kpdi_set_problem has not been called");
  HIPCHK(c->exp_x.reserve((size_t)c->m_pad * c->kpad * sizeof(float)));
  {
    // the preparation kernels write every column of every valid row; only the rows beyond m need zeros
    // (from the start of the 128-pattern tile m falls into: a tile's rows are interleaved)
    const size_t first = (size_t)(c->m / kpdi::TILE_DICT) * kpdi::TILE_DICT;
    if (first < (size_t)c->m_pad) {
      int rc = queue_fill(c, c->exp_x.as<float>() + first * c->kpad, ((size_t)c->m_pad - first) * c->kpad, 0u);
      if (rc) return rc;
    }
  }
  {
    int rc = flush_fills(c);  // (with whatever push_chunk_dev queued ahead: one launch)
    if (rc) return rc;
  }
  bool fused = false;
  int rc = flush_preprocess(c, true, &fused);
  if (rc) return rc;
  if (fused) {
    c->exp_prepared = true;
    return KPDI_OK;
  }
  kpdi::PrepLaunch p;
  p.raw = c->exp_raw.p;
  p.dtype = c->exp_dtype;
  p.npix = c->npix;
  p.row_map = c->have_nav_mask ? c->row_map.as<int>() : nullptr;
  p.pix_map = c->have_sig_mask ? c->pix_map.as<int>() : nullptr;
  p.quad_desc = c->have_sig_mask && c->have_quad_desc ? c->quad_desc.as<unsigned>() : nullptr;
  p.k = c->k_kept;
  p.kpad = c->kpad;
  p.n_out = c->m;
  p.metric = prep_metric(c);
  p.operand_form = operand_form(c);
  p.f16_rows = kpdi::F16_TILE;
  p.f16_step = kpdi::f16_geometry(c->f16_waves).step;
  p.out = c->exp_x.as<float>();
  {
    ScopedTimer t(c, &c->ev_prep);
    HIPCHK(kpdi::launch_prep(p, c->stream));
  }
  c->exp_prepared = true;
  return KPDI_OK;
}

int ensure_running(kpdi_ctx *c) {
  if (c->run_valid) return KPDI_OK;
  const size_t n = (size_t)c->m * c->keep_n;
  for (int j = 0; j < 2; ++j) {
    HIPCHK(c->run_s[j].reserve(std::max<size_t>(n, 1) * sizeof(float)));
    HIPCHK(c->run_i[j].reserve(std::max<size_t>(n, 1) * sizeof(int)));
  }
  c->run_cur = 0;
  c->run_empty = true;  // the first merge of the sweep takes the partial lists alone; nothing to initialise
  if (c->exact64) {
    HIPCHK(c->run64_s.reserve(std::max<size_t>(n, 1) * sizeof(double)));
    HIPCHK(c->run64_i.reserve(std::max<size_t>(n, 1) * sizeof(int)));
    HIPCHK(kpdi::launch_fill_topk64(c->run64_s.as<double>(), c->run64_i.as<int>(), (int64_t)n, c->stream));
    HIPCHK(c->cert64.reserve(2 * sizeof(unsigned)));
    HIPCHK(hipMemsetAsync(c->cert64.p, 0, 2 * sizeof(unsigned), c->stream));
  }
  c->bound_key = -1;  // a new sweep starts without a shared bound
  c->run_valid = true;
  return KPDI_OK;
}

// How the 8 XCDs tile a launch's rows x nsplit workgroups (match_device.h: block_rb_sp): among the grids
// (xr x xs = 8) the one whose XCDs stream the fewest operand bytes per tile round - (rows / xr) experimental blocks
// of 256 patterns + (nsplit / xs) dictionary tiles.  xs must divide nsplit; xr need not divide rows: the grid is then
// laid over rows rounded up to a multiple of xr (`rows_grid`), and the workgroups of the missing row blocks leave at
// once - the last launch of a large experimental set (configs[3]: 157 row blocks = 4 x 32 + 29) keeps the rectangles
// of the others instead of 29 row blocks x 1 split per XCD, 2.5 x their operand traffic (KPDI_XCD_PAD=0: only
// grids that divide rows, as before round 4).  At most an eighth more workgroups are launched for it.  Only for the
// kernels of match16.hip (`may_pad`): measured on one rank's share of configs[3] the fabric traffic of a sweep falls
// from 45.8 to 35.2 GB and the wide kernel's 40 000 x 12 500 step from 28.02 to 27.85 ms, while match.hip's step
// (dynamic draws) gets SLOWER, 28.02 -> 28.4 ms (profiles/r04_xcd_pad.txt).
// 0 x 0 = plain mapping (KPDI_XCD_GRID=0 forces it).
void plan_xcd_grid(const kpdi_ctx *c, int rows, int nsplit, int tile_dict, bool may_pad, int *xr, int *xs, int *rows_grid) {
  *xr = *xs = 0;
  *rows_grid = rows;
  if (!c->sw.xcd_grid) return;
  long best = -1;
  for (int r = 1; r <= 8; r *= 2) {
    const int sgrid = 8 / r;
    const int rows_pad = (rows + r - 1) / r * r;
    if (nsplit % sgrid != 0) continue;
    if (rows_pad != rows && (!may_pad || !c->sw.xcd_pad || 8 * (rows_pad - rows) > rows)) continue;
    const long cost = (long)(rows_pad / r) * kpdi::TILE_EXP + (long)(nsplit / sgrid) * tile_dict;
    if (best < 0 || cost < best) {
      best = cost;
      *xr = r;
      *xs = sgrid;
      *rows_grid = rows_pad;
    }
  }
}

// one match launch over the prepared chunk -> partial lists
//
// Tail: the dictionary tiles of a row block are shared by `nsplit` workgroups; when their number is a
// small non-multiple of nsplit (a rank's share of a sharded dictionary: 98 tiles over 16 workgroups)
// whole tiles would leave most workgroups idle during the last round (makespan 7 tile-times for 6.1 of
// work).  The last n_tiles % nsplit tiles are then handed out as QUARTER tiles by a second launch of the
// kernel's 32-row form, whose lists join the merge as a third source.
double wide_tail_plan(const kpdi_ctx *c, int n_tiles, int nsplit, int *shift);
// What a match launch needs initialised before it starts - the shared bound (when its plan changes), the tile counters of
// the main and the tail launch - is QUEUED here (queue_fill), so that it shares one launch with whatever else the sweep
// initialises; the plan itself is returned for run_match.
typedef kpdi_ctx::MatchPlan MatchPlan;
int match_setup(kpdi_ctx *c, int n_chunk, int n_tiles, int nsplit, int rows_per_launch, int list_len, bool bounded,
                bool allow_tail, MatchPlan *pl) {
  const int row_blocks = c->m_pad / kpdi::TILE_EXP;
  *pl = MatchPlan{};
  if (allow_tail && c->compute == KPDI_COMPUTE_F32 && !c->wide32 && !bounded && row_blocks <= rows_per_launch &&
      !c->sw.no_tail) {
    const int rounds = n_tiles / nsplit, rem = n_tiles % nsplit;
    if (rounds >= 1 && rounds < 32 && rem > 0 && 4 * rem <= 3 * nsplit) pl->tail_tiles = rem;
  }
  pl->n_main = n_tiles - pl->tail_tiles;
  {
    // the published ranks are only comparable under one plan: (re)initialise when it changes
    int used;
    kpdi::bound_plan(lists_per_split(c) * nsplit, list_len, &pl->bound_rank, &pl->bound_grouped, &used);
    const int key = (pl->bound_rank << 8) | (pl->bound_grouped << 7) | used;
    if (key != c->bound_key || bounded) {
      HIPCHK(c->gthr.reserve((size_t)c->m_pad * kpdi::BOUND_SLOTS * sizeof(unsigned)));
      int rc = queue_fill(c, c->gthr.p, (size_t)c->m_pad * kpdi::BOUND_SLOTS, 0u, used);
      if (rc) return rc;
      c->bound_key = bounded ? -1 : key;  // bounded passes always start from scratch
    }
  }
  // Tile hand-out (match.hip): a workgroup's first `fixed_draws` tiles are fixed (sp, sp + nsplit, ...) so that
  // the workgroups sharing an XCD stream the same operands at the same pace (the XCD's L2 then serves them:
  // 26 -> ~12 GB crossing the fabric per config-2 launch); the last ~20 % are drawn from the row block's counter,
  // which evens out the speeds at the end (all tiles fixed left CUs idle for the last ~10 % of the launch).
  // KPDI_FIXED_FRAC overrides the fixed share.
  {
    const double frac = c->sw.fixed_frac;
    const int per_wg = pl->n_main / nsplit;
    pl->fixed_draws = (pl->tail_tiles > 0 || pl->n_main % nsplit == 0) && frac > 0 ? per_wg + 1 : std::max(3, (int)(frac * per_wg));
  }
  const size_t ctr_words = (size_t)row_blocks;
  HIPCHK(c->tile_ctr.reserve(2 * ctr_words * sizeof(unsigned)));  // second half: the tail launch
  int rc = queue_fill(c, c->tile_ctr.p, ctr_words, (unsigned)pl->fixed_draws * (unsigned)nsplit);
  if (rc) return rc;
  if (pl->tail_tiles > 0) {
    pl->tail_units = (std::min(n_chunk, n_tiles * kpdi::TILE_DICT) - pl->n_main * kpdi::TILE_DICT + 31) / 32;
    pl->tail_nsplit = std::min(nsplit, pl->tail_units);
    rc = queue_fill(c, c->tile_ctr.as<unsigned>() + ctr_words, ctr_words, 3u * (unsigned)pl->tail_nsplit);
    if (rc) return rc;
  }
  return KPDI_OK;
}

int run_match(kpdi_ctx *c, const float *dict_y, int n_chunk, int n_tiles, int nsplit, int rows_per_launch,
              int list_len, int64_t global_start, const float *bound_s, const int *bound_i, bool allow_tail = false) {
  c->tail_nsplit = 0;
  MatchPlan pl;
  const kpdi_ctx::MatchSetup &ps = c->presetup;
  if (ps.valid && ps.n_chunk == n_chunk && ps.n_tiles == n_tiles && ps.nsplit == nsplit && ps.rows_per_launch == rows_per_launch &&
      ps.list_len == list_len && bound_s == nullptr && allow_tail) {
    pl = c->preplan;  // queued (and flushed with the preparation's own initialisations) by push_chunk_dev
  } else {
    int rc = match_setup(c, n_chunk, n_tiles, nsplit, rows_per_launch, list_len, bound_s != nullptr, allow_tail, &pl);
    if (rc) return rc;
  }
  c->presetup.valid = false;
  {
    int rc = flush_fills(c);
    if (rc) return rc;
  }
  const int tail_tiles = pl.tail_tiles, n_main = pl.n_main;
  const bool f16 = uses16(c);
  const int lists_per_split = ::lists_per_split(c);
  const size_t part = (size_t)c->m_pad * lists_per_split * nsplit * list_len;
  HIPCHK(c->part_s.reserve(part * sizeof(float)));
  HIPCHK(c->part_i.reserve(part * sizeof(int)));
  kpdi::MatchLaunch ml;
  ml.dict = dict_y;
  ml.exp = c->exp_x.as<float>();
  ml.kpad = c->kpad;
  ml.n_tiles = n_main;
  ml.n_valid = n_chunk;
  ml.m_pad = c->m_pad;
  ml.nsplit = nsplit;
  ml.idx_base = (int)global_start;
  ml.list_len = list_len;
  ml.part_scores = c->part_s.as<float>();
  ml.part_idx = c->part_i.as<int>();
  ml.bound_score = bound_s;
  ml.bound_idx = bound_i;
  ml.operand_form = operand_form(c);
  if (c->wide32) {  // (float32 form only: the same guards in the float16 schedule cost its 32-cycle MFMAs 10 %)
    (void)wide_tail_plan(c, n_main, nsplit, &ml.tail_shift);
    ml.tail_first = n_main - n_main % nsplit;
  }
  ml.bound_rank = pl.bound_rank;
  ml.bound_grouped = pl.bound_grouped;
  ml.gthr = c->gthr.as<unsigned>();
  ml.tile_groups = 1;
  ml.fixed_draws = pl.fixed_draws;
  const size_t ctr_bytes = (size_t)(c->m_pad / kpdi::TILE_EXP) * sizeof(unsigned);
  ml.tile_ctr = c->tile_ctr.as<unsigned>();
  const int row_blocks = c->m_pad / kpdi::TILE_EXP;
  int launched_rows = 0;
  {
    ScopedTimer t(c, &c->ev_match);  // one timed region = the whole sweep of this chunk
    // several launches (large experimental sets) alternate between two streams: the workgroups
    // of launch j+1 start on the CUs that launch j's tail leaves idle
    const bool two = row_blocks > rows_per_launch && !c->sw.one_stream;
    const bool tail2 = tail_tiles > 0 && c->sw.tail_stream2;  // the tail launch runs on the second stream
    if (two || tail2) {
      if (!c->stream2) {
        HIPCHK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
      }
      HIPCHK(hipEventRecord(c->ev_fork, c->stream));
      HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    }
    int j = 0;
    // (a launch's grid may be padded, plan_xcd_grid: the largest of the sweep's launches - the full ones and the last one)
    int grid_rows = 0;
    const int full_rows = std::min(rows_per_launch, row_blocks);
    for (int rows : {full_rows, row_blocks % rows_per_launch ? row_blocks % rows_per_launch : full_rows}) {
      int xr, xs, rg;
      plan_xcd_grid(c, rows, nsplit, dict_tile(c), f16, &xr, &xs, &rg);
      grid_rows = std::max(grid_rows, rg);
    }
    const size_t scratch16 = f16 ? kpdi::match16_scratch_bytes(grid_rows * nsplit, c->f16_waves, list_len) : 0;
    launched_rows = grid_rows;
    if (f16) HIPCHK(c->list16.reserve((two ? 2 : 1) * scratch16));
    for (int r0 = 0; r0 < row_blocks; r0 += rows_per_launch, ++j) {
      ml.row_first = r0;
      ml.rows = std::min(rows_per_launch, row_blocks - r0);
      plan_xcd_grid(c, ml.rows, nsplit, dict_tile(c), f16, &ml.xcd_rows, &ml.xcd_splits, &ml.rows_grid);
      hipStream_t st = (two && (j & 1)) ? c->stream2 : c->stream;
      if (f16) {
        // launches on the two streams overlap: each stream has its own list scratch
        char *scratch = (char *)c->list16.p + ((two && (j & 1)) ? scratch16 : 0);
        HIPCHK(kpdi::launch_match16(ml, c->f16_waves, scratch, st));
      } else {
        HIPCHK(kpdi::launch_match(ml, st));
      }
    }
    if (two) {
      HIPCHK(hipEventRecord(c->ev_join, c->stream2));
      HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    }
    if (tail_tiles > 0) {
      // 32-row units over the rows [n_main * 128, n_chunk): the same shared bound (a slot then holds the
      // larger of a main list's and a tail list's published entry - still backed by that many candidates)
      const int units = pl.tail_units;
      const int ns_t = pl.tail_nsplit;
      const size_t part_t = (size_t)c->m_pad * 2 * ns_t * list_len;
      HIPCHK(c->tail_s.reserve(part_t * sizeof(float)));
      HIPCHK(c->tail_i.reserve(part_t * sizeof(int)));
      kpdi::MatchLaunch tl = ml;
      tl.row_tiles = 1;
      tl.row_base = n_main * kpdi::TILE_DICT;
      tl.n_tiles = units;
      tl.nsplit = ns_t;
      tl.tile_groups = 1;
      tl.fixed_draws = 3;
      tl.xcd_rows = tl.xcd_splits = tl.rows_grid = 0;
      tl.part_scores = c->tail_s.as<float>();
      tl.part_idx = c->tail_i.as<int>();
      tl.tile_ctr = c->tile_ctr.as<unsigned>() + ctr_bytes / sizeof(unsigned);  // (initialised with the main launch's)
      tl.row_first = 0;
      tl.rows = row_blocks;
      // The tail launch does not depend on the main launch (lists of its own, counters of its own, the shared bound is a
      // filter that is valid however stale).  On the second stream (KPDI_TAIL_STREAM2=1) it is dispatched beside the main
      // launch - measured (round 3, rocprofv3 trace of one rank's share at N = 8): no gain, the main launch's persistent
      // workgroups hold every CU until they all finish within microseconds of each other, and the join event costs 10 us -
      // so it stays behind the main launch on the same stream.
      if (!tail2) {
        HIPCHK(kpdi::launch_match(tl, c->stream));
      } else {
        HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));  // recorded in front of the main launch (below)
        HIPCHK(kpdi::launch_match(tl, c->stream2));
        HIPCHK(hipEventRecord(c->ev_join, c->stream2));
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
      }
      c->tail_nsplit = ns_t;
    }
  }
  c->cnt.match_launches += 1;
  c->cnt.match_form = operand_form(c);
  c->cnt.match_flops += 2.0 * (double)c->m * (double)n_chunk * (double)c->k_kept;
  c->cnt.match_grid = launched_rows * nsplit;  // workgroups of the sweep's largest launch (its padding included)
  c->cnt.match_nsplit = nsplit;
  return KPDI_OK;
}

// raw chunk (device) -> prepared layout at `out` (n_pad rows of kpad floats, tiles of 128 patterns);
// `out` may point into a larger buffer at a tile boundary
int prepare_chunk(kpdi_ctx *c, const void *d_patterns, int dtype, int64_t n_chunk, float *out) {
  const int tile = dict_tile(c);
  const int n_pad = kpdi::round_up(n_chunk, tile);
  const int n_tiles = n_pad / tile;
  if (n_pad > n_chunk && c->tail_queued != out) {  // rows of the last tile are interleaved: clear the whole tile
    int rc = queue_fill(c, out + (size_t)(n_tiles - 1) * tile * c->kpad, (size_t)tile * c->kpad, 0u);
    if (rc) return rc;
  }
  c->tail_queued = nullptr;
  {
    int rc = flush_fills(c);
    if (rc) return rc;
  }
  kpdi::PrepLaunch p;
  p.raw = d_patterns;
  p.dtype = dtype;
  p.npix = c->npix;
  p.row_map = nullptr;
  p.pix_map = c->have_sig_mask ? c->pix_map.as<int>() : nullptr;
  p.quad_desc = c->have_sig_mask && c->have_quad_desc ? c->quad_desc.as<unsigned>() : nullptr;
  p.k = c->k_kept;
  p.kpad = c->kpad;
  p.n_out = (int)n_chunk;
  p.metric = prep_metric(c);
  p.operand_form = operand_form(c);
  p.f16_rows = kpdi::f16_geometry(c->f16_waves).dict_tile;
  p.f16_step = kpdi::f16_geometry(c->f16_waves).step;
  p.out = out;
  {
    ScopedTimer t(c, &c->ev_prep);
    HIPCHK(kpdi::launch_prep(p, c->stream));
  }
  return KPDI_OK;
}

// Which f32 match kernel serves this sweep.  match16.hip's one-wave-per-SIMD form (256 x 256 tiles, lists out of the
// registers) does a unit of work 3 % faster than match.hip (128 x 256 tiles) but hands out whole 256-pattern tiles
// statically, match.hip 128-pattern tiles with a dynamic tail of quarter tiles: the estimated makespans decide.  The two
// kernels read different operand layouts (and row paddings), so the choice is made when the first chunk of a sweep
// arrives - nothing prepared yet, no resident chunks - and stands until then again.  KPDI_F32_WIDE = 1 / 0 forces it.
double wide_tail_plan(const kpdi_ctx *c, int n_tiles, int nsplit, int *shift);
void decide_form(kpdi_ctx *c, int64_t n_chunk) {
  if (c->compute != KPDI_COMPUTE_F32 || c->wide_mode >= 0) return;
  if (c->exp_prepared || !c->held.empty()) return;
  const int row_blocks = c->have_exp ? c->m_pad / kpdi::TILE_EXP : 16;
  int rpl = 0;
  // match.hip: whole rounds of 128-pattern tiles + (when few rounds) a quarter-tile tail launch
  const int t128 = (int)((n_chunk + 127) / 128);
  const int ns = choose_nsplit(c, row_blocks, t128, &rpl, 0);
  const int launches = (row_blocks + rpl - 1) / rpl;
  double classic = (t128 + ns - 1) / ns;
  if (launches == 1) {
    const int rounds = t128 / ns, rem = t128 % ns;
    if (rounds >= 1 && rounds < 32 && rem > 0 && 4 * rem <= 3 * ns)
      classic = rounds + ((4 * rem + ns - 1) / ns) * 0.25 + kpdi::FORM_CLASSIC_TAIL;
  }
  classic = launches * (classic + kpdi::FORM_CLASSIC_LAUNCH);  // + ~0.1 ms per launch
  // match16.hip, float32 form: whole rounds of 256-pattern tiles, two 128-tile units each at 1 / 1.03 of the time
  const int t256 = (int)((n_chunk + 255) / 256);
  const int nsw = choose_nsplit(c, row_blocks, t256, &rpl, 1);
  int shift = 0;
  // (its launch costs more: the first tile's 64 candidates per lane go to the buffers, the lists are built at the end -
  // 0.27 ms against 0.1 ms, measured on one rank's share of configs[1] at N = 8)
  // (fitted between K = 2819 and 14 400: no extrapolation below)
  const double gain = kpdi::FORM_WIDE_GAIN + kpdi::FORM_WIDE_GAIN_K * std::max(-0.3, 1.0 - 3600.0 / std::max(c->k_kept, 1));
  const double wide_launch = c->sw.wide_launch;  // (KPDI_FORM_WIDE_LAUNCH: fitting runs)
  const double wide = ((row_blocks + rpl - 1) / rpl) *
                      ((t256 / nsw + wide_tail_plan(c, t256, nsw, &shift)) * 2.0 / gain + wide_launch) *
                      (nsw % 8 == 0 ? 1.0 : kpdi::FORM_WIDE_ODD);
  const bool w = wide < classic;
  if (w == c->wide32) return;
  c->wide32 = w;
  c->kpad = kpdi::round_up(c->k_kept + (c->metric == KPDI_METRIC_NDP ? 1 : 0), w ? kpdi::F16_STEP / 2 : kpdi::TILE_K);
  c->cnt.kpad = c->kpad;
}

// match16.hip, float32 form: how the last n_tiles % nsplit tiles of a launch are handed out - as whole tiles (one more
// round, shift 0) or as halves / quarters of a tile (a half / a quarter of a round each, at ~1.1 / 1.25 of the time per
// row because the experimental fragments are reused by fewer row groups).  Returns the cost of that last round in
// tile-times.
double wide_tail_plan(const kpdi_ctx *c, int n_tiles, int nsplit, int *shift) {
  *shift = 0;
  const int left = n_tiles % nsplit;
  if (left == 0) return 0.0;
  double best = 1.0;
  if (!c->sw.no_tail)
    for (int sh = 1; sh <= 2; ++sh) {
      const double cost = (double)(((left << sh) + nsplit - 1) / nsplit) / (1 << sh) *
                          (sh == 1 ? kpdi::FORM_WIDE_HALF : kpdi::FORM_WIDE_QUARTER);
      if (cost < best - 1e-9) best = cost, *shift = sh;
    }
  return best;
}

int check_chunk_args(kpdi_ctx *c, int dtype, int64_t n_chunk, int64_t global_start) {
  if (!c->have_problem) return fail(KPDI_EINVAL, "kpdi_set_problem has not been called");
  if (n_chunk <= 0) return fail(KPDI_EINVAL, "dictionary chunk must hold at least one pattern");
  if (kpdi::dtype_size(dtype) == 0) return fail(KPDI_EINVAL, "unknown dtype %d", dtype);
  if (global_start < 0 || global_start + n_chunk >= (int64_t)INT_MAX)
    return fail(KPDI_EINVAL, "dictionary indices must fit in int32");
  return KPDI_OK;
}

int sweep_prepared(kpdi_ctx *c, const float *y, int64_t n_chunk, int64_t global_start, const void *raw = nullptr,
                   int raw_dtype = 0);
void release_held(kpdi_ctx *c);

int push_chunk_dev(kpdi_ctx *c, const void *d_patterns, int dtype, int64_t n_chunk, int64_t global_start) {
  int rc = resolve_exact64(c);  // (before this chunk's preparation overwrites what extra passes of the last one would read)
  if (rc) return rc;
  rc = check_chunk_args(c, dtype, n_chunk, global_start);
  if (rc) return rc;
  if (!c->have_exp) return fail(KPDI_EINVAL, "kpdi_set_experimental has not been called");
  if (c->m == 0) return KPDI_OK;
  decide_form(c, n_chunk);
  const int tile = dict_tile(c);
  const int n_pad = kpdi::round_up(n_chunk, tile);
  HIPCHK(c->dict_y.reserve((size_t)n_pad * c->kpad * sizeof(float)));
  // Everything the step initialises - the shared bound and the tile counters of the match launch, the zero rows behind a
  // partial last tile of this chunk and of the experimental matrix - is queued BEFORE the first preparation kernel and
  // goes out as ONE launch (one rank's share of a sharded job is a 3 ms step: five small launches were 2 % of it).
  if (!c->exact64 && c->keep_n <= kpdi::KMAX_LIMIT) {
    rc = ensure_running(c);
    if (rc) return rc;
    kpdi_ctx::MatchSetup &ps = c->presetup;
    ps.n_chunk = (int)n_chunk;
    ps.n_tiles = n_pad / tile;
    const int row_blocks = c->m_pad / kpdi::TILE_EXP;
    ps.rows_per_launch = row_blocks;
    ps.nsplit = choose_nsplit(c, row_blocks, ps.n_tiles, &ps.rows_per_launch);
    ps.list_len = kpdi::match_list_len(c->keep_n);
    rc = match_setup(c, ps.n_chunk, ps.n_tiles, ps.nsplit, ps.rows_per_launch, ps.list_len, false, true, &c->preplan);
    if (rc) return rc;
    ps.valid = true;
    if (n_pad > n_chunk) {
      float *out = c->dict_y.as<float>();
      rc = queue_fill(c, out + (size_t)(ps.n_tiles - 1) * tile * c->kpad, (size_t)tile * c->kpad, 0u);
      if (rc) return rc;
      c->tail_queued = out;
    }
  }
  rc = prepare_experimental(c);  // (flushes the queue in front of its kernel)
  if (rc) return rc;
  rc = prepare_chunk(c, d_patterns, dtype, n_chunk, c->dict_y.as<float>());
  if (rc) return rc;
  return sweep_prepared(c, c->dict_y.as<float>(), n_chunk, global_start, d_patterns, dtype);
}

// One screening pass over a prepared chunk: the ranks [done, done + kp) of every pattern WITHIN this chunk ->
// columns done .. of loc_s / loc_i (row stride `stride`); pass p only admits candidates ranked strictly
// after the last entry of pass p-1 (bound_s / bound_i = the last column so far)
int local_pass(kpdi_ctx *c, const float *y, int n_chunk, int n_tiles, int nsplit, int rows_per_launch,
               int64_t global_start, int done, int kp, int stride) {
  const int len = kpdi::match_list_len(kp);
  c->bound_key = -1;  // each pass ranks a different slice: its shared bound starts from scratch
  int rc = run_match(c, y, n_chunk, n_tiles, nsplit, rows_per_launch, len, global_start,
                     done ? c->bound_s.as<float>() : nullptr, done ? c->bound_i.as<int>() : nullptr);
  if (rc) return rc;
  kpdi::MergeLaunch pm{};
  pm.m = c->m;
  pm.k = kp;
  pm.n_src = 1;
  pm.src_scores[0] = c->part_s.as<float>();
  pm.src_idx[0] = c->part_i.as<int>();
  const int lps = lists_per_split(c);
  pm.src_lists[0] = lps * nsplit;
  pm.src_len[0] = len;
  pm.src_row_stride[0] = lps * nsplit * len;
  pm.src_list_stride[0] = len;
  pm.out_scores = c->loc_s.as<float>();
  pm.out_idx = c->loc_i.as<int>();
  pm.out_stride = stride;
  pm.out_offset = done;
  {
    ScopedTimer t(c, &c->ev_merge);
    HIPCHK(kpdi::launch_merge(pm, c->stream));
  }
  HIPCHK(kpdi::launch_last_column(c->loc_s.as<float>(), c->loc_i.as<int>(), c->m, stride, done + kp - 1,
                                  c->bound_s.as<float>(), c->bound_i.as<int>(), c->stream));
  return KPDI_OK;
}

// float64 arithmetic (rescore.hip): screen keep_n + 12 candidates of the chunk in f32, rescore them in double from
// the raw patterns, merge into the running float64 best-k, certify; uncertified patterns get up to EXTRA64 more
// screening passes of 32 candidates
constexpr int MARGIN64 = 12, EXTRA64 = 3;

// screening passes of the pending chunk up to `target` candidates per pattern (each: f32 match + rescoring in double +
// merge into the running float64 best-k with its certification), then the read-back of the last merge's verdict
int exact64_passes(kpdi_ctx *c, int64_t target) {
  kpdi_ctx::Pending64 &q = c->pend64;
  const int k = c->keep_n;
  const int pass = pass_entries(c);
  unsigned *cert = c->cert64.as<unsigned>();
  while (q.done < target) {
    const int kp = (int)std::min<int64_t>(q.done == 0 ? kpdi::KMAX_LIMIT : pass, target - q.done);
    int rc = local_pass(c, q.y, (int)q.n_chunk, q.n_tiles, q.nsplit, q.rows_per_launch, q.global_start, q.done, kp, q.cap);
    if (rc) return rc;
    ScopedTimer t(c, &c->ev_rescore);
    kpdi::RescoreLaunch r{};
    r.exp_raw = c->exp_raw.p;
    r.exp_dtype = c->exp_dtype;
    r.row_map = c->have_nav_mask ? c->row_map.as<int>() : nullptr;
    r.dict_raw = q.raw;
    r.dict_dtype = q.raw_dtype;
    r.n_chunk = q.n_chunk;
    r.global_start = q.global_start;
    r.pix_map = c->have_sig_mask ? c->pix_map.as<int>() : nullptr;
    r.k = c->k_kept;
    r.npix = c->npix;
    r.metric = c->metric;
    r.m = c->m;
    r.cand_s = c->loc_s.as<float>();
    r.cand_i = c->loc_i.as<int>();
    r.cand_stride = q.cap;
    r.cand_offset = q.done;
    r.n_cand = kp;
    r.cand_s64 = c->cand64.as<double>();
    r.max_diff = cert;
    HIPCHK(kpdi::launch_rescore(r, c->stream));
    HIPCHK(hipMemsetAsync(cert + 1, 0, sizeof(unsigned), c->stream));
    kpdi::Merge64Launch g{};
    g.m = c->m;
    g.k = k;
    g.run_s = c->run64_s.as<double>();
    g.run_i = c->run64_i.as<int>();
    g.cand_s64 = c->cand64.as<double>() + q.done;
    g.cand_i = c->loc_i.as<int>() + q.done;
    g.lists = 1;
    g.len = kp;
    g.row_stride = q.cap;
    g.list_stride = 0;
    g.out_s = c->run64_s.as<double>();
    g.out_i = c->run64_i.as<int>();
    g.cand_s32 = c->loc_s.as<float>();
    g.s32_stride = q.cap;
    g.s32_col = q.done + kp - 1;
    g.enumerated_all = q.done + kp >= q.n_chunk;
    g.max_diff = cert;
    // what an unscreened candidate's float64 score may exceed its float32 score by: 8 x the largest difference seen
    // among the rescored pairs of the sweep (a STATISTICAL bound: ~130 000 samples per chunk at configs[1], taken
    // from the best-scoring pairs, whose partial sums - and rounding errors - are the largest), never less than
    // 1e-6; KPDI_F64_EPS=worstcase raises the floor to the worst-case accumulation bound of a K-term float32 dot
    // product of unit vectors, (K + 2) 2^-24 (2.1e-4 at K = 3600): a certificate that holds for any data, at the price
    // of more screening passes where the k-th and the screened-last scores are closer than that
    g.eps_floor = c->sw.f64_worstcase ? (float)((c->k_kept + 2) * 0x1p-24 * 1.01) + 1e-6f : 1e-6f;
    g.uncertified = (int *)(cert + 1);
    HIPCHK(kpdi::launch_merge64(g, c->stream));
    q.done += kp;
  }
  HIPCHK(hipMemcpyAsync(q.flag.p, cert + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipEventRecord(q.ready, c->stream));
  return KPDI_OK;
}

// Look at the pending chunk's verdict; patterns it left uncertified get up to EXTRA64 more screening passes of the SAME
// chunk - whose prepared form, raw patterns and candidate buffers are still in place: every entry point comes through
// here (use_device) before it touches them.
int resolve_exact64(kpdi_ctx *c) {
  kpdi_ctx::Pending64 &q = c->pend64;
  if (!q.active) return KPDI_OK;
  q.active = false;  // (an error below leaves no half-resolved chunk behind)
  bool more = false;
  int uncertified = 0;
  for (;;) {
    HIPCHK(hipEventSynchronize(q.ready));
    uncertified = *(volatile int *)q.flag.p;
    if (uncertified == 0 || q.done >= q.n_chunk || q.extra == EXTRA64) break;
    ++q.extra;
    c->cnt.rescore_extra_passes += 1;
    more = true;
    int rc = exact64_passes(c, std::min<int64_t>((int64_t)q.done + pass_entries(c), q.n_chunk));
    if (rc) return rc;
  }
  c->cnt.uncertified_patterns += uncertified;
  // the extra passes read the chunk's staging buffer after staged_upload released it: release both again, behind them
  if (more && c->copy_stream)
    for (int b = 0; b < 2; ++b) HIPCHK(hipEventRecord(c->stage_free[b], c->stream));
  return KPDI_OK;
}

// float64 arithmetic (rescore.hip): screen keep_n + 12 candidates of the chunk in f32, rescore them in double from
// the raw patterns, merge into the running float64 best-k, certify; uncertified patterns get up to EXTRA64 more
// screening passes of 32 candidates.  The verdict of the first passes is read back asynchronously: a caller streaming
// host chunks (kpdi_push_dictionary_chunk) looks at it only after the NEXT chunk's upload has been queued, so that
// upload and sweep overlap as they do in the float32 modes; everyone else resolves it before returning.
int sweep_exact64(kpdi_ctx *c, const float *y, int64_t n_chunk, int64_t global_start, const void *raw, int raw_dtype,
                  int n_tiles, int nsplit, int rows_per_launch) {
  const int k = c->keep_n;
  if ((int64_t)k + MARGIN64 > 4096) return fail(KPDI_EINVAL, "float64 arithmetic supports keep_n <= %d", 4096 - MARGIN64);
  const int pass = pass_entries(c);
  const int cap = kpdi::round_up(k + MARGIN64, pass) + pass * EXTRA64;
  const size_t n = (size_t)c->m * cap;
  HIPCHK(c->loc_s.reserve(n * sizeof(float)));
  HIPCHK(c->loc_i.reserve(n * sizeof(int)));
  HIPCHK(c->cand64.reserve(n * sizeof(double)));
  HIPCHK(c->bound_s.reserve((size_t)c->m_pad * sizeof(float)));
  HIPCHK(c->bound_i.reserve((size_t)c->m_pad * sizeof(int)));
  HIPCHK(kpdi::launch_fill_topk(c->bound_s.as<float>(), c->bound_i.as<int>(), c->m_pad, c->stream));
  HIPCHK(kpdi::launch_fill_topk(c->loc_s.as<float>(), c->loc_i.as<int>(), (int64_t)n, c->stream));
  kpdi_ctx::Pending64 &q = c->pend64;
  if (!q.ready) HIPCHK(hipEventCreateWithFlags(&q.ready, hipEventDisableTiming));
  HIPCHK(q.flag.reserve(sizeof(int)));
  q.y = y;
  q.raw = raw;
  q.raw_dtype = raw_dtype;
  q.n_tiles = n_tiles;
  q.nsplit = nsplit;
  q.rows_per_launch = rows_per_launch;
  q.cap = cap;
  q.done = 0;
  q.extra = 0;
  q.n_chunk = n_chunk;
  q.global_start = global_start;
  int rc = exact64_passes(c, std::min<int64_t>((int64_t)k + MARGIN64, n_chunk));
  if (rc) return rc;
  q.active = true;
  return q.defer && !c->sw.f64_sync ? KPDI_OK : resolve_exact64(c);  // (KPDI_F64_SYNC: round 2's behaviour, A/B)
}

// every experimental pattern against one prepared chunk, merged into the running best-k
int sweep_prepared(kpdi_ctx *c, const float *y, int64_t n_chunk, int64_t global_start, const void *raw, int raw_dtype) {
  int rc = prepare_experimental(c);
  if (rc) return rc;
  rc = ensure_running(c);
  if (rc) return rc;
  const int n_tiles = kpdi::round_up(n_chunk, dict_tile(c)) / dict_tile(c);

  const int row_blocks = c->m_pad / kpdi::TILE_EXP;
  int rows_per_launch = row_blocks;
  const int nsplit = choose_nsplit(c, row_blocks, n_tiles, &rows_per_launch);
  const int k = c->keep_n;
  if (c->exact64) {
    if (!raw)
      return fail(KPDI_EINVAL, "float64 arithmetic rescoring reads the RAW dictionary patterns: resident (held) chunks keep "
                               "only the prepared form - push the chunks instead");
    c->final_valid = false;
    return sweep_exact64(c, y, n_chunk, global_start, raw, raw_dtype, n_tiles, nsplit, rows_per_launch);
  }
  const int cur = c->run_cur, nxt = cur ^ 1;
  c->final_valid = false;

  kpdi::MergeLaunch mg{};
  mg.m = c->m;
  mg.out_scores = c->run_s[nxt].as<float>();
  mg.out_idx = c->run_i[nxt].as<int>();
  mg.out_stride = k;
  mg.out_offset = 0;
  mg.k = k;
  // first source: the running best-k - unless this is the first chunk of the sweep (nothing to merge with, and
  // nothing was initialised: ensure_running)
  int ns = 0;
  if (!c->run_empty) {
    mg.src_scores[ns] = c->run_s[cur].as<float>();
    mg.src_idx[ns] = c->run_i[cur].as<int>();
    mg.src_lists[ns] = 1;
    mg.src_len[ns] = k;
    mg.src_row_stride[ns] = k;
    mg.src_list_stride[ns] = k;
    ++ns;
  }

  if (k <= kpdi::KMAX_LIMIT) {
    const int len = kpdi::match_list_len(k);
    rc = run_match(c, y, (int)n_chunk, n_tiles, nsplit, rows_per_launch, len, global_start, nullptr, nullptr, true);
    if (rc) return rc;
    mg.src_scores[ns] = c->part_s.as<float>();
    mg.src_idx[ns] = c->part_i.as<int>();
    const int lps = lists_per_split(c);
    mg.src_lists[ns] = lps * nsplit;
    mg.src_len[ns] = len;
    mg.src_row_stride[ns] = lps * nsplit * len;
    mg.src_list_stride[ns] = len;
    ++ns;
    if (c->tail_nsplit > 0) {
      mg.src_scores[ns] = c->tail_s.as<float>();
      mg.src_idx[ns] = c->tail_i.as<int>();
      mg.src_lists[ns] = 2 * c->tail_nsplit;
      mg.src_len[ns] = len;
      mg.src_row_stride[ns] = 2 * c->tail_nsplit * len;
      mg.src_list_stride[ns] = len;
      ++ns;
    }
    mg.n_src = ns;
  } else {
    // keep_n > 32: passes of 32 ranks; pass p only admits candidates ranked
    // strictly after the last entry of pass p-1
    const size_t n = (size_t)c->m * k;
    HIPCHK(c->loc_s.reserve(n * sizeof(float)));
    HIPCHK(c->loc_i.reserve(n * sizeof(int)));
    HIPCHK(c->bound_s.reserve((size_t)c->m_pad * sizeof(float)));
    HIPCHK(c->bound_i.reserve((size_t)c->m_pad * sizeof(int)));
    HIPCHK(kpdi::launch_fill_topk(c->bound_s.as<float>(), c->bound_i.as<int>(), c->m_pad, c->stream));
    const int kk = (int)std::min<int64_t>(k, n_chunk);
    if (kk < k) HIPCHK(kpdi::launch_fill_topk(c->loc_s.as<float>(), c->loc_i.as<int>(), (int64_t)n, c->stream));
    for (int done = 0; done < kk;) {  // (the first pass is unbounded: up to 32 entries in every form)
      const int kp = std::min(done == 0 ? kpdi::KMAX_LIMIT : pass_entries(c), kk - done);
      rc = local_pass(c, y, (int)n_chunk, n_tiles, nsplit, rows_per_launch, global_start, done, kp, k);
      if (rc) return rc;
      done += kp;
    }
    mg.src_scores[ns] = c->loc_s.as<float>();
    mg.src_idx[ns] = c->loc_i.as<int>();
    mg.src_lists[ns] = 1;
    mg.src_len[ns] = k;
    mg.src_row_stride[ns] = k;
    mg.src_list_stride[ns] = k;
    mg.n_src = ns + 1;
  }
  rc = wait_result_copy(c);
  if (rc) return rc;
  {
    ScopedTimer t(c, &c->ev_merge);
    HIPCHK(kpdi::launch_merge(mg, c->stream));
  }
  c->run_cur = nxt;
  c->run_empty = false;
  return KPDI_OK;
}

int set_experimental_common(kpdi_ctx *c, const void *src, bool src_on_device, int dtype, int64_t m_all,
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
  c->final_valid = false;
  return KPDI_OK;
}

}  // namespace

namespace {

// Host chunk -> device in pieces of `per` patterns through two staging buffers on the copy
// stream; `consume(d_piece, n, offset)` queues the work that reads a piece on the compute
// stream.  The upload of piece j+1 overlaps whatever `consume` queued for piece j - pieces
// of this call or of the previous call (a caller streaming chunk after chunk, like the
// reference's loop).  On return the host buffer has been consumed.
template <typename F>
int staged_upload(kpdi_ctx *c, const void *patterns, size_t row_bytes, const std::vector<int64_t> &pieces, F consume) {
  const int64_t per = *std::max_element(pieces.begin(), pieces.end());
  if (!c->copy_stream) {
    HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
      HIPCHK(hipEventCreateWithFlags(&c->stage_filled[b], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&c->stage_free[b], hipEventDisableTiming));
      HIPCHK(hipEventRecord(c->stage_free[b], c->stream));
    }
  }
  for (int b = 0; b < 2; ++b)
    if (c->stage[b].cap < (size_t)per * row_bytes) {
      // growing a buffer frees it: everything queued on it must have finished (and a float64 chunk whose certification
      // has not been looked at yet may still want to read it)
      int rc = resolve_exact64(c);
      if (rc) return rc;
      HIPCHK(hipStreamSynchronize(c->copy_stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      HIPCHK(c->stage[b].reserve((size_t)per * row_bytes));
    }
  int64_t start = 0;
  for (const int64_t n : pieces) {
    const int b = c->stage_next;
    c->stage_next ^= 1;
    HIPCHK(hipStreamWaitEvent(c->copy_stream, c->stage_free[b], 0));
    HIPCHK(hipMemcpyAsync(c->stage[b].p, (const char *)patterns + (size_t)start * row_bytes, (size_t)n * row_bytes,
                          hipMemcpyHostToDevice, c->copy_stream));
    HIPCHK(hipEventRecord(c->stage_filled[b], c->copy_stream));
    c->cnt.h2d_bytes += (double)n * row_bytes;
    HIPCHK(hipStreamWaitEvent(c->stream, c->stage_filled[b], 0));
    int rc = consume(c->stage[b].p, n, start);
    if (rc) return rc;
    HIPCHK(hipEventRecord(c->stage_free[b], c->stream));  // the prep kernel has consumed the piece
    start += n;
  }
  HIPCHK(hipStreamSynchronize(c->copy_stream));  // the caller's buffer is free again
  return KPDI_OK;
}

// How a host chunk is cut for the upload/sweep pipeline: uniform pieces (the remainder last),
// each one match launch set.  Short pieces bound the two ends that do not overlap (the upload
// of the first piece, the sweep of the last); long pieces waste less on whole tiles per
// workgroup and on the per-piece prep/merge.  Which wins depends on whether the job is
// upload-bound (few experimental patterns) or sweep-bound (many), so the size is picked by
// playing the two-stage pipeline through for every candidate with the launch plan's own cost
// model: upload at ~56 GB/s (pageable memory over PCIe 5, measured), one workgroup-tile
// (256 x 128 x kpad MACs) at 88 % of a CU's f32 MFMA rate, 0.18 ms per piece of fixed work
// (launch ramp, prep, merge; fitted).  The model lands within ~0.3 ms of the measurements at
// config 2 (tools/pcie_probe.py; upload alone 25.5 ms): 32-tile pieces 27.2 ms, 48 tiles 27.5 ms
// (about the model's pick), 96 tiles 28.7 ms, 192 tiles 31.3 ms; with 10 000 experimental
// patterns it picks 128 tiles (60.7 ms; 64 tiles 63.1 ms, 256 tiles 65.5 ms).
// KPDI_UPLOAD_TILES=<tiles per piece> overrides.
std::vector<int64_t> upload_pieces(const kpdi_ctx *c, int64_t n_chunk, size_t row_bytes) {
  const int64_t tiles = (n_chunk + kpdi::TILE_DICT - 1) / kpdi::TILE_DICT;
  int64_t piece = tiles;
  if (c->sw.upload_tiles > 0) {
    piece = c->sw.upload_tiles;
  } else if (c->have_exp && c->m_pad > 0) {
    const int row_blocks = c->m_pad / kpdi::TILE_EXP;
    const double t_tile = 2.0 * kpdi::TILE_EXP * kpdi::TILE_DICT * c->kpad / (157.3e12 / 256 * 0.88);
    const double t_row = row_bytes / 56e9, t_fixed = 0.18e-3;
    auto sweep_time = [&](int64_t t) {
      int rpl = 0;
      const int ns = choose_nsplit(c, row_blocks, (int)t, &rpl);
      return ((row_blocks + rpl - 1) / rpl) * (double)((t + ns - 1) / ns) * t_tile + t_fixed;
    };
    double best = 1e30;
    for (int64_t cand = 32; cand <= 512 + 8; cand += 8) {
      const int64_t p = cand > 512 ? tiles : std::min(cand, tiles);  // last candidate: one piece
      const double s_full = sweep_time(p), s_rest = tiles % p ? sweep_time(tiles % p) : 0.0;
      double copied = 0, swept = 0;
      for (int64_t left = tiles; left > 0; left -= p) {
        const int64_t t = std::min(p, left);
        copied += t * kpdi::TILE_DICT * t_row;
        swept = std::max(swept, copied) + (t == p ? s_full : s_rest);
      }
      if (swept < best - 1e-9) best = swept, piece = p;
    }
  }
  if (uses16(c)) piece = (piece + 1) / 2 * 2;  // whole 256-pattern tiles of match16.hip
  std::vector<int64_t> out;
  for (int64_t left = n_chunk, per = piece * kpdi::TILE_DICT; left > 0; left -= per) out.push_back(std::min(per, left));
  return out;
}

// a new resident chunk: its prepared buffer, sized for n patterns
int new_held_chunk(kpdi_ctx *c, int64_t n_chunk, int64_t global_start, float **out) {
  c->held.emplace_back();
  kpdi_ctx::HeldChunk &h = c->held.back();
  const hipError_t e = h.y.reserve((size_t)kpdi::round_up(n_chunk, dict_tile(c)) * c->kpad * sizeof(float));
  if (e != hipSuccess) {
    c->held.pop_back();
    return fail(KPDI_ENOMEM, "no device memory for a resident chunk of %lld patterns: %s", (long long)n_chunk,
                hipGetErrorString(e));
  }
  h.n = n_chunk;
  h.start = global_start;
  *out = h.y.as<float>();
  return KPDI_OK;
}

void release_held(kpdi_ctx *c) {
  if (c->held.empty()) return;
  (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  for (auto &h : c->held) h.y.release();
  c->held.clear();
}

}  // namespace

namespace {

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

}  // namespace

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
                    &c->part_i, &c->tail_s, &c->tail_i, &c->list16, &c->run_s[0], &c->run_s[1], &c->run_i[0], &c->run_i[1], &c->loc_s, &c->loc_i,
                    &c->bound_s, &c->bound_i, &c->gthr, &c->tile_ctr, &c->gather_s, &c->gather_i, &c->bg, &c->taps, &c->inv_map, &c->pre_scratch,
                    &c->mp_packed, &c->dcos, &c->rot, &c->proj_out,
                    &c->ref_raw, &c->ref_map, &c->ref_rowcol, &c->ref_pat, &c->ref_sqn, &c->ref_in, &c->ref_out,
                    &c->ref_idx, &c->osm_idx, &c->osm_out, &c->stage[0], &c->stage[1]})
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
  }
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

int kpdi_push_dictionary_chunk(kpdi_ctx *c, const void *patterns, int dtype, int64_t n_chunk, int64_t global_start) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!c->have_problem) return fail(KPDI_EINVAL, "kpdi_set_problem has not been called");
  if (!patterns) return fail(KPDI_EINVAL, "patterns pointer is NULL");
  const size_t es = kpdi::dtype_size(dtype);
  if (es == 0) return fail(KPDI_EINVAL, "unknown dtype %d", dtype);
  if (n_chunk <= 0) return fail(KPDI_EINVAL, "dictionary chunk must hold at least one pattern");
  int rc = use_device(c, true);
  if (rc) return rc;
  // the sweep of the last piece may still be running on return (KPDI_COMPUTE_F64: with the look at its certification
  // left to the next call on the context, resolve_exact64)
  c->pend64.defer = true;
  rc = staged_upload(c, patterns, (size_t)c->npix * es, upload_pieces(c, n_chunk, (size_t)c->npix * es),
                     [&](const void *d_piece, int64_t n, int64_t offset) {
                       return push_chunk_dev(c, d_piece, dtype, n, global_start + offset);
                     });
  c->pend64.defer = false;
  return rc;
}

int kpdi_push_dictionary_chunk_dev(kpdi_ctx *c, const void *d_patterns, int dtype, int64_t n_chunk,
                                   int64_t global_start) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!d_patterns) return fail(KPDI_EINVAL, "patterns pointer is NULL");
  int rc = use_device(c);
  if (rc) return rc;
  return push_chunk_dev(c, d_patterns, dtype, n_chunk, global_start);
}

// ---- dictionary generation --------------------------------------------------
int kpdi_hold_dictionary_chunk(kpdi_ctx *c, const void *patterns, int dtype, int64_t n_chunk, int64_t global_start) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!patterns) return fail(KPDI_EINVAL, "patterns pointer is NULL");
  int rc = check_chunk_args(c, dtype, n_chunk, global_start);
  if (rc) return rc;
  rc = use_device(c);
  if (rc) return rc;
  float *y = nullptr;
  decide_form(c, n_chunk);
  rc = new_held_chunk(c, n_chunk, global_start, &y);
  if (rc) return rc;
  // pieces of whole tiles, so that every piece is prepared straight into its place
  std::vector<int64_t> pieces;
  for (int64_t left = n_chunk, per = 192 * kpdi::TILE_DICT; left > 0; left -= per) pieces.push_back(std::min(per, left));
  const size_t kpad = c->kpad;
  rc = staged_upload(c, patterns, (size_t)c->npix * kpdi::dtype_size(dtype), pieces,
                     [&](const void *d_piece, int64_t n, int64_t offset) {
                       return prepare_chunk(c, d_piece, dtype, n, y + (size_t)offset * kpad);
                     });
  if (rc) {
    (void)hipStreamSynchronize(c->stream);
    c->held.back().y.release();
    c->held.pop_back();
  }
  return rc;
}

int kpdi_hold_dictionary_chunk_dev(kpdi_ctx *c, const void *d_patterns, int dtype, int64_t n_chunk,
                                   int64_t global_start) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!d_patterns) return fail(KPDI_EINVAL, "patterns pointer is NULL");
  int rc = check_chunk_args(c, dtype, n_chunk, global_start);
  if (rc) return rc;
  rc = use_device(c);
  if (rc) return rc;
  float *y = nullptr;
  decide_form(c, n_chunk);
  rc = new_held_chunk(c, n_chunk, global_start, &y);
  if (rc) return rc;
  return prepare_chunk(c, d_patterns, dtype, n_chunk, y);
}

int kpdi_sweep_held(kpdi_ctx *c) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!c->have_problem) return fail(KPDI_EINVAL, "kpdi_set_problem has not been called");
  if (!c->have_exp) return fail(KPDI_EINVAL, "kpdi_set_experimental has not been called");
  if (c->held.empty()) return fail(KPDI_EINVAL, "no resident dictionary: kpdi_hold_dictionary_chunk has not been called");
  int rc = use_device(c);
  if (rc) return rc;
  if (c->m == 0) return KPDI_OK;
  for (auto &h : c->held) {
    rc = sweep_prepared(c, h.y.as<float>(), h.n, h.start);
    if (rc) return rc;
  }
  return KPDI_OK;
}

int kpdi_release_held(kpdi_ctx *c) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  int rc = use_device(c);
  if (rc) return rc;
  release_held(c);
  return KPDI_OK;
}

int kpdi_held_size(kpdi_ctx *c, int64_t *n_patterns, int64_t *n_bytes) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  int64_t n = 0, bytes = 0;
  for (auto &h : c->held) {
    n += h.n;
    bytes += (int64_t)h.y.cap;
  }
  if (n_patterns) *n_patterns = n;
  if (n_bytes) *n_bytes = bytes;
  return KPDI_OK;
}

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

namespace {
// rotations (host) -> device, then one pattern per rotation into `d_out`; `pcs` != NULL: one
// PC per pattern with the detector shape / orientation of `geom`
struct VarPc {
  const double *pcs;
  int nrows, ncols;
  const double *om;
};
int project_to_device(kpdi_ctx *c, const double *rotations, int64_t n, int rescale, double out_min, double out_max,
                      int dtype_out, void *d_out, const VarPc *var = nullptr) {
  if (!c->have_master) return fail(KPDI_EINVAL, "kpdi_set_master_pattern has not been called");
  if (!var && !c->have_dc) return fail(KPDI_EINVAL, "kpdi_set_detector has not been called");
  if (!rotations) return fail(KPDI_EINVAL, "rotations pointer is NULL");
  if (n <= 0 || n >= (int64_t)INT_MAX) return fail(KPDI_EINVAL, "need between 1 and 2^31-1 rotations per call");
  if (rescale && !(out_max > out_min)) return fail(KPDI_EINVAL, "rescale needs out_max > out_min");
  HIPCHK(c->rot.reserve((size_t)n * 7 * sizeof(double)));
  HIPCHK(hipMemcpyAsync(c->rot.p, rotations, (size_t)n * 4 * sizeof(double), hipMemcpyHostToDevice, c->stream));
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
  HIPCHK(hipStreamSynchronize(c->stream));
  return KPDI_OK;
}
}  // namespace

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

int kpdi_push_rotations_chunk(kpdi_ctx *c, const double *rotations, int64_t n, int64_t global_start, int rescale,
                              double out_min, double out_max) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!c->have_problem) return fail(KPDI_EINVAL, "kpdi_set_problem has not been called");
  int rc = use_device(c);
  if (rc) return rc;
  if (c->have_dc && c->dc_npix != c->npix)
    return fail(KPDI_EINVAL, "detector has %lld pixels but the problem's signal shape has %d", (long long)c->dc_npix,
                c->npix);
  if (n > 0) HIPCHK(c->dict_raw.reserve((size_t)n * c->npix * sizeof(float)));
  rc = project_to_device(c, rotations, n, rescale, out_min, out_max, KPDI_F32, c->dict_raw.p);
  if (rc) return rc;
  return push_chunk_dev(c, c->dict_raw.p, KPDI_F32, n, global_start);
}

int kpdi_hold_rotations_chunk(kpdi_ctx *c, const double *rotations, int64_t n, int64_t global_start, int rescale,
                              double out_min, double out_max) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  int rc = check_chunk_args(c, KPDI_F32, n, global_start);
  if (rc) return rc;
  rc = use_device(c);
  if (rc) return rc;
  if (c->have_dc && c->dc_npix != c->npix)
    return fail(KPDI_EINVAL, "detector has %lld pixels but the problem's signal shape has %d", (long long)c->dc_npix,
                c->npix);
  HIPCHK(c->dict_raw.reserve((size_t)n * c->npix * sizeof(float)));
  rc = project_to_device(c, rotations, n, rescale, out_min, out_max, KPDI_F32, c->dict_raw.p);
  if (rc) return rc;
  float *y = nullptr;
  decide_form(c, n);
  rc = new_held_chunk(c, n, global_start, &y);
  if (rc) return rc;
  return prepare_chunk(c, c->dict_raw.p, KPDI_F32, n, y);
}

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

size_t kpdi_dtype_size(int dtype) { return kpdi::dtype_size(dtype); }

int kpdi_reset_topk(kpdi_ctx *c) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  int rc = use_device(c);
  if (rc) return rc;
  c->run_valid = false;
  c->final_valid = false;
  return KPDI_OK;
}

}  // extern "C"
namespace {
// float64 arithmetic: the running double lists (all-gathered and merged over the ranks) to the host;
// exactly one of scores64 / scores32 is set
int finalize64(kpdi_ctx *c, double *scores64, float *scores32, int64_t *indices_out) {
  const int k = c->keep_n;
  const size_t n = (size_t)c->m * k;
  const double *d_s = c->run64_s.as<double>();
  const int *d_i = c->run64_i.as<int>();
  // the lists of all ranks: RCCL all-gather (one process per GPU, or an in-process communicator), or - members of an
  // in-process group with peer-copy gather - already copied into the gather buffers by kpdi::root_gather_p2p
  const int ranks = c->p2p_ranks ? c->p2p_ranks : (c->comm ? c->nranks : 0);
  const bool peer_copied = c->p2p_ranks > 0;
  c->p2p_ranks = 0;
  c->cnt.gather_ranks = ranks;
  if (ranks) {
    HIPCHK(c->final64_s.reserve(n * sizeof(double)));
    HIPCHK(c->final64_i.reserve(n * sizeof(int)));
    if (!peer_copied) {
      HIPCHK(c->gather64_s.reserve(n * ranks * sizeof(double)));
      HIPCHK(c->gather64_i.reserve(n * ranks * sizeof(int)));
      ncclResult_t r;
      {
        ScopedTimer t(c, &c->ev_comm);
        r = g_rccl.GroupStart();
        if (r == ncclSuccess) r = g_rccl.AllGather(d_s, c->gather64_s.p, n, ncclFloat64, c->comm, c->stream);
        if (r == ncclSuccess) r = g_rccl.AllGather(d_i, c->gather64_i.p, n, ncclInt32, c->comm, c->stream);
        ncclResult_t r2 = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = r2;
      }
      if (r != ncclSuccess) return fail(KPDI_ECOMM, "RCCL all-gather failed: %s", g_rccl.GetErrorString(r));
    }
    // merge64_kernel ranks a pattern's candidates in LDS (12 bytes each): the per-rank lists join in groups that fit -
    // all at once for ordinary keep_n, a few ranks at a time for very long lists (8 ranks x keep_n > 1600 exceeded the
    // LDS of one launch and used to fail here, after the whole sweep, with a bare HIP error)
    const size_t lds_entries = (150 * 1024) / (sizeof(double) + sizeof(int));
    if ((size_t)2 * k > lds_entries)
      return fail(KPDI_EINVAL, "keep_n = %d is too large for the float64 merge of several ranks (limit %zu)", k, lds_entries / 2);
    {
      ScopedTimer t(c, &c->ev_merge);
      for (int r0 = 0; r0 < ranks;) {
        const size_t room = lds_entries - (r0 ? (size_t)k : 0);
        const int group = (int)std::min<size_t>(ranks - r0, std::max<size_t>(room / k, 1));
        kpdi::Merge64Launch g{};
        g.m = c->m;
        g.k = k;
        g.run_s = r0 ? c->final64_s.as<double>() : nullptr;  // the result so far (in place: read into LDS first)
        g.run_i = r0 ? c->final64_i.as<int>() : nullptr;
        g.cand_s64 = c->gather64_s.as<double>() + (size_t)r0 * n;
        g.cand_i = c->gather64_i.as<int>() + (size_t)r0 * n;
        g.lists = group;
        g.len = k;
        g.row_stride = k;
        g.list_stride = (int64_t)n;
        g.out_s = c->final64_s.as<double>();
        g.out_i = c->final64_i.as<int>();
        HIPCHK(kpdi::launch_merge64(g, c->stream));
        r0 += group;
      }
    }
    d_s = c->final64_s.as<double>();
    d_i = c->final64_i.as<int>();
  }
  c->final_idx = d_i;
  c->final_valid = true;
  if (!indices_out) return KPDI_OK;  // a group member that only takes part in the all-gather (kpdi::finalize_participate)
  // through the page-locked staging buffer of kpdi_finalize (a copy into pageable memory is pinned on the fly by the
  // runtime: milliseconds, and slower kernels behind it)
  std::vector<double> hs_pageable;
  std::vector<int> hi_pageable;
  double *hs;
  int *hi;
  if (c->pin_out.reserve(n * (sizeof(double) + sizeof(int))) == hipSuccess) {
    hs = (double *)c->pin_out.p;
    hi = (int *)(hs + n);
  } else {
    (void)hipGetLastError();
    hs_pageable.resize(n);
    hi_pageable.resize(n);
    hs = hs_pageable.data();
    hi = hi_pageable.data();
  }
  c->result_i32 = nullptr;
  HIPCHK(hipMemcpyAsync(hs, d_s, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(hi, d_i, n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (size_t i = 0; i < n; ++i) {
    if (scores64) scores64[i] = hs[i];
    if (scores32) scores32[i] = (float)hs[i];
    indices_out[i] = (int64_t)hi[i];
  }
  return KPDI_OK;
}
}  // namespace
extern "C" {

int kpdi_finalize_f64(kpdi_ctx *c, double *scores_out, int64_t *indices_out) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!c->have_exp || !c->have_problem) return fail(KPDI_EINVAL, "nothing to finalise");
  if (!scores_out || !indices_out) return fail(KPDI_EINVAL, "output pointer is NULL");
  if (!c->exact64) return fail(KPDI_EINVAL, "kpdi_finalize_f64 needs a problem set up with KPDI_COMPUTE_F64");
  int rc = use_device(c);
  if (rc) return rc;
  if (c->m == 0) return KPDI_OK;
  rc = ensure_running(c);  // a rank that pushed nothing contributes empty lists
  if (rc) return rc;
  return finalize64(c, scores_out, nullptr, indices_out);
}

// kpdi_finalize in two halves.  finalize_enqueue: (all-gather + merge over the ranks,) the result's device-to-host copies
// into page-locked slot `slot`, an event behind them - nothing waits.  finalize_collect: wait for that event, hand the slot's
// contents to the caller.  kpdi_finalize = both; kpdi_finalize_async / kpdi_finalize_wait let a caller that indexes map
// after map queue the NEXT map's kernels before it collects this one's result (the hand-over - synchronisation, copies,
// widening the indices - is ~0.1 ms of host time per call during which the GPU otherwise idles: 3 % of one rank's 3 ms
// share of configs[1] at N = 8).
namespace {
// this rank's running lists, made presentable: a rank that pushed nothing contributes empty lists
int own_lists(kpdi_ctx *c) {
  int rc = ensure_running(c);
  if (rc) return rc;
  if (c->run_empty && !c->exact64) {
    const size_t n0 = (size_t)c->m * c->keep_n;
    rc = wait_result_copy(c);
    if (!rc) rc = queue_fill_topk(c, c->run_s[c->run_cur].as<float>(), c->run_i[c->run_cur].as<int>(), n0);
    if (!rc) rc = flush_fills(c);
    if (rc) return rc;
    c->run_empty = false;
  }
  return KPDI_OK;
}

// the FINAL lists of the sweep on this rank: its own, or - with a communicator / in an in-process group - the merge of
// every rank's (RCCL all-gather, or lists that kpdi::root_gather_p2p has already peer-copied into the gather buffers)
int final_lists(kpdi_ctx *c, const float **out_s, const int **out_i) {
  int rc = own_lists(c);
  if (rc) return rc;
  const int k = c->keep_n;
  const size_t n = (size_t)c->m * k;
  const float *d_s = c->run_s[c->run_cur].as<float>();
  const int *d_i = c->run_i[c->run_cur].as<int>();
  const int ranks = c->p2p_ranks ? c->p2p_ranks : (c->comm ? c->nranks : 0);
  const bool peer_copied = c->p2p_ranks > 0;
  c->p2p_ranks = 0;
  c->cnt.gather_ranks = ranks;
  if (ranks) {  // also with one rank: keeps the RCCL path testable on a single GPU
    rc = wait_result_copy(c);  // (the merge below writes the other half of the ping-pong pair)
    if (rc) return rc;
    if (!peer_copied) {
      HIPCHK(c->gather_s.reserve(n * ranks * sizeof(float)));
      HIPCHK(c->gather_i.reserve(n * ranks * sizeof(int)));
      ncclResult_t r;
      {
        ScopedTimer t(c, &c->ev_comm);
        r = g_rccl.GroupStart();
        if (r == ncclSuccess) r = g_rccl.AllGather(d_s, c->gather_s.p, n, ncclFloat32, c->comm, c->stream);
        if (r == ncclSuccess) r = g_rccl.AllGather(d_i, c->gather_i.p, n, ncclInt32, c->comm, c->stream);
        ncclResult_t r2 = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = r2;
      }
      if (r != ncclSuccess) return fail(KPDI_ECOMM, "RCCL all-gather failed: %s", g_rccl.GetErrorString(r));
    }
    const int nxt = c->run_cur ^ 1;
    kpdi::MergeLaunch mg{};
    mg.m = c->m;
    mg.k = k;
    mg.n_src = 1;
    mg.src_scores[0] = c->gather_s.as<float>();
    mg.src_idx[0] = c->gather_i.as<int>();
    mg.src_lists[0] = ranks;
    mg.src_len[0] = k;
    mg.src_row_stride[0] = k;
    mg.src_list_stride[0] = (int)n;
    mg.out_scores = c->run_s[nxt].as<float>();
    mg.out_idx = c->run_i[nxt].as<int>();
    mg.out_stride = k;
    mg.out_offset = 0;
    {
      ScopedTimer t(c, &c->ev_merge);
      HIPCHK(kpdi::launch_merge(mg, c->stream));
    }
    d_s = c->run_s[nxt].as<float>();
    d_i = c->run_i[nxt].as<int>();
    // the per-rank running list (run_cur) is left untouched: finalize is idempotent
  }
  c->final_idx = d_i;
  c->final_valid = true;
  *out_s = d_s;
  *out_i = d_i;
  return KPDI_OK;
}

int finalize_enqueue(kpdi_ctx *c, int slot, bool own_stream) {
  const float *d_s = nullptr;
  const int *d_i = nullptr;
  int rc = final_lists(c, &d_s, &d_i);
  if (rc) return rc;
  const size_t n = (size_t)c->m * c->keep_n;
  kpdi_ctx::ResultSlot &rs = c->slots[slot];
  HIPCHK(rs.pin.reserve(n * (sizeof(float) + sizeof(int))));
  if (!rs.ready) HIPCHK(hipEventCreateWithFlags(&rs.ready, hipEventDisableTiming));
  rs.n = n;
  rs.pending = true;
  c->result_i32 = nullptr;
  float *h_s = (float *)rs.pin.p;
  if (!own_stream) {  // kpdi_finalize waits right away: the hop to another stream would only add latency (+15 us measured)
    HIPCHK(hipMemcpyAsync(h_s, d_s, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(h_s + n, d_i, n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipEventRecord(rs.ready, c->stream));
    return KPDI_OK;
  }
  if (!c->result_stream) {
    HIPCHK(hipStreamCreateWithFlags(&c->result_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->result_done, hipEventDisableTiming));
  }
  HIPCHK(hipEventRecord(c->result_done, c->stream));
  HIPCHK(hipStreamWaitEvent(c->result_stream, c->result_done, 0));
  HIPCHK(hipMemcpyAsync(h_s, d_s, n * sizeof(float), hipMemcpyDeviceToHost, c->result_stream));
  HIPCHK(hipMemcpyAsync(h_s + n, d_i, n * sizeof(int), hipMemcpyDeviceToHost, c->result_stream));
  HIPCHK(hipEventRecord(rs.ready, c->result_stream));
  c->result_copy = rs.ready;
  return KPDI_OK;
}

int finalize_collect(kpdi_ctx *c, int slot, float *scores_out, int64_t *indices_out) {
  kpdi_ctx::ResultSlot &rs = c->slots[slot];
  if (!rs.pending) return fail(KPDI_EINVAL, "no result is pending in slot %d", slot);
  HIPCHK(hipEventSynchronize(rs.ready));
  rs.pending = false;
  const size_t n = rs.n;
  const float *h_s = (const float *)rs.pin.p;
  const int *h_i = (const int *)(h_s + n);
  memcpy(scores_out, h_s, n * sizeof(float));
  for (size_t i = 0; i < n; ++i) indices_out[i] = (int64_t)h_i[i];
  c->result_i32 = (const int32_t *)h_i;  // (valid until this slot is used again: two finalize calls on)
  c->result_n = (int64_t)n;
  return KPDI_OK;
}

int finalize_args(kpdi_ctx *c) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!c->have_exp || !c->have_problem) return fail(KPDI_EINVAL, "nothing to finalise");
  return use_device(c);
}
}  // namespace

int kpdi_finalize(kpdi_ctx *c, float *scores_out, int64_t *indices_out) {
  int rc = finalize_args(c);
  if (rc) return rc;
  if (!scores_out || !indices_out) return fail(KPDI_EINVAL, "output pointer is NULL");
  if (c->m == 0) return KPDI_OK;
  if (c->exact64) {
    rc = ensure_running(c);
    if (rc) return rc;
    return finalize64(c, nullptr, scores_out, indices_out);
  }
  // the slot of an outstanding kpdi_finalize_async ticket is never touched (its copy may still be in flight and its
  // ticket must stay collectable): take the other one, or fail like kpdi_finalize_async does
  int slot = c->next_slot;
  if (c->slots[slot].pending) slot ^= 1;
  if (c->slots[slot].pending)
    return fail(KPDI_EINVAL, "two results are already pending: collect one with kpdi_finalize_wait first");
  c->next_slot = slot ^ 1;
  rc = finalize_enqueue(c, slot, false);
  if (rc) return rc;
  return finalize_collect(c, slot, scores_out, indices_out);
}

int kpdi_finalize_async(kpdi_ctx *c, int *ticket) {
  int rc = finalize_args(c);
  if (rc) return rc;
  if (!ticket) return fail(KPDI_EINVAL, "ticket is NULL");
  if (c->exact64) return fail(KPDI_EINVAL, "kpdi_finalize_async: not available in float64 arithmetic (use kpdi_finalize_f64)");
  if (c->m == 0) return fail(KPDI_EINVAL, "no experimental patterns to finalise");
  int slot = c->next_slot;
  if (c->slots[slot].pending) slot ^= 1;
  if (c->slots[slot].pending)
    return fail(KPDI_EINVAL, "two results are already pending: collect one with kpdi_finalize_wait first");
  c->next_slot = slot ^ 1;
  rc = finalize_enqueue(c, slot, true);
  if (rc) return rc;
  *ticket = slot;
  return KPDI_OK;
}

int kpdi_finalize_wait(kpdi_ctx *c, int ticket, float *scores_out, int64_t *indices_out) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!scores_out || !indices_out) return fail(KPDI_EINVAL, "output pointer is NULL");
  if (ticket < 0 || ticket > 1) return fail(KPDI_EINVAL, "bad ticket %d", ticket);
  int rc = use_device(c);
  if (rc) return rc;
  return finalize_collect(c, ticket, scores_out, indices_out);
}

int kpdi_pending_result_size(kpdi_ctx *c, int ticket, int64_t *n) {
  if (!c || !n) return fail(KPDI_EINVAL, "NULL argument");
  if (ticket < 0 || ticket > 1 || !c->slots[ticket].pending) return fail(KPDI_EINVAL, "no result is pending for ticket %d", ticket);
  *n = (int64_t)c->slots[ticket].n;
  return KPDI_OK;
}

int kpdi_result_indices_i32(kpdi_ctx *c, const int32_t **indices, int64_t *n) {
  if (!c || !indices || !n) return fail(KPDI_EINVAL, "NULL argument");
  *indices = c->result_i32;
  *n = c->result_i32 ? c->result_n : 0;
  return KPDI_OK;
}

int kpdi_comm_unique_id(uint8_t *id_out) {
  if (!id_out) return fail(KPDI_EINVAL, "id_out is NULL");
  if (!g_rccl.load()) return fail(KPDI_ECOMM, "cannot load librccl: %s", g_rccl.why.c_str());
  static_assert(sizeof(ncclUniqueId) == KPDI_UNIQUE_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return fail(KPDI_ECOMM, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
  memcpy(id_out, &id, sizeof id);
  return KPDI_OK;
}

int kpdi_comm_init(kpdi_ctx *c, int rank, int nranks, const uint8_t *id) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(KPDI_EINVAL, "bad rank %d / %d", rank, nranks);
  if (!id) return fail(KPDI_EINVAL, "id is NULL");
  int rc = use_device(c);
  if (rc) return rc;
  if (!g_rccl.load()) return fail(KPDI_ECOMM, "cannot load librccl: %s", g_rccl.why.c_str());
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, uid, rank);
  if (r != ncclSuccess) {
    c->comm = nullptr;
    return fail(KPDI_ECOMM, "ncclCommInitRank(rank %d of %d, device %d): %s", rank, nranks, c->device, g_rccl.GetErrorString(r));
  }
  c->rank = rank;
  c->nranks = nranks;
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
  if (on < 0 || on > 2) return fail(KPDI_EINVAL, "profiling level %d (0 off, 1 every phase, 2 match launches only)", on);
  c->profiling = on;
  return KPDI_OK;
}

int kpdi_get_counters(kpdi_ctx *c, kpdi_counters *out) {
  if (!c || !out) return fail(KPDI_EINVAL, "NULL argument");
  int rc = use_device(c);
  if (rc) return rc;
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
  c->cnt.f64_certificate = c->exact64 ? (c->sw.f64_worstcase ? 2 : 1) : 0;
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
  c->cnt.kpad = kpad;
  c->cnt.k_kept = kk;
  c->cnt.gather_ranks = gr;
  return KPDI_OK;
}

}  // extern "C"

// ---- hooks for in-process groups of contexts (group.hip; declared in group_hooks.h) -------------------------------
namespace kpdi {

const char *thread_error() { return g_err.c_str(); }

// one RCCL communicator over the contexts of ONE process (ncclCommInitAll: no unique id, no sockets, no environment)
int comm_init_all(kpdi_ctx *const *ctx, int n) {
  if (!ctx || n < 1) return fail(KPDI_EINVAL, "comm_init_all: no contexts");
  if (!g_rccl.load()) return fail(KPDI_ECOMM, "cannot load librccl: %s", g_rccl.why.c_str());
  std::vector<int> devs(n);
  for (int i = 0; i < n; ++i) {
    if (!ctx[i]) return fail(KPDI_EINVAL, "comm_init_all: context %d is NULL", i);
    if (ctx[i]->comm) return fail(KPDI_EINVAL, "comm_init_all: context %d already has a communicator", i);
    devs[i] = ctx[i]->device;
  }
  std::vector<ncclComm_t> comms(n, nullptr);
  ncclResult_t r = g_rccl.CommInitAll(comms.data(), n, devs.data());
  if (r != ncclSuccess) return fail(KPDI_ECOMM, "ncclCommInitAll over %d device(s): %s", n, g_rccl.GetErrorString(r));
  for (int i = 0; i < n; ++i) {
    ctx[i]->comm = comms[i];
    ctx[i]->rank = i;
    ctx[i]->nranks = n;
  }
  return KPDI_OK;
}

int finalize_precheck(kpdi_ctx *c, int kind) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!c->have_exp || !c->have_problem) return fail(KPDI_EINVAL, "nothing to finalise");
  if (kind == FINALIZE_F64 && !c->exact64) return fail(KPDI_EINVAL, "kpdi_finalize_f64 needs a problem set up with KPDI_COMPUTE_F64");
  if (kind == FINALIZE_ASYNC && c->exact64)
    return fail(KPDI_EINVAL, "kpdi_finalize_async: not available in float64 arithmetic (use kpdi_finalize_f64)");
  if (kind == FINALIZE_ASYNC && c->m == 0) return fail(KPDI_EINVAL, "no experimental patterns to finalise");
  if (!c->exact64 && c->m > 0 && c->slots[0].pending && c->slots[1].pending)
    return fail(KPDI_EINVAL, "two results are already pending: collect one with kpdi_finalize_wait first");
  return KPDI_OK;
}

void gather_abandon(kpdi_ctx *c) {
  if (c) c->p2p_ranks = 0;
}

int64_t sweep_round_rows(const kpdi_ctx *c) {
  if (!c || !c->have_exp || c->m_pad <= 0) return 4096;
  const int row_blocks = c->m_pad / kpdi::TILE_EXP;
  const int per_row_block = std::max(1, c->n_cu * kpdi::match_blocks_per_cu() / row_blocks);
  return (int64_t)per_row_block * kpdi::F16_TILE;
}

// RCCL gather, members other than the one that hands the result to the host: the all-gather + merge of
// kpdi_finalize without the copies (every rank of a collective has to take part in it)
int finalize_participate(kpdi_ctx *c) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!c->have_exp || !c->have_problem) return fail(KPDI_EINVAL, "nothing to finalise");
  int rc = use_device(c);
  if (rc) return rc;
  if (c->m == 0) return KPDI_OK;
  if (c->exact64) {
    rc = ensure_running(c);
    return rc ? rc : finalize64(c, nullptr, nullptr, nullptr);
  }
  const float *d_s = nullptr;
  const int *d_i = nullptr;
  return final_lists(c, &d_s, &d_i);
}

// peer-copy gather, every member: its running lists are presentable and final - an event on its stream says when
int member_lists_ready(kpdi_ctx *c, ListsView *v) {
  if (!c || !v) return fail(KPDI_EINVAL, "NULL argument");
  if (!c->have_exp || !c->have_problem) return fail(KPDI_EINVAL, "nothing to finalise");
  int rc = use_device(c);
  if (rc) return rc;
  *v = ListsView{};
  v->device = c->device;
  v->f64 = c->exact64;
  v->n = (size_t)c->m * c->keep_n;
  if (c->m == 0) return KPDI_OK;
  rc = own_lists(c);
  if (rc) return rc;
  if (c->exact64) {
    v->scores = c->run64_s.p;
    v->idx = c->run64_i.as<int>();
  } else {
    v->scores = c->run_s[c->run_cur].p;
    v->idx = c->run_i[c->run_cur].as<int>();
  }
  if (!c->lists_final) HIPCHK(hipEventCreateWithFlags(&c->lists_final, hipEventDisableTiming));
  HIPCHK(hipEventRecord(c->lists_final, c->stream));
  v->ready = c->lists_final;
  return KPDI_OK;
}

// peer-copy gather, the root member: every member's lists -> the root's gather buffers (hipMemcpyPeerAsync on the
// root's stream behind the members' events; xGMI between devices, a plain device copy when members share a device).
// The root's next finalize merges them exactly like all-gathered ones.  *read_done: recorded behind the copies.
int root_gather_p2p(kpdi_ctx *c, const ListsView *v, int n, hipEvent_t *read_done) {
  if (!c || !v || n < 1 || !read_done) return fail(KPDI_EINVAL, "root_gather_p2p: bad arguments");
  int rc = use_device(c);
  if (rc) return rc;
  *read_done = nullptr;
  const size_t cnt = (size_t)c->m * c->keep_n;
  if (cnt == 0) return KPDI_OK;
  const size_t es = c->exact64 ? sizeof(double) : sizeof(float);
  DevBuf &gs = c->exact64 ? c->gather64_s : c->gather_s;
  DevBuf &gi = c->exact64 ? c->gather64_i : c->gather_i;
  for (int j = 0; j < n; ++j)
    if (v[j].n != cnt || v[j].f64 != c->exact64 || !v[j].scores || !v[j].idx)
      return fail(KPDI_EINVAL, "group member %d holds %zu list entries (%s), the root %zu (%s): the members of a group must "
                  "be set up alike", j, v[j].n, v[j].f64 ? "float64" : "float32", cnt, c->exact64 ? "float64" : "float32");
  HIPCHK(gs.reserve(cnt * n * es));
  HIPCHK(gi.reserve(cnt * n * sizeof(int)));
  {
    ScopedTimer t(c, &c->ev_comm);
    for (int j = 0; j < n; ++j) {
      HIPCHK(hipStreamWaitEvent(c->stream, v[j].ready, 0));
      HIPCHK(hipMemcpyPeerAsync((char *)gs.p + (size_t)j * cnt * es, c->device, v[j].scores, v[j].device, cnt * es, c->stream));
      HIPCHK(hipMemcpyPeerAsync((char *)gi.p + (size_t)j * cnt * sizeof(int), c->device, v[j].idx, v[j].device,
                                cnt * sizeof(int), c->stream));
    }
  }
  if (!c->peer_read) HIPCHK(hipEventCreateWithFlags(&c->peer_read, hipEventDisableTiming));
  HIPCHK(hipEventRecord(c->peer_read, c->stream));
  *read_done = c->peer_read;
  c->p2p_ranks = n;
  return KPDI_OK;
}

// peer-copy gather, the other members: whoever next writes this member's lists waits for the root's copies of them
void member_lists_borrowed(kpdi_ctx *c, hipEvent_t read_done) {
  if (c && read_done) c->result_copy = read_done;
}

int context_device(const kpdi_ctx *c) { return c ? c->device : -1; }
int context_gather_ranks(const kpdi_ctx *c) { return c ? (c->comm ? c->nranks : 0) : 0; }

}  // namespace kpdi
