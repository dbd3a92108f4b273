// context.h - what the host translation units of libkpdi.so share: the context (one GPU, one stream), its buffers, the
// error convention, and the internal entry points between api.hip (life cycle, set-up, pre-processing, counters),
// sweep.hip (preparation + match + merge of a dictionary chunk, uploads, resident chunks), exact64.hip (float64
// arithmetic), finalize.hip (hand-over of the result, communicators, hooks of kpdi_group) and extras.hip (dictionary
// generation, refinement, orientation similarity map).  Nothing here is part of the C ABI (include/kpdi.h).
//
// What one context holds (all in the HBM of ONE MI355X):
//   raw experimental patterns (m_all x npix, caller's dtype)      - pre-processed in place
//   prepared experimental matrix X (m_pad x kpad f32)             - built once per set
//   raw + prepared dictionary chunk Y (n_pad x kpad f32)          - rebuilt per chunk
//   per-lane partial lists of the match kernel                    - [m_pad][2*nsplit][len]
//   running best-k (m x k: f32 score, i32 dictionary index)       - ping-pong pair
// The running best-k is the whole state of the sweep, exactly as in the
// reference's loop (indexing/_dictionary_indexing.py:97-98).
#pragma once
#include "../../include/kpdi.h"
#include "kernels.h"
#include "plan.h"
#include "group_hooks.h"

#include <dlfcn.h>
#include <limits.h>
#include <math.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace kpdi {

// the calling thread's error message (kpdi_last_error); returns `code`
int fail(int code, const char *fmt, ...);

#define HIPCHK(expr)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess)                                                                \
      return kpdi::fail(KPDI_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
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
  decltype(&ncclCommAbort) CommAbort = nullptr;  // (optional: older libraries lack it)
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
    CommAbort = (decltype(CommAbort))dlsym(h, "ncclCommAbort");
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
extern Rccl g_rccl;  // (finalize.hip)

}  // namespace kpdi

struct kpdi_ctx {
  int device = 0;
  int n_cu = 256;
  hipStream_t stream = nullptr;
  kpdi::Switches sw;

  // problem
  bool have_problem = false;
  int sy = 0, sx = 0, npix = 0;
  int k_kept = 0, kpad = 0;
  bool have_sig_mask = false;
  kpdi::DevBuf pix_map;  // int[k_kept]
  // signal mask as gather descriptors, one per 4 kept pixels (prep.hip: prep_wave_gather_kernel), when every such
  // quad lies in at most two runs of consecutive detector pixels (a circular mask: one run per detector row)
  kpdi::DevBuf quad_desc;
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
  kpdi::DevBuf exp_raw, row_map, exp_x;

  // dictionary chunk
  kpdi::DevBuf dict_raw, dict_y;
  // prepared chunks kept resident for sweeps against several experimental sets
  struct HeldChunk {
    kpdi::DevBuf y;
    int64_t n = 0, start = 0;
    kpdi::IndexSegments seg;  // seg.n > 1: several small held chunks prepared as ONE matrix (rows -> dictionary indices)
  };
  std::vector<HeldChunk> held;
  std::vector<int> kept_pixels;  // host copy of pix_map: tells whether a new problem keeps the layout
  // Small chunks waiting for ONE sweep (sweep.hip: "coalescing").  The reference's loop hands the metric a tenth of the
  // dictionary per iteration (doc/tutorials/pattern_matching.ipynb:582); swept alone such a chunk fills three quarters
  // of one tile round.  Their RAW patterns are appended here and prepared + matched together once a few rounds have
  // come in (or when the result is asked for); the merge translates rows back to dictionary indices (IndexSegments).
  struct PendingChunks {
    kpdi::DevBuf raw;      // [capacity rows][npix] of `dtype`
    // Pushed chunks only: a SECOND buffer, filled while the rows of the first are still waiting for their preparation
    // kernel - so that a host chunk can be uploaded STRAIGHT into its rows on the copy stream (no staging buffer, no
    // device-to-device copy queued behind the sweeps on the compute stream: that copy held the staging buffers until the
    // sweep in front of it had finished, and the uploads of a chunked call ran after its sweeps instead of beside them).
    kpdi::DevBuf raw_b;
    int cur = 0;                                        // 0: `raw` is being filled, 1: `raw_b`
    hipEvent_t consumed[2] = {nullptr, nullptr};        // behind the preparation kernel that read buffer i (compute stream)
    bool consumed_set[2] = {false, false};
    hipEvent_t filled = nullptr;                        // behind the last upload into the current buffer (copy stream)
    bool filled_pending = false;                        // ... which the next flush has to wait for
    kpdi::DevBuf &buf() { return cur ? raw_b : raw; }
    int dtype = -1;
    int64_t rows = 0, capacity = 0;
    struct Segment {
      int64_t row0, n, start;  // rows [row0, row0 + n) hold the dictionary patterns [start, start + n)
    };
    std::vector<Segment> seg;
  } pending,       // pushed chunks: swept together
    pending_hold;  // chunks to be held resident (kpdi_hold_*): prepared together into ONE held chunk
  // host-pointer pushes are cut into pieces whose upload (copy stream) overlaps the sweep of
  // the previous piece (compute stream): two staging buffers, events for hand-over
  kpdi::DevBuf stage[2];
  hipStream_t stream2 = nullptr;  // second compute stream of multi-launch sweeps
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipStream_t copy_stream = nullptr;
  int stage_next = 0;
  hipEvent_t stage_filled[2] = {nullptr, nullptr}, stage_free[2] = {nullptr, nullptr};

  // top-k state
  kpdi::DevBuf part_s, part_i;       // partial lists of one match launch
  kpdi::DevBuf part_cnt;             // match16.hip: entries every partial list holds (only those are written)
  bool part_counted = false;         // the last run_match's partial lists come with counts
  kpdi::DevBuf tail_s, tail_i;       // partial lists of the quarter-tile tail launch (match.hip: ROWT = 1)
  kpdi::DevBuf epi_stats;            // 4 x u64: what the epilogues of match16.hip did (profiling level 3; kpdi_counters.epi_*)
  kpdi::DevBuf list16;               // float16 form: home of the per-lane lists during a launch (match16.hip)
  int tail_nsplit = 0;         // lists per pattern / 2 of the last run_match's tail launch, 0 = none
  int tail_lists = 0;          // lists per pattern in tail_s / tail_i after the last run_match (0 = none): 2 * tail_nsplit
                               // (match.hip's tail launch) or 1 (tailgemm.hip)
  kpdi::DevBuf tail_scores;          // tailgemm.hip: scores of the last partial round, [rows][m_pad]
  kpdi::DevBuf run_s[2], run_i[2];   // running best-k ping-pong
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
    int gemm_rows = 0;  // float32 form of match16.hip: dictionary rows behind the whole rounds that tailgemm.hip takes (0 = none)
  } preplan;
  bool final_valid = false;       // `final_idx` points at the lists kpdi_finalize handed out last
  const int *final_idx = nullptr;
  kpdi::DevBuf osm_idx, osm_out;
  kpdi::DevBuf gthr;                            // shared rejection bound of the match kernel
  int bound_key = -1;                     // plan the bound array was initialised for (-1: none)
  kpdi::DevBuf tile_ctr;                        // dynamic tile counters of the match kernel
  kpdi::DevBuf loc_s, loc_i, bound_s, bound_i;  // multi-pass (keep_n > 32)
  kpdi::DevBuf gather_s, gather_i;              // RCCL all-gather target
  // float64 arithmetic (KPDI_COMPUTE_F64): the f32 path screens, rescore.hip rescores and keeps the best-k in double
  bool exact64 = false;
  kpdi::DevBuf run64_s, run64_i;                // running float64 best-k [m][keep_n]
  kpdi::DevBuf cand64;                          // float64 scores of the screened candidates [m][columns]
  kpdi::DevBuf cert64;                          // [0]: bits of max |f32 - f64| over the sweep; [1]: uncertified patterns of a merge
  kpdi::DevBuf gather64_s, gather64_i, final64_s, final64_i;
  kpdi::PinBuf pin_out;                         // float64 results on their way to the caller
  // the certification read-back of the last float64 chunk, not yet looked at (sweep_exact64 / resolve_exact64)
  struct Pending64 {
    bool active = false, defer = false;
    const float *y = nullptr;
    const void *raw = nullptr;
    int raw_dtype = 0, n_tiles = 0, nsplit = 0, rows_per_launch = 0, cap = 0, done = 0, extra = 0;
    int64_t n_chunk = 0, global_start = 0;
    hipEvent_t ready = nullptr;
    kpdi::PinBuf flag;  // int: patterns the last merge could not certify
  } pend64;
  // kpdi_finalize[_async]: two page-locked slots (scores + indices on their way to the caller) with an event each
  struct ResultSlot {
    kpdi::PinBuf pin;
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
  kpdi::DevBuf bg, taps, inv_map, pre_scratch;

  // dictionary generation (project.hip)
  bool have_master = false, have_dc = false;
  int mp_npx = 0, mp_npy = 0;
  int64_t dc_npix = 0;
  kpdi::DevBuf mp_packed, dcos, rot, proj_out;
  // rotations of pushed / held chunks go through a ring of page-locked slots (the caller's array may be a temporary:
  // it is copied here on the host, 32 bytes per pattern) so that no push has to wait for the stream
  struct RotStage {
    kpdi::PinBuf pin;
    hipEvent_t copied = nullptr;  // the upload out of this slot has run
  } rot_stage[4];
  int rot_next = 0;

  // refinement (refine.hip)
  bool have_ref = false;
  int ref_nrows = 0, ref_ncols = 0, ref_k = 0;
  int64_t ref_n = 0;
  double ref_om[9] = {};
  kpdi::DevBuf ref_raw, ref_map, ref_rowcol, ref_pat, ref_sqn, ref_in, ref_out, ref_idx;

  // comm
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  // kpdi_comm_drop has run: a kpdi_comm_init that is still inside ncclCommInitRank on another thread (it hung, its caller
  // gave up) must not install its communicator when it finally returns
  std::atomic<bool> comm_dropped{false};
  void *selftest_left[2] = {nullptr, nullptr};  // device buffers of a kpdi_comm_selftest that timed out, freed by kpdi_comm_drop
  // in-process groups (group.hip): the members' lists peer-copied into gather_s / gather_i (gather64_*) of the ROOT
  // member instead of an RCCL all-gather; `p2p_ranks` > 0 = that many lists are waiting there for the next finalize
  int p2p_ranks = 0;
  hipEvent_t lists_final = nullptr;  // this member's running lists are final (recorded on `stream`)
  hipEvent_t peer_read = nullptr;    // root: the peer copies of the members' lists have run

  // measurement: 0 off; 1 every phase bracketed by HIP events; 2 the match launches (and the all-gather) only - an event
  // record between two kernels costs ~6 us of idle GPU (profiles/r04_share_timeline.txt: 71 us per 3 ms step with level 1)
  int profiling = 0;
  bool timed(const void *list) const {
    return profiling == 1 || profiling == 3 || (profiling == 2 && (list == &ev_match || list == &ev_comm));
  }
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

namespace kpdi {

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

// ---- api.hip
int drain_events(kpdi_ctx *c, std::vector<std::pair<hipEvent_t, hipEvent_t>> &list, double *ms_sum);
int wait_result_copy(kpdi_ctx *c);
int flush_fills(kpdi_ctx *c);
int queue_fill(kpdi_ctx *c, void *p, size_t words, unsigned value, int bound_used = -1);
int queue_fill_topk(kpdi_ctx *c, float *scores, int *idx, size_t n);
void dtype_range(int dtype, float *omin, float *omax);
// first thing every entry point does.  `keep_pending`: the one caller (kpdi_push_dictionary_chunk) that starts its upload
// BEFORE it looks at the float64 certification of the previous chunk
int use_device(kpdi_ctx *c, bool keep_pending = false);
int results_to_host(kpdi_ctx *c, void *dst, const void *d_src, size_t bytes);

// what the prep kernels are told: `ndp` is evaluated in its centred form (prep.hip) except in the float16 form
inline int prep_metric(const kpdi_ctx *c) { return c->metric == KPDI_METRIC_NDP && c->compute != KPDI_COMPUTE_F16 ? 2 : c->metric; }
// the match kernel in use is match16.hip's (plane-major operand blocks, tiles of 256, lists in scratch)
inline bool uses16(const kpdi_ctx *c) { return c->compute == KPDI_COMPUTE_F16 || c->wide32; }
// operand form of the prepared matrices: 0 f32 tiled, 1 split f16, 2 float16, 3 f32 plane-major (kernels.h)
inline int operand_form(const kpdi_ctx *c) { return c->wide32 ? 3 : c->compute; }
// patterns per dictionary tile of the match kernel in use (the float16 form has its own kernel)
inline int dict_tile(const kpdi_ctx *c) { return uses16(c) ? f16_geometry(c->f16_waves).dict_tile : TILE_DICT; }
// lists per pattern and dictionary split the match kernel writes
inline int lists_per_split(const kpdi_ctx *c) { return uses16(c) ? 4 : 2; }
// entries ranked per pass when keep_n needs several (bounded) passes
inline int pass_entries(const kpdi_ctx *c) { return c->wide32 ? 20 : KMAX_LIMIT; }
// the planner's view of this context (plan.h)
inline plan::Env plan_env(const kpdi_ctx *c) {
  plan::Env e;
  e.n_cu = c->n_cu;
  e.blocks_per_cu = match_blocks_per_cu();
  e.sw = c->sw;
  return e;
}

// ---- sweep.hip
int flush_preprocess(kpdi_ctx *c, bool with_prep, bool *prep_done);
int prepare_experimental(kpdi_ctx *c);
int ensure_running(kpdi_ctx *c);
int run_match(kpdi_ctx *c, const float *dict_y, int n_chunk, int n_tiles, int nsplit, int rows_per_launch, int list_len,
              int64_t global_start, const float *bound_s, const int *bound_i, bool allow_tail = false);
int local_pass(kpdi_ctx *c, const float *y, int n_chunk, int n_tiles, int nsplit, int rows_per_launch, int64_t global_start,
               int done, int kp, int stride);
int prepare_chunk(kpdi_ctx *c, const void *d_patterns, int dtype, int64_t n_chunk, float *out);
void decide_form(kpdi_ctx *c, int64_t n_chunk);
int check_chunk_args(kpdi_ctx *c, int dtype, int64_t n_chunk, int64_t global_start);
// `may_wait`: the chunk is a whole push of the caller's (not a piece of a larger upload) and may wait, if it is small, for
// more chunks to be swept with (flush_pending: before anything reads or resets the running lists)
int push_chunk_dev(kpdi_ctx *c, const void *d_patterns, int dtype, int64_t n_chunk, int64_t global_start, bool may_wait = false);
int flush_pending(kpdi_ctx *c, bool hold = false);
void discard_pending(kpdi_ctx *c, bool hold = false);
// Where the next `n_chunk` rows of `dtype` (dictionary patterns from `global_start` on) would join the pending rows:
// *slot = device address for their raw patterns, or nullptr when this chunk cannot wait (it is then swept at once).
// What is pending may be swept first to make room / keep the order.  pending_commit() after the slot has been filled
// (by work queued on the context's stream) makes the rows part of the pending matrix and sweeps it when it is due.
// `hold`: the chunk is to stay resident (the pending rows become one held chunk instead of being swept).
int pending_slot(kpdi_ctx *c, int dtype, int64_t n_chunk, int64_t global_start, void **slot, bool hold = false);
int pending_commit(kpdi_ctx *c, int64_t n_chunk, int64_t global_start, bool hold = false, bool eager = false);
// hold a raw chunk resident in device memory: it joins the pending rows when it is small, else it is prepared at once
int hold_chunk_dev(kpdi_ctx *c, const void *d_patterns, int dtype, int64_t n_chunk, int64_t global_start, bool may_wait);
int sweep_prepared(kpdi_ctx *c, const float *y, int64_t n_chunk, int64_t global_start, const void *raw = nullptr, int raw_dtype = 0,
                   const IndexSegments *seg = nullptr);
int new_held_chunk(kpdi_ctx *c, int64_t n_chunk, int64_t global_start, float **out);
void release_held(kpdi_ctx *c);

// ---- exact64.hip
int resolve_exact64(kpdi_ctx *c);
int sweep_exact64(kpdi_ctx *c, const float *y, int64_t n_chunk, int64_t global_start, const void *raw, int raw_dtype, int n_tiles,
                  int nsplit, int rows_per_launch);
int finalize64(kpdi_ctx *c, double *scores64, float *scores32, int64_t *indices_out);

// ---- finalize.hip
int own_lists(kpdi_ctx *c);
int final_lists(kpdi_ctx *c, const float **out_s, const int **out_i);

// ---- extras.hip
// rotations (host) -> device, then one pattern per rotation into `d_out`; `var` != NULL: one PC per pattern with the
// detector shape / orientation of `var`
struct VarPc {
  const double *pcs;
  int nrows, ncols;
  const double *om;
};
// `stay_async`: return without waiting for the stream (the rotations are staged through the context's pinned ring)
int project_to_device(kpdi_ctx *c, const double *rotations, int64_t n, int rescale, double out_min, double out_max, int dtype_out,
                      void *d_out, const VarPc *var = nullptr, bool stay_async = false);

}  // namespace kpdi
