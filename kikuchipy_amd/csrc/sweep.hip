// sweep.hip - one dictionary chunk against the resident experimental set: preparation, the match launch(es) the
// planner (plan.h) lays out, the merge into the running best-k; the upload pipeline of host chunks; resident (held)
// chunks; and the C ABI entry points that push / hold chunks (include/kpdi.h).
// (one of the host translation units api.hip was split into in round 5: context.h holds what they share)
#include "context.h"

using namespace kpdi;

namespace kpdi {

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

// one match launch over the prepared chunk -> partial lists
//
// Tail: the dictionary tiles of a row block are shared by `nsplit` workgroups; when their number is a
// small non-multiple of nsplit (a rank's share of a sharded dictionary: 98 tiles over 16 workgroups)
// whole tiles would leave most workgroups idle during the last round (makespan 7 tile-times for 6.1 of
// work).  The last n_tiles % nsplit tiles are then handed out as QUARTER tiles by a second launch of the
// kernel's 32-row form, whose lists join the merge as a third source.
// What a match launch needs initialised before it starts - the shared bound (when its plan changes), the tile counters of
// the main and the tail launch - is QUEUED here (queue_fill), so that it shares one launch with whatever else the sweep
// initialises; the plan itself is returned for run_match.
typedef kpdi_ctx::MatchPlan MatchPlan;
static int match_setup(kpdi_ctx *c, int n_chunk, int n_tiles, int nsplit, int rows_per_launch, int list_len, bool bounded,
                bool allow_tail, MatchPlan *pl) {
  const int row_blocks = c->m_pad / kpdi::TILE_EXP;
  *pl = MatchPlan{};
  if (allow_tail && c->compute == KPDI_COMPUTE_F32 && !c->wide32)
    pl->tail_tiles = plan::classic_tail_tiles(plan_env(c), n_tiles, nsplit, row_blocks <= rows_per_launch, bounded);
  pl->n_main = n_tiles - pl->tail_tiles;
  if (allow_tail && c->compute == KPDI_COMPUTE_F32 && c->wide32 && !bounded) {
    // float32 form of match16.hip: the last partial round inside the kernel (halves / quarters of a tile) or as a kernel
    // of its own (tailgemm.hip), whichever the model prices lower
    int shift = 0, rows = 0;
    const double inside = plan::wide_tail(plan_env(c), n_tiles, nsplit, &shift);
    const double own = plan::wide_gemm_tail(plan_env(c), row_blocks, n_tiles, nsplit, n_chunk, &rows);
    if (rows > 0 && own < inside) {
      pl->gemm_rows = rows;
      pl->n_main = n_tiles - n_tiles % nsplit;
    }
  }
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
  pl->fixed_draws = plan::classic_fixed_draws(plan_env(c), pl->n_main, nsplit, pl->tail_tiles);
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
              int list_len, int64_t global_start, const float *bound_s, const int *bound_i, bool allow_tail) {
  c->tail_nsplit = 0;
  c->tail_lists = 0;
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
  const int lists_per_split = kpdi::lists_per_split(c);
  const size_t part = (size_t)c->m_pad * lists_per_split * nsplit * list_len;
  HIPCHK(c->part_s.reserve(part * sizeof(float)));
  HIPCHK(c->part_i.reserve(part * sizeof(int)));
  c->part_counted = f16;
  if (f16) HIPCHK(c->part_cnt.reserve((size_t)c->m_pad * lists_per_split * nsplit * sizeof(int)));
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
  ml.part_cnt = f16 ? c->part_cnt.as<int>() : nullptr;
  ml.bound_score = bound_s;
  ml.bound_idx = bound_i;
  ml.operand_form = operand_form(c);
  if (c->wide32 && pl.gemm_rows == 0) {  // (float32 form only: the same guards in the float16 schedule cost its 32-cycle MFMAs 10 %)
    (void)plan::wide_tail(plan_env(c), n_main, nsplit, &ml.tail_shift);
    ml.tail_first = n_main - n_main % nsplit;
  }
  if (f16) {
    // whole-tile rounds are walked in a permuted order (plan.h: tile_order_stride); partial units stay last
    const int whole = (c->wide32 && ml.tail_shift > 0) ? ml.tail_first : n_main;
    ml.perm_stride = plan::tile_order_stride(plan_env(c), whole, nsplit, &ml.perm_rounds);
    if (c->profiling == 3) {  // the developer level: the epilogues count what they do (~50 us of atomics per launch)
      if (!c->epi_stats.p) {
        HIPCHK(c->epi_stats.reserve(4 * sizeof(unsigned long long)));
        HIPCHK(hipMemsetAsync(c->epi_stats.p, 0, 4 * sizeof(unsigned long long), c->stream));
      }
      ml.epi_stats = c->epi_stats.as<unsigned long long>();
    }
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
      plan::xcd_grid(plan_env(c), rows, nsplit, dict_tile(c), f16, &xr, &xs, &rg);
      grid_rows = std::max(grid_rows, rg);
    }
    const size_t scratch16 = f16 ? kpdi::match16_scratch_bytes(grid_rows * nsplit, c->f16_waves, list_len) : 0;
    launched_rows = grid_rows;
    if (f16) HIPCHK(c->list16.reserve((two ? 2 : 1) * scratch16));
    for (int r0 = 0; r0 < row_blocks; r0 += rows_per_launch, ++j) {
      ml.row_first = r0;
      ml.rows = std::min(rows_per_launch, row_blocks - r0);
      plan::xcd_grid(plan_env(c), ml.rows, nsplit, dict_tile(c), f16, &ml.xcd_rows, &ml.xcd_splits, &ml.rows_grid);
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
      c->tail_lists = 2 * ns_t;
    }
    if (pl.gemm_rows > 0) {
      // the rows behind the whole rounds: scores by tailgemm.hip, then - the shared bound is final now - the few that reach
      // it become one sorted list per pattern, a merge source of its own
      const int rows = pl.gemm_rows, groups = (rows + 31) / 32;
      HIPCHK(c->tail_scores.reserve((size_t)groups * 32 * c->m_pad * sizeof(float)));
      HIPCHK(c->tail_s.reserve((size_t)c->m_pad * kpdi::tail_select_lists() * list_len * sizeof(float)));
      HIPCHK(c->tail_i.reserve((size_t)c->m_pad * kpdi::tail_select_lists() * list_len * sizeof(int)));
      kpdi::TailGemmLaunch tg;
      tg.dict = dict_y;
      tg.exp = c->exp_x.as<float>();
      tg.kpad = c->kpad;
      tg.tile_first = n_main;
      tg.row_groups = groups;
      tg.m_pad = c->m_pad;
      tg.scores = c->tail_scores.as<float>();
      HIPCHK(kpdi::launch_tail_gemm(tg, c->stream));
      kpdi::TailSelectLaunch ts;
      ts.scores = tg.scores;
      ts.rows = rows;
      ts.m = c->m;
      ts.m_pad = c->m_pad;
      ts.idx_first = (int)global_start + n_main * kpdi::F16_TILE;
      ts.gthr = c->gthr.as<unsigned>();
      ts.bound_grouped = pl.bound_grouped;
      ts.list_len = list_len;
      ts.out_scores = c->tail_s.as<float>();
      ts.out_idx = c->tail_i.as<int>();
      HIPCHK(kpdi::launch_tail_select(ts, c->stream));
      c->tail_lists = kpdi::tail_select_lists();
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

// Which f32 match kernel serves this sweep (plan.h: prefer_wide).  The two kernels read different operand layouts (and
// row paddings), so the choice is made when the first chunk of a sweep arrives - nothing prepared yet, no resident
// chunks - and stands until then again.  KPDI_F32_WIDE = 1 / 0 forces it.
void decide_form(kpdi_ctx *c, int64_t n_chunk) {
  if (c->compute != KPDI_COMPUTE_F32 || c->wide_mode >= 0) return;
  if (c->exp_prepared || !c->held.empty()) return;
  const int row_blocks = c->have_exp ? c->m_pad / kpdi::TILE_EXP : 16;
  const bool w = plan::prefer_wide(plan_env(c), row_blocks, c->k_kept, n_chunk);
  if (w == c->wide32) return;
  c->wide32 = w;
  c->kpad = kpdi::round_up(c->k_kept + (c->metric == KPDI_METRIC_NDP ? 1 : 0), w ? kpdi::F16_STEP / 2 : kpdi::TILE_K);
  c->cnt.kpad = c->kpad;
}

int check_chunk_args(kpdi_ctx *c, int dtype, int64_t n_chunk, int64_t global_start) {
  if (!c->have_problem) return fail(KPDI_EINVAL, "kpdi_set_problem has not been called");
  if (n_chunk <= 0) return fail(KPDI_EINVAL, "dictionary chunk must hold at least one pattern");
  if (kpdi::dtype_size(dtype) == 0) return fail(KPDI_EINVAL, "unknown dtype %d", dtype);
  if (global_start < 0 || global_start + n_chunk >= (int64_t)INT_MAX)
    return fail(KPDI_EINVAL, "dictionary indices must fit in int32");
  return KPDI_OK;
}

// prepare + sweep one raw chunk resident in device memory; `seg`: it is a coalesced matrix (rows -> dictionary indices)
static int sweep_raw_chunk(kpdi_ctx *c, const void *d_patterns, int dtype, int64_t n_chunk, int64_t global_start,
                           const IndexSegments *seg, hipEvent_t raw_consumed = nullptr) {
  int rc;
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
    ps.nsplit = plan::choose_nsplit(plan_env(c), uses16(c), row_blocks, ps.n_tiles, &ps.rows_per_launch);
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
  if (raw_consumed) HIPCHK(hipEventRecord(raw_consumed, c->stream));  // (the raw rows may be overwritten from here on)
  return sweep_prepared(c, c->dict_y.as<float>(), n_chunk, global_start, d_patterns, dtype, seg);
}

// ---- coalescing of small chunks (context.h: PendingChunks) ---------------------------------------------------------
// A chunk may wait when it is a whole push of fewer than two tile rounds, in f32 / f16 arithmetic (float64 rescoring
// reads the chunk's own raw patterns: it sweeps at once).  Chunks to be HELD resident (kpdi_hold_*) wait the same way
// and become one held chunk (eight rounds' worth at a time).
// It is appended when the pending rows are of its dtype, come before it in the dictionary (rows and indices must rise
// together: the match kernel breaks ties by row) and a segment is free; otherwise the pending rows are swept first.
// Three rounds' worth of rows are swept as soon as they are there.  KPDI_NO_COALESCE=1: every chunk at once (A/B).
static int64_t round_of(const kpdi_ctx *c) {
  return c->have_exp && c->m_pad > 0 ? plan::round_rows(plan_env(c), c->m_pad / kpdi::TILE_EXP) : 4096;
}

static bool may_coalesce(const kpdi_ctx *c, int64_t n_chunk, bool hold) {
  if (c->sw.no_coalesce || (!hold && c->exact64)) return false;
  return n_chunk < 2 * round_of(c);
}

void discard_pending(kpdi_ctx *c, bool hold) {
  kpdi_ctx::PendingChunks &p = hold ? c->pending_hold : c->pending;
  p.rows = 0;
  p.seg.clear();
}

int flush_pending(kpdi_ctx *c, bool hold) {
  kpdi_ctx::PendingChunks &p = hold ? c->pending_hold : c->pending;
  if (p.rows == 0) return KPDI_OK;
  IndexSegments seg;
  seg.n = (int)p.seg.size();
  for (int t = 0; t < seg.n; ++t) {
    seg.row0[t] = (int)p.seg[t].row0;
    seg.delta[t] = (int)(p.seg[t].start - p.seg[t].row0);
  }
  const int64_t rows = p.rows, start = p.seg[0].start;
  const bool one = seg.n == 1;  // a lone chunk: dictionary indices straight from the match kernel, as if it had not waited
  discard_pending(c, hold);
  if (hold) {  // the pending rows become ONE resident chunk
    float *y = nullptr;
    decide_form(c, rows);
    int rc = new_held_chunk(c, rows, one ? start : 0, &y);
    if (rc) return rc;
    if (!one) c->held.back().seg = seg;
    return prepare_chunk(c, p.raw.p, p.dtype, rows, y);
  }
  c->cnt.coalesced_sweeps += one ? 0 : 1;
  // rows uploaded on the copy stream (kpdi_push_dictionary_chunk) must have arrived; the buffer is handed to the
  // preparation kernel and the OTHER one takes what is pushed next
  if (p.filled_pending) {
    HIPCHK(hipStreamWaitEvent(c->stream, p.filled, 0));
    p.filled_pending = false;
  }
  const int b = p.cur;
  const void *raw = p.buf().p;
  if (!p.consumed[b]) HIPCHK(hipEventCreateWithFlags(&p.consumed[b], hipEventDisableTiming));
  p.cur ^= 1;
  p.capacity = 0;  // (of the buffer that is filled next: pending_slot sizes it)
  p.consumed_set[b] = true;
  return sweep_raw_chunk(c, raw, p.dtype, rows, one ? start : 0, one ? nullptr : &seg, p.consumed[b]);
}

int push_chunk_dev(kpdi_ctx *c, const void *d_patterns, int dtype, int64_t n_chunk, int64_t global_start, bool may_wait) {
  int rc = resolve_exact64(c);  // (before this chunk's preparation overwrites what extra passes of the last one would read)
  if (rc) return rc;
  rc = check_chunk_args(c, dtype, n_chunk, global_start);
  if (rc) return rc;
  if (!c->have_exp) return fail(KPDI_EINVAL, "kpdi_set_experimental has not been called");
  if (c->m == 0) return KPDI_OK;
  void *slot = nullptr;
  if (may_wait) {
    rc = pending_slot(c, dtype, n_chunk, global_start, &slot);
    if (rc) return rc;
  }
  if (!slot) {
    rc = flush_pending(c);
    return rc ? rc : sweep_raw_chunk(c, d_patterns, dtype, n_chunk, global_start, nullptr);
  }
  HIPCHK(hipMemcpyAsync(slot, d_patterns, (size_t)n_chunk * c->npix * kpdi::dtype_size(dtype), hipMemcpyDeviceToDevice, c->stream));
  return pending_commit(c, n_chunk, global_start);
}

int hold_chunk_dev(kpdi_ctx *c, const void *d_patterns, int dtype, int64_t n_chunk, int64_t global_start, bool may_wait) {
  void *slot = nullptr;
  int rc = KPDI_OK;
  if (may_wait) {
    rc = pending_slot(c, dtype, n_chunk, global_start, &slot, true);
    if (rc) return rc;
  }
  if (slot) {
    HIPCHK(hipMemcpyAsync(slot, d_patterns, (size_t)n_chunk * c->npix * kpdi::dtype_size(dtype), hipMemcpyDeviceToDevice, c->stream));
    return pending_commit(c, n_chunk, global_start, true);
  }
  rc = flush_pending(c, true);  // (held chunks keep the order they were handed over in)
  if (rc) return rc;
  float *y = nullptr;
  decide_form(c, n_chunk);
  rc = new_held_chunk(c, n_chunk, global_start, &y);
  return rc ? rc : prepare_chunk(c, d_patterns, dtype, n_chunk, y);
}

int pending_slot(kpdi_ctx *c, int dtype, int64_t n_chunk, int64_t global_start, void **slot, bool hold) {
  *slot = nullptr;
  if (!c->have_problem || !may_coalesce(c, n_chunk, hold)) return KPDI_OK;
  if (!hold && (!c->have_exp || c->m == 0)) return KPDI_OK;
  kpdi_ctx::PendingChunks &p = hold ? c->pending_hold : c->pending;
  const size_t row_bytes = (size_t)c->npix * kpdi::dtype_size(dtype);
  const int64_t round = round_of(c);
  if (p.rows > 0 && (p.dtype != dtype || (int)p.seg.size() == INDEX_SEGMENTS || p.rows + n_chunk > p.capacity ||
                     global_start < p.seg.back().start + p.seg.back().n)) {
    int rc = flush_pending(c, hold);
    if (rc) return rc;
  }
  kpdi::DevBuf &rb = p.buf();
  if (p.rows == 0) {
    // room for what goes together (up to eight rounds) + the chunk that takes it there; at most 2 GiB
    const int64_t want = std::min<int64_t>(10 * round, std::max<int64_t>((int64_t)((2ull << 30) / row_bytes), n_chunk));
    if (rb.cap < (size_t)want * row_bytes) {
      HIPCHK(hipStreamSynchronize(c->stream));  // (work queued earlier may still read the buffer that is about to go)
      HIPCHK(rb.reserve((size_t)want * row_bytes));
    }
    p.capacity = (int64_t)(rb.cap / row_bytes);
    p.dtype = dtype;
  }
  *slot = (char *)rb.p + (size_t)p.rows * row_bytes;
  return KPDI_OK;
}

int pending_commit(kpdi_ctx *c, int64_t n_chunk, int64_t global_start, bool hold, bool eager) {
  kpdi_ctx::PendingChunks &p = hold ? c->pending_hold : c->pending;
  const int64_t round = round_of(c);
  p.seg.push_back({p.rows, n_chunk, global_start});
  p.rows += n_chunk;
  if (!hold) c->final_valid = false;
  bool flush = p.rows + round / 4 > p.capacity || p.rows >= 8 * round;
  if (!flush && !hold && p.rows >= 3 * round) {
    // Three rounds are worth a launch set.  A chunk that came over the host link (`eager`) is swept at once: its sweep
    // runs beside the uploads that follow, and whatever is still pending when the last chunk has arrived is swept with
    // nothing left to hide behind.  Rows that were already on the device (resident or simulated chunks: the pushes only
    // queue work) gain nothing from an early sweep while the GPU is still busy with the one before - they wait for
    // company up to eight rounds: fewer launch ramps, less round padding.
    flush = eager || c->sw.no_coalesce_wait || hipStreamQuery(c->stream) == hipSuccess;
    (void)hipGetLastError();  // (hipErrorNotReady is an answer, not an error)
  }
  if (flush) return flush_pending(c, hold);
  return KPDI_OK;
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
  pm.src_cnt[0] = c->part_counted ? c->part_cnt.as<int>() : nullptr;
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

// every experimental pattern against one prepared chunk, merged into the running best-k
int sweep_prepared(kpdi_ctx *c, const float *y, int64_t n_chunk, int64_t global_start, const void *raw, int raw_dtype,
                   const IndexSegments *seg) {
  int rc = prepare_experimental(c);
  if (rc) return rc;
  rc = ensure_running(c);
  if (rc) return rc;
  const int n_tiles = kpdi::round_up(n_chunk, dict_tile(c)) / dict_tile(c);

  const int row_blocks = c->m_pad / kpdi::TILE_EXP;
  int rows_per_launch = row_blocks;
  const int nsplit = plan::choose_nsplit(plan_env(c), uses16(c), row_blocks, n_tiles, &rows_per_launch);
  const int k = c->keep_n;
  if (c->exact64) {
    if (seg) return fail(KPDI_EINVAL, "float64 arithmetic rescoring reads a chunk's own raw patterns: a dictionary held as coalesced chunks "
                                      "keeps only the prepared form - push the chunks instead");
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

  // a coalesced matrix is ranked by ROW (index base 0): every list the chunk contributes - also the bounded passes'
  // bounds, compared inside the match kernel - lives in row space until this merge translates it (rows and dictionary
  // indices rise together, so "after (score, row)" is "after (score, index)")
  const int64_t idx_base = seg ? 0 : global_start;
  if (seg) {
    mg.seg = *seg;
    mg.seg_sources = ~0u << ns;
  }
  if (k <= kpdi::KMAX_LIMIT) {
    const int len = kpdi::match_list_len(k);
    rc = run_match(c, y, (int)n_chunk, n_tiles, nsplit, rows_per_launch, len, idx_base, nullptr, nullptr, true);
    if (rc) return rc;
    mg.src_scores[ns] = c->part_s.as<float>();
    mg.src_idx[ns] = c->part_i.as<int>();
    mg.src_cnt[ns] = c->part_counted ? c->part_cnt.as<int>() : nullptr;
    const int lps = lists_per_split(c);
    mg.src_lists[ns] = lps * nsplit;
    mg.src_len[ns] = len;
    mg.src_row_stride[ns] = lps * nsplit * len;
    mg.src_list_stride[ns] = len;
    ++ns;
    if (c->tail_lists > 0) {
      mg.src_scores[ns] = c->tail_s.as<float>();
      mg.src_idx[ns] = c->tail_i.as<int>();
      mg.src_lists[ns] = c->tail_lists;
      mg.src_len[ns] = len;
      mg.src_row_stride[ns] = c->tail_lists * len;
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
      rc = local_pass(c, y, (int)n_chunk, n_tiles, nsplit, rows_per_launch, idx_base, done, kp, k);
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

}  // namespace kpdi

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

std::vector<int64_t> upload_pieces(const kpdi_ctx *c, int64_t n_chunk, size_t row_bytes) {
  const int row_blocks = c->have_exp && c->m_pad > 0 ? c->m_pad / kpdi::TILE_EXP : 0;
  return plan::upload_pieces(plan_env(c), uses16(c), row_blocks, c->kpad, n_chunk, row_bytes);
}

}  // namespace

namespace kpdi {

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
  discard_pending(c, true);
  if (c->held.empty()) return;
  (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  for (auto &h : c->held) h.y.release();
  c->held.clear();
}

}  // namespace kpdi

namespace {

// A small host chunk that may wait for company (coalescing, above) is uploaded STRAIGHT into its rows of the pending buffer
// on the copy stream: nothing of it is queued on the compute stream until the rows are swept, so the uploads of a chunked
// call run beside the sweeps of the chunks before them.  (Through a staging buffer + a device-to-device copy on the compute
// stream - staged_upload below - the copy sat behind the previous sweep, held its staging buffer until then and the next
// upload waited for it: configs[1] as 33 chunks from host memory took uploads + sweeps, 60 ms.)  *done = false: the chunk
// cannot wait (no slot); nothing has happened and the caller takes the staged path.
static int direct_upload(kpdi_ctx *c, const void *patterns, int dtype, size_t es, int64_t n_chunk, int64_t global_start, bool *done) {
  *done = false;
  int rc = resolve_exact64(c);
  if (rc) return rc;
  rc = check_chunk_args(c, dtype, n_chunk, global_start);
  if (rc) return rc;
  void *slot = nullptr;
  rc = pending_slot(c, dtype, n_chunk, global_start, &slot);
  if (rc || !slot) return rc;
  kpdi_ctx::PendingChunks &p = c->pending;
  if (!c->copy_stream) {
    HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
      HIPCHK(hipEventCreateWithFlags(&c->stage_filled[b], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&c->stage_free[b], hipEventDisableTiming));
      HIPCHK(hipEventRecord(c->stage_free[b], c->stream));
    }
  }
  if (!p.filled) HIPCHK(hipEventCreateWithFlags(&p.filled, hipEventDisableTiming));
  // (the rows this buffer held before have been read by their preparation kernel)
  if (p.consumed_set[p.cur]) HIPCHK(hipStreamWaitEvent(c->copy_stream, p.consumed[p.cur], 0));
  HIPCHK(hipMemcpyAsync(slot, patterns, (size_t)n_chunk * c->npix * es, hipMemcpyHostToDevice, c->copy_stream));
  HIPCHK(hipEventRecord(p.filled, c->copy_stream));
  p.filled_pending = true;
  *done = true;
  c->cnt.h2d_bytes += (double)n_chunk * c->npix * es;
  rc = pending_commit(c, n_chunk, global_start, false, true);  // (sweeps the pending rows when enough of them are there)
  const hipError_t e = hipStreamSynchronize(c->copy_stream);   // the caller's buffer is free again
  if (!rc && e != hipSuccess) return fail(KPDI_EHIP, "upload of a dictionary chunk: %s", hipGetErrorString(e));
  return rc;
}

}  // namespace

extern "C" {

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
  const std::vector<int64_t> pieces = upload_pieces(c, n_chunk, (size_t)c->npix * es);
  const bool one_piece = pieces.size() == 1;  // (pieces of a larger upload are sized for its pipeline: they sweep at once)
  if (one_piece && c->have_exp && c->m > 0 && !c->sw.no_direct_upload) {
    bool done = false;
    rc = direct_upload(c, patterns, dtype, es, n_chunk, global_start, &done);
    if (rc || done) {
      c->pend64.defer = false;
      return rc;
    }
  }
  rc = staged_upload(c, patterns, (size_t)c->npix * es, pieces,
                     [&](const void *d_piece, int64_t n, int64_t offset) {
                       return push_chunk_dev(c, d_piece, dtype, n, global_start + offset, one_piece);
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
  return push_chunk_dev(c, d_patterns, dtype, n_chunk, global_start, true);
}

// ---- dictionary generation --------------------------------------------------
int kpdi_hold_dictionary_chunk(kpdi_ctx *c, const void *patterns, int dtype, int64_t n_chunk, int64_t global_start) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!patterns) return fail(KPDI_EINVAL, "patterns pointer is NULL");
  int rc = check_chunk_args(c, dtype, n_chunk, global_start);
  if (rc) return rc;
  rc = use_device(c);
  if (rc) return rc;
  if (n_chunk <= 192 * kpdi::TILE_DICT) {
    // one upload piece: it may join the pending rows of small held chunks (hold_chunk_dev), else it is prepared at once
    return staged_upload(c, patterns, (size_t)c->npix * kpdi::dtype_size(dtype), {n_chunk},
                         [&](const void *d_piece, int64_t n, int64_t) { return hold_chunk_dev(c, d_piece, dtype, n, global_start, true); });
  }
  rc = flush_pending(c, true);  // (held chunks keep the order they were handed over in)
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
  return hold_chunk_dev(c, d_patterns, dtype, n_chunk, global_start, true);
}

int kpdi_sweep_held(kpdi_ctx *c) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!c->have_problem) return fail(KPDI_EINVAL, "kpdi_set_problem has not been called");
  if (!c->have_exp) return fail(KPDI_EINVAL, "kpdi_set_experimental has not been called");
  int rc = use_device(c);
  if (rc) return rc;
  rc = flush_pending(c, true);  // (small chunks that were waiting to be held together)
  if (rc) return rc;
  if (c->held.empty()) return fail(KPDI_EINVAL, "no resident dictionary: kpdi_hold_dictionary_chunk has not been called");
  if (c->m == 0) return KPDI_OK;
  rc = flush_pending(c);
  if (rc) return rc;
  for (auto &h : c->held) {
    rc = sweep_prepared(c, h.y.as<float>(), h.n, h.start, nullptr, 0, h.seg.n > 1 ? &h.seg : nullptr);
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
  if (c->pending_hold.rows > 0) {  // (chunks that are still waiting to be prepared together count)
    int rc = use_device(c);
    if (!rc) rc = flush_pending(c, true);
    if (rc) return rc;
  }
  int64_t n = 0, bytes = 0;
  for (auto &h : c->held) {
    n += h.n;
    bytes += (int64_t)h.y.cap;
  }
  if (n_patterns) *n_patterns = n;
  if (n_bytes) *n_bytes = bytes;
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
  if (c->have_exp && c->m > 0 && n > 0 && check_chunk_args(c, KPDI_F32, n, global_start) == KPDI_OK) {
    // a small chunk is SIMULATED straight into its place among the pending rows (no copy; sweep.hip: coalescing)
    rc = resolve_exact64(c);
    if (rc) return rc;
    void *slot = nullptr;
    rc = pending_slot(c, KPDI_F32, n, global_start, &slot);
    if (rc) return rc;
    if (slot) {
      rc = project_to_device(c, rotations, n, rescale, out_min, out_max, KPDI_F32, slot, nullptr, true);
      return rc ? rc : pending_commit(c, n, global_start);
    }
  }
  if (n > 0) HIPCHK(c->dict_raw.reserve((size_t)n * c->npix * sizeof(float)));
  rc = project_to_device(c, rotations, n, rescale, out_min, out_max, KPDI_F32, c->dict_raw.p, nullptr, true);
  if (rc) return rc;
  return push_chunk_dev(c, c->dict_raw.p, KPDI_F32, n, global_start, true);
}

int kpdi_push_rotations_chunk_varying_pc(kpdi_ctx *c, const double *rotations, const double *pcs, int64_t n,
                                         int64_t global_start, const double *om, int rescale, double out_min, double out_max) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!pcs || !om) return fail(KPDI_EINVAL, "NULL argument");
  if (!c->have_problem) return fail(KPDI_EINVAL, "kpdi_set_problem has not been called");
  int rc = check_chunk_args(c, KPDI_F32, n, global_start);
  if (rc) return rc;
  rc = use_device(c);
  if (rc) return rc;
  // (every pattern has its own direction cosines: projected into the raw chunk buffer, then swept like a resident chunk;
  // the projection centres are read before this returns)
  HIPCHK(c->dict_raw.reserve((size_t)n * c->npix * sizeof(float)));
  const VarPc var{pcs, c->sy, c->sx, om};
  rc = project_to_device(c, rotations, n, rescale, out_min, out_max, KPDI_F32, c->dict_raw.p, &var);
  if (rc) return rc;
  return push_chunk_dev(c, c->dict_raw.p, KPDI_F32, n, global_start, true);
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
  void *slot = nullptr;
  rc = pending_slot(c, KPDI_F32, n, global_start, &slot, true);
  if (rc) return rc;
  if (slot) {  // a small chunk: simulated straight into its place among the rows that will be held together
    rc = project_to_device(c, rotations, n, rescale, out_min, out_max, KPDI_F32, slot, nullptr, true);
    return rc ? rc : pending_commit(c, n, global_start, true);
  }
  HIPCHK(c->dict_raw.reserve((size_t)n * c->npix * sizeof(float)));
  rc = project_to_device(c, rotations, n, rescale, out_min, out_max, KPDI_F32, c->dict_raw.p);
  if (rc) return rc;
  return hold_chunk_dev(c, c->dict_raw.p, KPDI_F32, n, global_start, false);
}

}  // extern "C"
