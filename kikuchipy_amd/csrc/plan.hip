// plan.hip - kpdi_plan_describe (include/kpdi.h): what the planner of plan.h decides for a sweep, as plain data.  Host
// arithmetic only: callable without a GPU (tests/test_planner.py).
#include "../../include/kpdi.h"
#include "group_hooks.h"
#include "plan.h"

extern "C" int kpdi_plan_describe(int64_t m, int64_t n_chunk, int k_kept, int keep_n, int n_cu, int form, kpdi_plan *out) {
  using namespace kpdi;
  if (!out) return fail_msg(KPDI_EINVAL, "out is NULL");
  if (m < 1 || n_chunk < 1 || k_kept < 1 || keep_n < 1 || n_cu < 1 || m >= (1ll << 31) - TILE_EXP || n_chunk >= (1ll << 31) - F16_TILE)
    return fail_msg(KPDI_EINVAL, "kpdi_plan_describe: sizes must be positive (and fit 32 bits)");
  if (form != -1 && form != 0 && form != 3) return fail_msg(KPDI_EINVAL, "form must be -1 (automatic), 0 (match.hip) or 3 (match16.hip, f32)");
  plan::Env e;
  e.n_cu = n_cu;
  e.blocks_per_cu = match_blocks_per_cu();
  e.sw.read();
  *out = kpdi_plan{};
  const int m_pad = round_up(m, TILE_EXP), row_blocks = m_pad / TILE_EXP;
  const bool wide = form < 0 ? plan::prefer_wide(e, row_blocks, k_kept, n_chunk) : form == 3;
  const int tile = wide ? F16_TILE : TILE_DICT;
  const int n_tiles = round_up(n_chunk, tile) / tile;
  int rpl = row_blocks;
  const int nsplit = plan::choose_nsplit(e, wide, row_blocks, n_tiles, &rpl);
  out->form = wide ? 3 : 0;
  out->tile = tile;
  out->row_blocks = row_blocks;
  out->n_tiles = n_tiles;
  out->nsplit = nsplit;
  out->rows_per_launch = rpl;
  out->launches = (row_blocks + rpl - 1) / rpl;
  out->round_rows = plan::round_rows(e, row_blocks);
  const bool bounded = keep_n > KMAX_LIMIT;  // (passes after the first are bounded; the first pass plans like a plain sweep)
  if (!wide) {
    out->tail_tiles = plan::classic_tail_tiles(e, n_tiles, nsplit, row_blocks <= rpl, false);
    out->n_main = n_tiles - out->tail_tiles;
    out->fixed_draws = plan::classic_fixed_draws(e, out->n_main, nsplit, out->tail_tiles);
    if (out->tail_tiles > 0) {
      out->tail_units = (int)((std::min<int64_t>(n_chunk, (int64_t)n_tiles * TILE_DICT) - (int64_t)out->n_main * TILE_DICT + 31) / 32);
      out->tail_nsplit = std::min(nsplit, out->tail_units);
    }
  } else {
    out->n_main = n_tiles;
    const double inside = plan::wide_tail(e, n_tiles, nsplit, &out->tail_shift);
    out->tail_first = n_tiles - n_tiles % nsplit;
    int rows = 0;
    if (!bounded && plan::wide_gemm_tail(e, row_blocks, n_tiles, nsplit, n_chunk, &rows) < inside && rows > 0) {
      out->tail_gemm_rows = rows;  // (sweep.hip: match_setup)
      out->n_main = out->tail_first;
      out->tail_shift = 0;
    }
    out->perm_stride = plan::tile_order_stride(e, (out->tail_shift > 0 || rows > 0) ? out->tail_first : n_tiles, nsplit, &out->perm_rounds);
  }
  for (int r0 = 0, j = 0; r0 < row_blocks; r0 += rpl, ++j) {
    if (j >= KPDI_PLAN_MAX_LAUNCHES) {
      out->n_launch_desc = KPDI_PLAN_MAX_LAUNCHES;
      return KPDI_OK;  // (`launches` says how many there are; the first KPDI_PLAN_MAX_LAUNCHES are described)
    }
    out->launch[j].row_first = r0;
    out->launch[j].rows = std::min(rpl, row_blocks - r0);
    plan::xcd_grid(e, out->launch[j].rows, nsplit, tile, wide, &out->launch[j].xcd_rows, &out->launch[j].xcd_splits,
                   &out->launch[j].rows_grid);
    out->n_launch_desc = j + 1;
  }
  return KPDI_OK;
}
