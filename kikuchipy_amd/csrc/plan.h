// plan.h - the launch planner of a sweep: pure functions of the problem (experimental patterns M, dictionary patterns of
// the chunk n, kept pixels K, keep_n), the device (compute units) and the developer switches - no HIP call, no context.
// sweep.hip asks it how to lay a chunk on the chip; kpdi_plan_describe (include/kpdi.h) exports the answer so that
// tests/test_planner.py can check the invariants on a machine without a GPU: every row block covered exactly once,
// padded grids at most 9/8 of the rows, no launch without tiles, for shares of 1 ... 300 000 patterns.
//
// Units of the cost model: the time of one 128-pattern dictionary tile of match.hip against one 256-pattern row block
// (form_model.h holds the fitted constants).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "form_model.h"
#include "kernels.h"

namespace kpdi {

// Developer / A-B switches: ONE convention - read from the environment by kpdi_set_problem and
// fixed for the context until the next kpdi_set_problem (a variable changed in between is seen then, never mid-sweep).
struct Switches {
  double odd_wide = FORM_ODD_SPLIT_WIDE, odd_classic = FORM_ODD_SPLIT_CLASSIC;
  double wide_launch = FORM_WIDE_LAUNCH, fixed_frac = 0.8;
  bool xcd_grid = true, xcd_pad = true, no_tail = false, one_stream = false, tail_stream2 = false;
  bool f64_statistical = false, f64_sync = false, no_coalesce = false, no_direct_upload = false, no_coalesce_wait = false;
  bool natural_order = false;
  int tail_gemm = -1;  // KPDI_TAIL_GEMM: 0 never, 1 whenever possible, else by cost
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
    if (const char *e = getenv("KPDI_F64_EPS")) f64_statistical = !strcmp(e, "statistical");
    f64_sync = getenv("KPDI_F64_SYNC") != nullptr;
    no_coalesce = getenv("KPDI_NO_COALESCE") != nullptr;
    no_direct_upload = getenv("KPDI_NO_DIRECT_UPLOAD") != nullptr;
    no_coalesce_wait = getenv("KPDI_NO_COALESCE_WAIT") != nullptr;
    if (const char *e = getenv("KPDI_UPLOAD_TILES")) upload_tiles = atol(e);
    if (const char *e = getenv("KPDI_TILE_ORDER")) natural_order = !strcmp(e, "natural");
    if (const char *e = getenv("KPDI_TAIL_GEMM")) tail_gemm = atoi(e);
  }
};

namespace plan {

// what the planner knows about the device and the build
struct Env {
  int n_cu = 256;          // compute units (one persistent workgroup each)
  int blocks_per_cu = 1;   // workgroups a CU holds (match_blocks_per_cu())
  Switches sw;
};

// How a sweep of `row_blocks` x `n_tiles` tile pairs is laid on the CUs (one persistent workgroup per CU): `nsplit`
// workgroups share the dictionary tiles of a row block, and a launch covers as many row blocks as fit the chip; larger
// experimental sets take several launches.  The plan minimises the makespan counted in tiles: launches *
// ceil(n_tiles / nsplit), plus a small per-launch cost.  `wide`: the kernels of match16.hip (static hand-out) pay more
// for splits that are not a multiple of 8 - such a launch has no XCD grid (profiles/r03_form_choice.json).
inline int choose_nsplit(const Env &e, bool wide, int row_blocks, int n_tiles, int *rows_per_launch) {
  const int cap = e.n_cu * e.blocks_per_cu;
  const double odd = wide ? e.sw.odd_wide : e.sw.odd_classic;
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
inline void xcd_grid(const Env &e, int rows, int nsplit, int tile_dict, bool may_pad, int *xr, int *xs, int *rows_grid) {
  *xr = *xs = 0;
  *rows_grid = rows;
  if (!e.sw.xcd_grid) return;
  long best = -1;
  for (int r = 1; r <= 8; r *= 2) {
    const int sgrid = 8 / r;
    const int rows_pad = (rows + r - 1) / r * r;
    if (nsplit % sgrid != 0) continue;
    if (rows_pad != rows && (!may_pad || !e.sw.xcd_pad || 8 * (rows_pad - rows) > rows)) continue;
    const long cost = (long)(rows_pad / r) * TILE_EXP + (long)(nsplit / sgrid) * tile_dict;
    if (best < 0 || cost < best) {
      best = cost;
      *xr = r;
      *xs = sgrid;
      *rows_grid = rows_pad;
    }
  }
}

// match16.hip, float32 form: how the last n_tiles % nsplit tiles of a launch are handed out - as whole tiles (one more
// round, shift 0) or as halves / quarters of a tile (a half / a quarter of a round each, at ~1.1 / 1.25 of the time per
// row because the experimental fragments are reused by fewer row groups).  Returns the cost of that last round in
// tile-times.
inline double wide_tail(const Env &e, int n_tiles, int nsplit, int *shift) {
  *shift = 0;
  const int left = n_tiles % nsplit;
  if (left == 0) return 0.0;
  double best = 1.0;
  if (!e.sw.no_tail)
    for (int sh = 1; sh <= 2; ++sh) {
      const double cost = (double)(((left << sh) + nsplit - 1) / nsplit) / (1 << sh) * (sh == 1 ? FORM_WIDE_HALF : FORM_WIDE_QUARTER);
      if (cost < best - 1e-9) best = cost, *shift = sh;
    }
  return best;
}

// match16.hip, float32 form: the last partial round as a kernel of its own (tailgemm.hip: 32 rows x 128 experimental
// patterns per workgroup, deep load pipeline, scores to a small matrix, a select pass behind it) instead of partial units
// inside the main kernel.  Worth it when few tiles are left per row block - one tile per 16 workgroups costs 0.31
// tile-times as a quarter-unit round and ~0.1 here.  Returns its cost in tile-times (1e30: not possible) for the
// dictionary rows [tail_first * 256, n_chunk); KPDI_TAIL_GEMM=0 / 1 forces it off / on.
inline double wide_gemm_tail(const Env &e, int row_blocks, int n_tiles, int nsplit, int64_t n_chunk, int *rows) {
  *rows = 0;
  const int left = n_tiles % nsplit;
  if (left == 0 || e.sw.tail_gemm == 0 || e.sw.no_tail) return 1e30;
  const int64_t r = n_chunk - (int64_t)(n_tiles - left) * F16_TILE;
  if (r <= 0 || r * row_blocks * TILE_EXP > (64ll << 20)) return 1e30;  // (the score matrix: at most 256 MB)
  const int64_t wgs = (r + 31) / 32 * row_blocks * 2;
  const int64_t rounds = (wgs + e.n_cu - 1) / e.n_cu;
  *rows = (int)r;
  return e.sw.tail_gemm == 1 ? 0.0 : rounds * FORM_TAIL_GEMM_UNIT + FORM_TAIL_GEMM_LAUNCH;
}

// match16.hip: the order in which a workgroup walks its whole-tile rounds (match_device.h: MatchArgs.perm_*): round j
// takes round-slot (j * stride) mod rounds, stride ~ 0.618 rounds and coprime to it - the first visits are spread over
// the whole dictionary whichever way it is ordered, so a dictionary in sampler order (or sorted by score) costs the
// fused top-k O(log rounds) "everything in this tile is a candidate" tiles instead of one per round.  Returns the
// stride; *rounds = 0 (natural order) when there are fewer than 3 whole rounds or KPDI_TILE_ORDER=natural.
inline int tile_order_stride(const Env &e, int whole_tiles, int nsplit, int *rounds) {
  *rounds = 0;
  const int r = nsplit > 0 ? whole_tiles / nsplit : 0;
  if (r < 3 || e.sw.natural_order) return 1;
  auto gcd = [](int a, int b) {
    while (b) {
      const int t = a % b;
      a = b;
      b = t;
    }
    return a;
  };
  // among the strides in [r / 2, 3 r / 4] coprime to r: the one whose walk 0, s, 2 s, ... (mod r) sets the fewest running
  // maxima - what a dictionary sorted by rising score shows a workgroup (a falling one shows its best tile first) - and,
  // among equals, the one closest to the golden section (an integer stride near 0.618 r can degenerate: r = 1184, s =
  // 733 sets 59 maxima, the best stride 11).  r <= 2048 is searched; beyond, the golden stride is taken as it is.
  const double golden = 0.6180339887 * r;
  int best = 1, best_records = 1 << 30;
  double best_dist = 1e30;
  for (int s = std::max(1, r / 2); s <= std::min(r - 1, 3 * r / 4 + 1); ++s) {
    if (gcd(s, r) != 1) continue;
    int records = 0;
    if (r <= 2048) {
      records = 1;
      for (int j = 1, p = 0, mx = 0; j < r; ++j) {
        p += s;
        if (p >= r) p -= r;
        if (p > mx) mx = p, ++records;
      }
    }
    const double dist = std::abs(s - golden);
    if (records < best_records || (records == best_records && dist < best_dist)) best = s, best_records = records, best_dist = dist;
  }
  if (best_records == (1 << 30)) return 1;  // (no coprime stride in the window: r = 4 -> 3 is found, r = 3 -> 2; never here)
  *rounds = r;
  return best;
}

// match.hip: when the dictionary tiles of a row block are a small non-multiple of nsplit (a rank's share of a sharded
// dictionary: 98 tiles over 16 workgroups) whole tiles would leave most workgroups idle during the last round
// (makespan 7 tile-times for 6.1 of work).  The last n_tiles % nsplit tiles are then handed out as QUARTER tiles by a
// second launch of the kernel's 32-row form.  Returns how many tiles that tail launch takes (0 = none).
inline int classic_tail_tiles(const Env &e, int n_tiles, int nsplit, bool single_launch, bool bounded) {
  if (!single_launch || bounded || e.sw.no_tail) return 0;
  const int rounds = n_tiles / nsplit, rem = n_tiles % nsplit;
  return (rounds >= 1 && rounds < 32 && rem > 0 && 4 * rem <= 3 * nsplit) ? rem : 0;
}

// Tile hand-out of match.hip: a workgroup's first `fixed_draws` tiles are fixed (sp, sp + nsplit, ...) so that the
// workgroups sharing an XCD stream the same operands at the same pace (the XCD's L2 then serves them: 26 -> ~12 GB
// crossing the fabric per config-2 launch); the last ~20 % are drawn from the row block's counter, which evens out the
// speeds at the end (all tiles fixed left CUs idle for the last ~10 % of the launch).  KPDI_FIXED_FRAC overrides.
inline int classic_fixed_draws(const Env &e, int n_main, int nsplit, int tail_tiles) {
  const double frac = e.sw.fixed_frac;
  const int per_wg = n_main / nsplit;
  return (tail_tiles > 0 || n_main % nsplit == 0) && frac > 0 ? per_wg + 1 : std::max(3, (int)(frac * per_wg));
}

// Which f32 match kernel serves a sweep whose chunks hold n_chunk patterns.  match16.hip's one-wave-per-SIMD form
// (256 x 256 tiles, lists out of the registers) does a unit of work 3 % faster than match.hip (128 x 256 tiles) but
// hands out whole 256-pattern tiles statically, match.hip 128-pattern tiles with a dynamic tail of quarter tiles: the
// estimated makespans decide.  Measured again in round 5 (profiles/r05_one_kernel.txt): forcing the wide kernel
// everywhere costs one rank's share of configs[1] at N = 8 (12 500 patterns: 3 tile rounds) 3.8 % - more than the 2 %
// that would retire match.hip - and nothing elsewhere.
inline bool prefer_wide(const Env &e, int row_blocks, int k_kept, int64_t n_chunk) {
  int rpl = 0;
  // match.hip: whole rounds of 128-pattern tiles + (when few rounds) a quarter-tile tail launch
  const int t128 = (int)((n_chunk + 127) / 128);
  const int ns = choose_nsplit(e, false, row_blocks, t128, &rpl);
  const int launches = (row_blocks + rpl - 1) / rpl;
  double classic = (t128 + ns - 1) / ns;
  if (launches == 1) {
    const int rounds = t128 / ns, rem = t128 % ns;
    if (rounds >= 1 && rounds < 32 && rem > 0 && 4 * rem <= 3 * ns)
      classic = rounds + ((4 * rem + ns - 1) / ns) * 0.25 + FORM_CLASSIC_TAIL;
  }
  classic = launches * (classic + FORM_CLASSIC_LAUNCH);  // + ~0.1 ms per launch
  // match16.hip, float32 form: whole rounds of 256-pattern tiles, two 128-tile units each at 1 / 1.03 of the time
  const int t256 = (int)((n_chunk + 255) / 256);
  const int nsw = choose_nsplit(e, true, row_blocks, t256, &rpl);
  int shift = 0;
  // (its launch costs more: the first tile's 64 candidates per lane go to the buffers, the lists are built at the end -
  // 0.27 ms against 0.1 ms, measured on one rank's share of configs[1] at N = 8)
  // (fitted between K = 2819 and 14 400: no extrapolation below)
  const double gain = FORM_WIDE_GAIN + FORM_WIDE_GAIN_K * std::max(-0.3, 1.0 - 3600.0 / std::max(k_kept, 1));
  int gemm_rows = 0;
  const double tail = std::min(wide_tail(e, t256, nsw, &shift), wide_gemm_tail(e, row_blocks, t256, nsw, n_chunk, &gemm_rows));
  const int launches_w = (row_blocks + rpl - 1) / rpl;
  // (a sweep of ONE launch pays the wide kernel's launch once and nothing else; the constant of sweeps of several launches
  // also carries what their last, partly filled launch loses - form_model.h)
  const double launch = launches_w == 1 && e.sw.wide_launch == FORM_WIDE_LAUNCH ? FORM_WIDE_LAUNCH_SINGLE : e.sw.wide_launch;
  const double wide = launches_w * ((t256 / nsw + tail) * 2.0 / gain + launch) * (nsw % 8 == 0 ? 1.0 : FORM_WIDE_ODD);
  return wide < classic;
}

// How a host chunk is cut for the upload/sweep pipeline: uniform pieces (the remainder last), each one match launch
// set.  Short pieces bound the two ends that do not overlap (the upload of the first piece, the sweep of the last);
// long pieces waste less on whole tiles per workgroup and on the per-piece prep/merge.  Which wins depends on whether
// the job is upload-bound (few experimental patterns) or sweep-bound (many), so the size is picked by playing the
// two-stage pipeline through for every candidate with the launch plan's own cost model: upload at ~56 GB/s (pageable
// memory over PCIe 5, measured), one workgroup-tile (256 x 128 x kpad MACs) at 88 % of a CU's f32 MFMA rate, 0.18 ms
// per piece of fixed work (launch ramp, prep, merge; fitted).  The model lands within ~0.3 ms of the measurements at
// config 2 (tools/pcie_probe.py; upload alone 25.5 ms): 32-tile pieces 27.2 ms, 48 tiles 27.5 ms (about the model's
// pick), 96 tiles 28.7 ms, 192 tiles 31.3 ms; with 10 000 experimental patterns it picks 128 tiles (60.7 ms; 64 tiles
// 63.1 ms, 256 tiles 65.5 ms).  KPDI_UPLOAD_TILES=<tiles per piece> overrides.  `row_blocks` = 0: no experimental set
// yet (one piece).  Returns the patterns of every piece.
inline std::vector<int64_t> upload_pieces(const Env &e, bool wide, int row_blocks, int kpad, int64_t n_chunk, size_t row_bytes) {
  const int64_t tiles = (n_chunk + TILE_DICT - 1) / TILE_DICT;
  int64_t piece = tiles;
  if (e.sw.upload_tiles > 0) {
    piece = e.sw.upload_tiles;
  } else if (row_blocks > 0) {
    const double t_tile = 2.0 * TILE_EXP * TILE_DICT * kpad / (157.3e12 / 256 * 0.88);
    const double t_row = row_bytes / 56e9, t_fixed = 0.18e-3;
    auto sweep_time = [&](int64_t t) {
      int rpl = 0;
      const int ns = choose_nsplit(e, wide, row_blocks, (int)t, &rpl);
      return ((row_blocks + rpl - 1) / rpl) * (double)((t + ns - 1) / ns) * t_tile + t_fixed;
    };
    double best = 1e30;
    for (int64_t cand = 32; cand <= 512 + 8; cand += 8) {
      const int64_t p = cand > 512 ? tiles : std::min(cand, tiles);  // last candidate: one piece
      const double s_full = sweep_time(p), s_rest = tiles % p ? sweep_time(tiles % p) : 0.0;
      double copied = 0, swept = 0;
      for (int64_t left = tiles; left > 0; left -= p) {
        const int64_t t = std::min(p, left);
        copied += t * TILE_DICT * t_row;
        swept = std::max(swept, copied) + (t == p ? s_full : s_rest);
      }
      if (swept < best - 1e-9) best = swept, piece = p;
    }
  }
  if (wide) piece = (piece + 1) / 2 * 2;  // whole 256-pattern tiles of match16.hip
  std::vector<int64_t> out;
  for (int64_t left = n_chunk, per = piece * TILE_DICT; left > 0; left -= per) out.push_back(std::min(per, left));
  return out;
}

// dictionary patterns one full round of a sweep covers (every CU one tile of 256): below it a chunk wastes most of its
// launch; chunks are coalesced (sweep.hip) and group pieces sized (group.hip) in this unit
inline int64_t round_rows(const Env &e, int row_blocks) {
  const int per_row_block = std::max(1, e.n_cu * e.blocks_per_cu / std::max(row_blocks, 1));
  return (int64_t)per_row_block * F16_TILE;
}

}  // namespace plan
}  // namespace kpdi
