// match_device.h - device helpers shared by the match kernels (match.hip: f32 / split-f16 forms,
// match16.hip: the float16 form): kernel arguments, the shared rejection bound, order-preserving
// score keys and the branch-free sorted-list insertion.
#pragma once
#include "kernels.h"
#include <limits.h>
#include <math.h>

namespace kpdi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float key_score32(unsigned u);

struct MatchArgs {
  const float *dict;
  const float *exp;
  int kpad, n_tiles, n_valid, nsplit, idx_base, row_first;
  int row_base;  // ROWT = 1 (32-row tail units): dictionary row of unit 0, a multiple of 32
  float *part_scores;
  int *part_idx;
  int *part_cnt;       // match16.hip: [m_pad][lists] entries written per list (kernels.h: MatchLaunch.part_cnt)
  const float *bound_score;
  const int *bound_idx;
  unsigned *tile_ctr;  // [row blocks] next tile to hand out dynamically; fixed_draws * nsplit at launch
  int tile_groups;     // (unused: 1)
  // A workgroup's first `fixed_draws` tiles are fixed (split sp: sp, sp + nsplit, ...), the rest is drawn
  // from the row block's counter.  xcd_rows x xcd_splits = 8 arranges the 8 XCDs as a grid over the launch's
  // (row block, split) workgroups (block_rb_sp below); xcd_rows = 0: block b -> (b / nsplit, b % nsplit).
  int fixed_draws, xcd_rows, xcd_splits, rows, rows_grid;  // rows_grid >= rows: see block_rb_sp
  // match16.hip, float32 form: the tiles [tail_first, n_tiles) are handed out as units of 256 >> tail_shift rows
  // (tail_shift = 1, 2; 0 = whole tiles only): the last round of a launch then costs a half / a quarter of a tile-time
  int tail_first, tail_shift;
  unsigned *gthr;      // [m_pad][BOUND_SLOTS] published list ranks (monotone keys), see shared_bound()
  int bound_rank;      // which entry (1-based) of its list a workgroup lane publishes
  int bound_grouped;   // 1: all 32 slots are in use and bound_rank == 1 (grouped form)
  // match16.hip: ORDER of a workgroup's tiles.  Round j < perm_rounds of split sp takes tile sp + nsplit * ((j *
  // perm_stride) mod perm_rounds) - a low-discrepancy walk over the dictionary (perm_stride ~ 0.618 perm_rounds, coprime;
  // plan.h: tile_order_stride) - the rounds from perm_rounds on (partial units, an incomplete last round) follow in
  // natural order.  perm_rounds = 0: natural order throughout.
  int perm_rounds, perm_stride;
  unsigned long long *epi_stats;  // nullptr, or 4 counters of what the epilogues did (kpdi_counters.epi_*)
};

// Block id -> (row block of the launch, dictionary split).  Block b runs on XCD b % 8 (observed, used for
// speed only); the XCDs tile the (rows x nsplit) grid of workgroups as xcd_rows x xcd_splits rectangles, so
// that an XCD hosts (rows / xcd_rows) row blocks x (nsplit / xcd_splits) splits: its L2 then serves a
// dictionary tile to all its row blocks and an experimental slab to all its splits - with fixed draws and
// workgroups advancing at the same pace the operands cross the fabric once per XCD, not once per workgroup.
// The grid is laid over rows_grid >= rows row blocks (rows rounded up to a multiple of xcd_rows, so that a launch of,
// say, 29 row blocks keeps its rectangles): a workgroup whose row block is >= rows has nothing to do and returns.
__device__ __forceinline__ void block_rb_sp(const MatchArgs &a, int b, int *rb, int *sp) {
  if (a.xcd_rows == 0) {
    *sp = b % a.nsplit;
    *rb = b / a.nsplit;
    return;
  }
  const int x = b & 7, j = b >> 3;
  const int rx = a.rows_grid / a.xcd_rows, sx = a.nsplit / a.xcd_splits;
  // (a padded grid numbers its row blocks across the XCD row groups, so that the missing ones - the highest numbers -
  // fall into different groups: 29 of 32 row blocks leave three groups of XCDs with 7 of 8 row blocks each instead of
  // one group with 5 of 8)
  *rb = a.rows_grid != a.rows ? (j % rx) * a.xcd_rows + x / a.xcd_splits : (x / a.xcd_splits) * rx + j % rx;
  *sp = (x % a.xcd_splits) * sx + j / rx;
}

// ---- shared rejection bound -------------------------------------------------------------
// A score T may be used to reject candidates (v < T cannot enter the final top-KMAX) whenever
// at least KMAX candidates >= T are known to exist.  Lists cover disjoint candidates, and a
// list whose j-th best is b holds j candidates >= b.  Every list publishes its j-th best
// (key, atomic max) into slot (list % 32) of its pattern's 128-byte line; then
//   * plain form   (fewer than 32 lists, j = ceil(KMAX / lists)): T = min over the used slots
//     (j * lists >= KMAX candidates); unused slots hold the key of +inf;
//   * grouped form (>= 32 lists, j = 1): the slots are split into >= KMAX groups; every group
//     maximum is one list's best, so T = min over groups of the group maximum is backed by
//     >= KMAX distinct candidates.  This is far tighter than any single list's KMAX-th best:
//     it sits near the true global KMAX-th best instead of ~lists*KMAX places below it.
// The bound is only a FILTER: monotone and valid however stale, no ordering or coherence
// needed, and it persists across the chunks of a sweep.
template <int KMAX>
__device__ __forceinline__ float shared_bound(const unsigned *line, bool grouped) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  unsigned k[BOUND_SLOTS];
#pragma unroll
  for (int c = 0; c < BOUND_SLOTS / 4; ++c) {
    const u32x4 q = __builtin_nontemporal_load((const u32x4 *)line + c);  // L2, not the stale L1
#pragma unroll
    for (int e = 0; e < 4; ++e) k[4 * c + e] = q[e];
  }
  unsigned t = 0xffffffffu;
  if (grouped) {
    // G groups, G = smallest supported count >= KMAX: 1, 8, 20 (12 pairs + 8 singles), 32
    constexpr int G = KMAX <= 1 ? 1 : (KMAX <= 8 ? 8 : (KMAX <= 20 ? 20 : 32));
    if (G == 20) {
#pragma unroll
      for (int g = 0; g < 12; ++g) t = min(t, max(k[2 * g], k[2 * g + 1]));
#pragma unroll
      for (int i = 24; i < 32; ++i) t = min(t, k[i]);
    } else {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        unsigned m = 0;
#pragma unroll
        for (int i = g; i < BOUND_SLOTS; i += G) m = max(m, k[i]);
        t = min(t, m);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < BOUND_SLOTS; ++i) t = min(t, k[i]);
  }
  return key_score32(t);
}

__device__ __forceinline__ float key_score32(unsigned u);
// The same bound with HALF the loads per lane, for wave layouts in which lanes l and l + 32 look at the same
// pattern (match16.hip): lane half h loads slots [16 h, 16 h + 16), reduces its groups, and the two halves are
// combined across the wave (ds_bpermute, no LDS memory).  bound_load_half() only issues the loads - several
// lines can be in flight before the first bound_reduce_half() waits for one.
struct BoundHalf {
  unsigned k[BOUND_SLOTS / 2];
};
__device__ __forceinline__ void bound_load_half(BoundHalf &r, const unsigned *line, int half) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int c = 0; c < BOUND_SLOTS / 8; ++c) {
    const u32x4 q = __builtin_nontemporal_load((const u32x4 *)line + half * (BOUND_SLOTS / 8) + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) r.k[4 * c + e] = q[e];
  }
}
template <int KMAX>
__device__ __forceinline__ float bound_reduce_half(const BoundHalf &r, bool grouped, int half) {
  constexpr int G = KMAX <= 1 ? 1 : (KMAX <= 8 ? 8 : (KMAX <= 20 ? 20 : 32));
  constexpr int H = BOUND_SLOTS / 2;
  unsigned t;
  if (grouped && G == 1) {  // one group: the maximum of everything
    t = 0;
#pragma unroll
    for (int i = 0; i < H; ++i) t = max(t, r.k[i]);
    t = max(t, (unsigned)__shfl_xor((int)t, 32, 64));
  } else if (grouped && G == 8) {  // group g = slots g, g + 8 (this half), g + 16, g + 24 (the other half)
    unsigned m[8];
    t = 0xffffffffu;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      m[g] = max(r.k[g], r.k[g + 8]);
      m[g] = max(m[g], (unsigned)__shfl_xor((int)m[g], 32, 64));
      t = min(t, m[g]);
    }
  } else {
    t = 0xffffffffu;
    if (grouped && G == 20) {  // 12 pairs (slots 0 .. 23) + 8 singles (24 .. 31): every group lies in one half
#pragma unroll
      for (int g = 0; g < H / 2; ++g) {
        const unsigned pair = max(r.k[2 * g], r.k[2 * g + 1]);
        const unsigned singles = min(r.k[2 * g], r.k[2 * g + 1]);
        t = min(t, (half == 1 && g >= 4) ? singles : pair);
      }
    } else {  // plain form, or 32 groups of one slot
#pragma unroll
      for (int i = 0; i < H; ++i) t = min(t, r.k[i]);
    }
    t = min(t, (unsigned)__shfl_xor((int)t, 32, 64));
  }
  return key_score32(t);
}

// float <-> unsigned key, order preserving (same map as merge.hip)
__device__ __forceinline__ unsigned score_key(float s) {
  const unsigned u = __float_as_uint(s);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_score32(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
}

// Insert (v, idx) into a descending sorted list; precondition v > s[KMAX-1].
// Equal scores keep arrival order (candidates arrive by increasing dictionary
// index), which is the engine's tie rule: lower dictionary index first.
// new s[j] = median(s[j-1], s[j], v) because s[j-1] >= s[j].  Branch-free: the index
// selects are bit blends the compiler folds to v_cndmask (nested ?: came out as ~20
// exec-mask branches per insertion).
__device__ __forceinline__ int blend(int mask, int if_set, int if_clear) {
  return (if_set & mask) | (if_clear & ~mask);
}

template <int KMAX>
__device__ __forceinline__ void list_insert(float (&s)[KMAX], int (&id)[KMAX], float v, int idx) {
  int above[KMAX];  // all ones where v ranks above entry j (monotone in j: 0..0 1..1)
#pragma unroll
  for (int j = 0; j < KMAX; ++j) above[j] = (v > s[j]) ? -1 : 0;
#pragma unroll
  for (int j = KMAX - 1; j >= 1; --j) {
    id[j] = blend(above[j], blend(above[j - 1], id[j - 1], idx), id[j]);
    s[j] = __builtin_amdgcn_fmed3f(s[j - 1], s[j], v);
  }
  id[0] = blend(above[0], idx, id[0]);
  s[0] = fmaxf(s[0], v);
}

// Does candidate (v, idx) rank before (s, id)?  Score descending, dictionary index ascending: the engine's total order.
__device__ __forceinline__ bool ranks_before(float v, int idx, float s, int id) { return v > s || (v == s && idx < id); }

// list_insert for candidates that arrive in ANY order of dictionary index (match16.hip walks the dictionary's tiles in
// a permuted order): the position is decided by (score, index), so equal scores end up lower index first whatever the
// arrival order.  Precondition: ranks_before(v, idx, s[KMAX - 1], id[KMAX - 1]).
template <int KMAX>
__device__ __forceinline__ void list_insert_lex(float (&s)[KMAX], int (&id)[KMAX], float v, int idx) {
  int above[KMAX];  // all ones where (v, idx) ranks before entry j (monotone in j: 0..0 1..1)
#pragma unroll
  for (int j = 0; j < KMAX; ++j) above[j] = ranks_before(v, idx, s[j], id[j]) ? -1 : 0;
#pragma unroll
  for (int j = KMAX - 1; j >= 1; --j) {
    id[j] = blend(above[j], blend(above[j - 1], id[j - 1], idx), id[j]);
    s[j] = __builtin_amdgcn_fmed3f(s[j - 1], s[j], v);
  }
  id[0] = blend(above[0], idx, id[0]);
  s[0] = fmaxf(s[0], v);
}

__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += (unsigned)__shfl_xor((int)v, o, 64);
  return v;
}

// smallest float above f (f finite or -inf)
__device__ __forceinline__ float next_up(float f) {
  const unsigned u = __float_as_uint(f);
  if (f == 0.f) return __uint_as_float(1u);
  return __uint_as_float((u & 0x80000000u) ? u - 1u : u + 1u);
}


}  // namespace kpdi
