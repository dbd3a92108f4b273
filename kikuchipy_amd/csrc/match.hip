// match.hip - the hot kernel: S = Y . X^T on the f32 MFMA pipe with a fused
// per-experimental-pattern top-k, so the (M x N) similarity matrix the
// reference materialises (indexing/_dictionary_indexing.py:195-198:
// einsum -> argtopk + topk) never exists.
//
// Reference semantics reproduced: SimilarityMetric.match()
// (similarity_metrics/_normalized_cross_correlation.py:161-183,
//  _normalized_dot_product.py:152-174) followed by `_match_chunk`'s
// argtopk/topk (indexing/_dictionary_indexing.py:193-203).
//
// Design (gfx950 / CDNA4)
//  * Operands are the PREPARED matrices (prep.hip): dictionary Y (n_pad x kpad)
//    and experimental X (m_pad x kpad), f32, zero-padded, both "K-major" (an NT
//    GEMM), stored tile/slab-blocked: every (128 patterns x 32 pixels) block is 16 KB
//    contiguous and already in LDS order (kernels.h: prepared_offset), so a slab is
//    one sequential 16 KB burst from HBM instead of 128 strided 128-byte rows.
//  * The dictionary is the MFMA A operand (rows of the accumulator tile), the
//    experimental patterns are the B operand (columns).  With
//    v_mfma_f32_32x32x2_f32 the accumulator column is lane&31, so every lane owns
//    ONE experimental pattern per 32x32 tile and sees 16 dictionary candidates
//    for it in its registers: top-k becomes a lane-local streaming insertion with
//    no cross-lane traffic.
//  * Workgroup = 4 waves, tile = 128 dictionary x 128 experimental patterns;
//    wave w owns experimental columns [32w, 32w+32) and all 128 dictionary rows
//    (4 accumulator tiles = 64 VGPRs).  A workgroup is persistent: it belongs to
//    one block of 128 experimental patterns, draws dictionary tiles (ascending) from
//    that block's counter and keeps its lanes' sorted best-KMAX lists in registers
//    for the whole sweep.
//  * Dictionary slabs: HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round
//    trip), two 16 KB stages, one barrier per 32-pixel slab.  Experimental slabs are
//    not shared between waves and go straight to VGPRs (see exp_base below).  The LDS image is lane-linear
//    (hardware rule), so the bank swizzle lives in the prepared layout itself
//    and is applied again on the ds_read_b128 fragment reads: 16-byte slot
//    w = ((row&1)<<3 | kq) ^ ((row>>1)&7) inside the 256-byte line of a row pair
//    -> conflict-free for the 16-lane groups of ds_read_b128.
//  * Each ds_read_b128 hands a lane 4 consecutive pixels of its row; MFMA j of a
//    group uses element j as the k-operand for both A and B, i.e. lanes 0-31
//    carry pixel j and lanes 32-63 pixel 4+j.  A and B use the same assignment,
//    which only permutes the summation order.
//  * Grid = row_blocks x nsplit with split = blockIdx % nsplit: hardware places
//    block b on XCD b%8, so (nsplit % 8 == 0) all workgroups of an XCD sweep the
//    same dictionary range at the same time and share its slabs in that XCD's L2.
//
// Algorithmic work per launch: 2 * M * n_chunk * K flops (K = kept pixels).
#include "kernels.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>

namespace kpdi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SLAB_BYTES = TILE_DICT * TILE_K * 4;  // one dictionary slab: 16 KB
constexpr int STAGE_BYTES = SLAB_BYTES;              // only the dictionary goes through LDS
constexpr int LDS_BYTES = 2 * STAGE_BYTES;           // double buffered: 32 KB per workgroup

struct MatchArgs {
  const float *dict;
  const float *exp;
  int kpad, n_tiles, n_valid, nsplit, idx_base;
  int xcd_row_groups, rows_per_group, splits_per_group;  // XCD-aware block map (0 = plain)
  float *part_scores;
  int *part_idx;
  const float *bound_score;
  const int *bound_idx;
  unsigned *tile_ctr;  // [row blocks] next dictionary tile to hand out, zero at launch
  unsigned *gthr;  // [m_pad] shared lower bound of each pattern's k-th best score (monotone key)
};

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
// new s[j] = median(s[j-1], s[j], v) because s[j-1] >= s[j].
// Branch-free: the index selects are written as bit blends (the compiler folds them to
// v_cndmask; nested ?: on the indices came out as ~20 exec-mask branches per insertion).
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
    // above entry j-1 too -> entry j-1 shifts down into j; else v lands in j (if above j)
    id[j] = blend(above[j], blend(above[j - 1], id[j - 1], idx), id[j]);
    s[j] = __builtin_amdgcn_fmed3f(s[j - 1], s[j], v);
  }
  id[0] = blend(above[0], idx, id[0]);
  s[0] = fmaxf(s[0], v);
}

template <int KMAX, bool BOUNDED>
__global__ __launch_bounds__(MATCH_THREADS, 2) void match_topk_kernel(MatchArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Block -> (experimental row block rb, dictionary split sp).  The dispatcher puts
  // block b on XCD b%8 (speed only, never correctness).  With the XCD-aware map each
  // XCD owns a (row group) x (split group) rectangle of the work: a dictionary slab is
  // then shared through that XCD's L2 by rows_per_group workgroups and an experimental
  // slab by splits_per_group workgroups, instead of (all rows) x 2.
  int sp, rb;
  if (a.xcd_row_groups > 0) {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int xr = x % a.xcd_row_groups, xs = x / a.xcd_row_groups;
    rb = xr * a.rows_per_group + j % a.rows_per_group;
    sp = xs * a.splits_per_group + j / a.rows_per_group;
  } else {
    sp = blockIdx.x % a.nsplit;
    rb = blockIdx.x / a.nsplit;
  }
  const int kpad = a.kpad;
  const int nslab = kpad / TILE_K;
  // Dictionary tiles are handed out dynamically: the nsplit workgroups of a row block
  // draw tile numbers from one counter, so all of them finish within one tile of each
  // other however unevenly the CU's two resident workgroups share the MFMA pipe (static
  // ranges left half of the workgroups idle for the last ~12 % of the launch).  Each
  // workgroup still sees ascending tile numbers, which the tie rule relies on.
  unsigned *tile_ctr = a.tile_ctr + rb;
  volatile int *ctrl = (volatile int *)(smem + LDS_BYTES);  // 4 control words behind the stages

  // ---- global -> LDS staging.  A prepared (tile, slab) block is 16 KB contiguous in
  // memory and already swizzled (kernels.h: prepared_offset), so wave wv just copies the
  // 1 KB pieces {wv, wv+4, wv+8, wv+12} of each operand's slab, lane-linear.
  const unsigned goff = (unsigned)lane * 16u;
  const size_t tile_bytes = (size_t)(kpad / TILE_K) * SLAB_BYTES;  // one 128-row tile, all slabs
  // The experimental operand is NOT shared between waves (wave wv only ever needs its own
  // 32 patterns), so it skips LDS: the prepared layout (kernels.h: prepared_exp_offset)
  // stores, per (32 patterns, slab), the four MFMA B fragments lane-linear - each one a
  // fully coalesced 1 KB global_load_dwordx4 straight into VGPRs, one slab ahead.
  const f32x4 *exp_base = (const f32x4 *)a.exp + ((size_t)rb * 4 + wv) * (size_t)nslab * 256 + lane;

  // ---- LDS -> MFMA fragments.  Lane l reads row (l&31) of a 32-row tile, pixel
  // quad kg*2 + (l>>5) of the slab.
  unsigned frag[4];
  {
    const int lr = lane & 31;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const int w = (((lane & 1) << 3) | (kg * 2 + (lane >> 5))) ^ ((lane >> 1) & 7);
      frag[kg] = (unsigned)((lr >> 1) * 256 + w * 16);
    }
  }

  // ---- per-lane running best lists for experimental pattern rb*128 + wv*32 + (lane&31)
  float best[KMAX];
  int best_idx[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    best[j] = -INFINITY;
    best_idx[j] = INT_MAX;
  }
  const int m_lane = rb * TILE_EXP + wv * 32 + (lane & 31);
  float ub = INFINITY;
  int ub_idx = -1;
  if (BOUNDED) {
    ub = a.bound_score[m_lane];
    ub_idx = a.bound_idx[m_lane];
  }
  // Shared threshold.  Every list's KMAX-th best score is a lower bound of the
  // pattern's global KMAX-th best, so the maximum over all lists (other lanes, other
  // workgroups, earlier chunks of the sweep) may be used to reject candidates: nothing
  // strictly below it can be in the final top-k.  It is only a FILTER - monotone and
  // valid however stale it is, so no ordering or coherence is required of it - and it
  // cuts the insertions per lane from ~k*ln(n_lane/k) to ~k*ln(N/k)/lists.
  unsigned gkey = 0x007fffffu;  // key(-inf)

  f32x16 acc[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;

  int tile, next_tile;
  if (tid == 0) {
    ctrl[0] = (int)atomicAdd(tile_ctr, 1u);
    ctrl[1] = (int)atomicAdd(tile_ctr, 1u);
  }
  __syncthreads();
  tile = __builtin_amdgcn_readfirstlane(ctrl[0]);
  next_tile = __builtin_amdgcn_readfirstlane(ctrl[1]);
  __syncthreads();
  if (tile >= a.n_tiles) goto write_out;

  {
    // next slab to fetch
    int ld_tile = tile, ld_slab = 0;
    int fetched = 0;  // thread 0: the tile number drawn during the current tile
    // The next slab = four 1 KB LDS-DMA pieces of the dictionary slab + the four B
    // fragments of the experimental slab.  They are issued ONE AT A TIME between MFMA
    // groups (below): the issue cycles of a memory instruction are free while an MFMA the
    // wave issued is still executing, but dead time when 8 of them sit in front of the MFMAs.
    const char *gd = nullptr;
    const f32x4 *ge = nullptr;
    f32x4 eb[4], eb_next[4];  // experimental fragments of this / the next slab
    auto next_slab = [&]() {
      gd = (const char *)a.dict + (size_t)ld_tile * tile_bytes + (size_t)ld_slab * SLAB_BYTES;
      ge = exp_base + (size_t)ld_slab * 256;
      if (++ld_slab == nslab) {
        ld_slab = 0;
        ld_tile = next_tile;  // slab 0 of the following tile is fetched during this tile's last slab
      }
    };
    auto issue_piece = [&](int stage, int c) {
      const char *g = gd + (wv + 4 * c) * 1024 + goff;
      char *l = smem + stage * STAGE_BYTES + (wv + 4 * c) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                       (__attribute__((address_space(3))) void *)l, 16, 0, 0);
    };

    next_slab();
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) issue_piece(0, pc);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) eb_next[kg] = ge[kg * 64];
    int slab = 0, stage = 0;
    for (;;) {
      // the slab of this step has landed (this wave's pieces: vmcnt; the other waves':
      // barrier) and every wave is done reading the other stage (computed on last step)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (slab == 1 && tid == 0) ctrl[2] = fetched;  // drawn at slab 0; the vmcnt above covers it
      __syncthreads();
      const bool more = slab + 1 < nslab || next_tile < a.n_tiles;
      if (more) next_slab();
      if (slab == 0 && tid == 0) fetched = (int)atomicAdd(tile_ctr, 1u);
      if (slab == nslab - 1)  // fetched now (L2, bypassing L1), consumed after this slab's MFMAs
        gkey = __hip_atomic_load(&a.gthr[m_lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

      const char *ls = smem + stage * STAGE_BYTES;
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) eb[kg] = eb_next[kg];
      // dictionary fragments of pixel group kg+1 are fetched while the 16 MFMAs of group kg run
      f32x4 fa[2][4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) fa[0][rt] = *(const f32x4 *)(ls + rt * 4096 + frag[0]);
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        const int cur = kg & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int rt = 0; rt < 4; ++rt)
            acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][rt][j], eb[kg][j], acc[rt], 0, 0, 0);
          // behind these 4 MFMAs (256 pipe cycles): next group's fragments (j = 0, 1) ...
          if (kg < 3 && j == 0) {
            fa[cur ^ 1][0] = *(const f32x4 *)(ls + 0 * 4096 + frag[kg + 1]);
            fa[cur ^ 1][1] = *(const f32x4 *)(ls + 1 * 4096 + frag[kg + 1]);
          }
          if (kg < 3 && j == 1) {
            fa[cur ^ 1][2] = *(const f32x4 *)(ls + 2 * 4096 + frag[kg + 1]);
            fa[cur ^ 1][3] = *(const f32x4 *)(ls + 3 * 4096 + frag[kg + 1]);
          }
          // ... and one eighth of the next slab (all of it goes out in the first half of the step)
          if (kg == 0 && more) issue_piece(stage ^ 1, j);
          if (kg == 1 && more) eb_next[j] = ge[j * 64];
          __builtin_amdgcn_sched_barrier(0);
        }
      }

      if (++slab == nslab) {
        // ---- epilogue: 64 candidates per lane, by increasing dictionary index.
        // The register index r is a (scalar) loop counter: the accumulator element is
        // fetched with relative VGPR addressing, so there are 4 copies of the insertion
        // code instead of 64.
        const int row0 = tile * TILE_DICT + 4 * (lane >> 5);
        const float gthr = key_score32(gkey);
        const float kth_before = best[KMAX - 1];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          // cheap screen of the 16 candidates of this accumulator tile (thresholds as of
          // now: a superset of what the exact loop below admits)
          bool any = false;
#pragma unroll
          for (int r = 0; r < 16; ++r) any = any || (acc[rt][r] >= gthr && acc[rt][r] > best[KMAX - 1]);
          if (__builtin_amdgcn_ballot_w64(any) != 0) {
#pragma unroll 1
            for (int r = 0; r < 16; ++r) {
              const int lrow = row0 + rt * 32 + (r & 3) + 8 * (r >> 2);
              const float v = acc[rt][r] + 0.f;  // -0 -> +0 so that ties compare as the merge does
              const int idx = a.idx_base + lrow;
              bool ok = lrow < a.n_valid && v >= gthr;
              if (BOUNDED) ok = ok && (v < ub || (v == ub && idx > ub_idx));
              if (ok && v > best[KMAX - 1]) list_insert<KMAX>(best, best_idx, v, idx);
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
        }
        // publish this list's KMAX-th best if it rose above the shared threshold
        if (best[KMAX - 1] > kth_before && best[KMAX - 1] > gthr)
          __hip_atomic_fetch_max(&a.gthr[m_lane], score_key(best[KMAX - 1]), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
        slab = 0;
        if (nslab == 1) {  // single-slab detectors: no later step of this tile published it
          if (tid == 0) ctrl[2] = fetched;
          __syncthreads();
        }
        tile = next_tile;
        next_tile = __builtin_amdgcn_readfirstlane(ctrl[2]);
        if (tile >= a.n_tiles) break;
      }
      stage ^= 1;
    }
  }

write_out : {
  const int m = m_lane;
  const int lists = 2 * a.nsplit;
  const size_t o = ((size_t)m * lists + (size_t)(sp * 2 + (lane >> 5))) * KMAX;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    a.part_scores[o + j] = best[j];
    a.part_idx[o + j] = best_idx[j];
  }
}
}

int match_list_len(int k) {
  if (k <= 1) return 1;
  if (k <= 8) return 8;
  if (k <= 20) return 20;
  return 32;
}

int match_blocks_per_cu() { return 2; }

template <int KMAX, bool BOUNDED>
static hipError_t launch_t(const MatchArgs &args, int grid, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)match_topk_kernel<KMAX, BOUNDED>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 16);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((match_topk_kernel<KMAX, BOUNDED>), dim3(grid), dim3(MATCH_THREADS), LDS_BYTES + 16, s, args);
  return hipGetLastError();
}

hipError_t launch_match(const MatchLaunch &a, hipStream_t s) {
  MatchArgs g;
  g.dict = a.dict;
  g.exp = a.exp;
  g.kpad = a.kpad;
  g.n_tiles = a.n_tiles;
  g.n_valid = a.n_valid;
  g.nsplit = a.nsplit;
  // choose (row groups) x (split groups) = 8 XCDs minimising rows_per_group + splits_per_group
  g.xcd_row_groups = 0;
  g.rows_per_group = g.splits_per_group = 0;
  {
    const int rbk = a.m_pad / TILE_EXP;
    int best_cost = 1 << 30;
    for (int gr = 1; gr <= 8; gr *= 2) {
      const int gs = 8 / gr;
      if (rbk % gr || a.nsplit % gs) continue;
      const int cost = rbk / gr + a.nsplit / gs;
      if (cost < best_cost) {
        best_cost = cost;
        g.xcd_row_groups = gr;
        g.rows_per_group = rbk / gr;
        g.splits_per_group = a.nsplit / gs;
      }
    }
    if (const char *e = getenv("KPDI_PLAIN_BLOCK_MAP")) {
      if (e[0] == '1') g.xcd_row_groups = 0;
    }
  }
  g.idx_base = a.idx_base;
  g.part_scores = a.part_scores;
  g.part_idx = a.part_idx;
  g.bound_score = a.bound_score;
  g.bound_idx = a.bound_idx;
  g.gthr = a.gthr;
  g.tile_ctr = a.tile_ctr;
  const int grid = (a.m_pad / TILE_EXP) * a.nsplit;
  const bool bounded = a.bound_score != nullptr;
#define KPDI_CASE(K)                                           \
  case K:                                                      \
    return bounded ? launch_t<K, true>(g, grid, s) : launch_t<K, false>(g, grid, s);
  switch (a.list_len) {
    KPDI_CASE(1)
    KPDI_CASE(8)
    KPDI_CASE(20)
    KPDI_CASE(32)
    default:
      return hipErrorInvalidValue;
  }
#undef KPDI_CASE
}

}  // namespace kpdi
