// match16.hip - the match kernel of the float16 form (KPDI_COMPUTE_F16): S = Y . X^T on the
// f16 MFMA pipe (v_mfma_f32_32x32x16_f16, float32 accumulation) with the fused per-pattern top-k
// of match.hip.  BASELINE.json configs[4] asks for exactly this ("LDS-tile retune + fp16 MFMA
// accumulate-fp32"); reference semantics as match.hip (SimilarityMetric.match + argtopk/topk,
// indexing/_dictionary_indexing.py:193-203) on operands rounded to float16.
//
// Why a kernel of its own.  A 32x32x16 f16 MFMA occupies the matrix pipe for 32 cycles, a quarter of
// the f32 instruction's 2 x 64 per 4 pixels: the f32 kernel's skeleton (one wave per SIMD, 128 x 256
// tile, lists in registers) leaves only ~1000 pipe cycles per 64-pixel slab to hide 12 LDS-DMA issues,
// 24 ds_read_b128, a barrier and the load latency, and ran at 0.31 of the 2.5 PFLOP/s peak.  Here:
//  * TWO WAVES PER SIMD (512 threads, <= 256 registers each): while one wave issues memory
//    instructions or waits at the barrier its partner's MFMAs keep the pipe busy.
//  * Workgroup tile 256 dictionary x 256 experimental patterns (2x the flops per loaded byte of the
//    128 x 256 tile); wave (wr, wc) owns rows [128 wr, +128) x columns [64 wc, +64) = 8 accumulators
//    = 128 accumulation registers.
//  * The per-lane best-k lists do NOT live in registers during the main loop: their home is a
//    coalesced scratch array; the epilogue of a tile (once per K / 16 MFMAs per accumulator) first
//    screens the accumulators against the threshold held in ONE register per list and only then loads
//    a list, inserts, and stores it back.  That is what frees the registers for two waves per SIMD.
//  * Operands (prep_device.h: half_slot): patterns in tiles of 256, pixels in steps of 48; a
//    (tile, step) block is 24 KB contiguous, stored PLANE-major: [6 planes][256 rows][8 pixels].  A
//    lane's MFMA fragment (row l & 31 of a 32-row group, pixels 8 (l >> 5) .. + 7 of a 16-pixel k-step)
//    is one 16-byte ds_read_b128 at plane * 4096 + row * 16: within every 16-lane group of the
//    instruction the rows are distinct mod 16 -> 16 distinct bank quads, conflict-free WITHOUT a swizzle,
//    and a block is copied verbatim by 24 lane-linear 1 KB LDS-DMA pieces.
//  * LDS = ring of three 48 KB stages (dictionary block + experimental block), filled two steps ahead by
//    6 pieces per wave and step; one barrier per step (after its first k-step), as in match.hip.
//
// Algorithmic work per launch: 2 * M * n_chunk * K flops (K = kept pixels).
#include "match_device.h"
#include <stdlib.h>

namespace kpdi {

constexpr int BLOCK16 = F16_TILE * F16_STEP * 2;  // one (tile, step) block: 24 KB
constexpr int STAGE16 = 2 * BLOCK16;              // dictionary block + experimental block
constexpr int NSTAGE16 = 3;
constexpr int LDS16 = NSTAGE16 * STAGE16;         // 144 KB (+ 32 B control words)
constexpr int KSTEPS16 = F16_STEP / 16;           // MFMA k-steps per step: 3

// acc += A x B for 16 pixels, A and B = 8 f16 per lane (one 16-byte LDS read); accumulator pinned to
// the accumulation registers, `s_nop 1` = the VALU-write -> MFMA-operand hazard hipcc does not pad
// inside an asm statement (match.hip: mfma_acc)
__device__ __forceinline__ void mfma16(f32x16 &c, const f32x4 &a, const f32x4 &b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// One of a wave's six 1 KB LDS-DMA pieces of a stage (i = 0..2 dictionary block, 3..5 experimental
// block); piece q = wv + 8 * (i % 3) of the block's 24.  `gd` / `ge`: wave-uniform block addresses.
__device__ __forceinline__ void issue_piece16(const char *gd, const char *ge, char *stage_base, int wv, int i,
                                              unsigned goff) {
  const bool is_exp = i >= 3;
  const int q = wv + 8 * (is_exp ? i - 3 : i);
  __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)(is_exp ? ge : gd), 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(
      rsrc, (__attribute__((address_space(3))) void *)(stage_base + (is_exp ? BLOCK16 : 0) + q * 1024), 16, (int)goff,
      q * 1024, 0, 0);
}

// One column group's 4 accumulators (128 dictionary rows x 32 patterns; 64 candidates per lane by
// increasing dictionary index) into the lane's list: first a screen with plain compares (bit r of
// `hot` = some lane of register r reaches the threshold), then only those registers go through the
// loop with the scalar register index (match.hip: scan_tile, FORM = 2).
template <int KMAX, bool BOUNDED>
__device__ __forceinline__ void scan16(f32x16 (&acc)[4], float (&best)[KMAX], int (&best_idx)[KMAX], float gthr,
                                       float ub, int ub_idx, int row0, int n_valid, int idx_base) {
  constexpr float unscale = 0x1p-24f;  // operands are stored scaled by 2^12 each
  float thr = fmaxf(gthr, next_up(best[KMAX - 1]));
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    unsigned hot = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      hot |= __builtin_amdgcn_ballot_w64(acc[rt][r] * unscale + 0.f >= thr) != 0 ? (1u << r) : 0u;
#pragma unroll 1
    while (hot != 0) {
      const int r = __builtin_ctz(hot);
      hot &= hot - 1;
      const float v = acc[rt][r] * unscale + 0.f;  // -0 -> +0 so that ties compare as the merge does
      const int lrow = row0 + rt * 32 + (r & 3) + 8 * (r >> 2);
      const int idx = idx_base + lrow;
      bool ok = lrow < n_valid && v >= thr;
      if (BOUNDED) ok = ok && (v < ub || (v == ub && idx > ub_idx));
      if (ok) {
        list_insert<KMAX>(best, best_idx, v, idx);
        thr = fmaxf(gthr, next_up(best[KMAX - 1]));
      }
    }
  }
}

// does any lane of the wave hold a candidate of this column group that reaches its threshold?
__device__ __forceinline__ bool any_candidate(const f32x16 (&acc)[4], float thr) {
  constexpr float unscale = 0x1p-24f;
  bool any = false;
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) any = any || (acc[rt][r] * unscale + 0.f >= thr);
  return __builtin_amdgcn_ballot_w64(any) != 0;
}

template <int KMAX, bool BOUNDED>
__global__ __launch_bounds__(MATCH16_THREADS, 2) void match16_kernel(MatchArgs a, float *ls_scores, int *ls_idx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7
  const int wr = wv >> 2, wc = wv & 3;
  const int sp = blockIdx.x % a.nsplit;
  const int rb = a.row_first + blockIdx.x / a.nsplit;
  const int n_tiles = a.n_tiles, n_valid = a.n_valid, idx_base = a.idx_base;
  const int nsteps = (2 * a.kpad) / F16_STEP;
  const size_t tile_bytes = (size_t)nsteps * BLOCK16;  // one 256-pattern tile, all steps
  unsigned *tile_ctr = a.tile_ctr + rb;
  volatile int *ctrl = (volatile int *)(smem + LDS16);  // control words behind the ring
  const unsigned goff = (unsigned)lane * 16u;
  const char *exp_base = (const char *)a.exp + (size_t)rb * tile_bytes;
  const char *dict_base = (const char *)a.dict;

  // LDS -> MFMA fragments: lane l reads row (l & 31) of a 32-row group, plane 2 ks + (l >> 5)
  const unsigned fa_off = (unsigned)((lane >> 5) * (F16_TILE * 16) + (wr * 128 + (lane & 31)) * 16);
  const unsigned fb_off = (unsigned)(BLOCK16 + (lane >> 5) * (F16_TILE * 16) + (wc * 64 + (lane & 31)) * 16);
#define KPDI_FA(base, rt, ks) (*(const f32x4 *)((base) + fa_off + (rt) * 512 + (ks) * (2 * F16_TILE * 16)))
#define KPDI_FB(base, cg, ks) (*(const f32x4 *)((base) + fb_off + (cg) * 512 + (ks) * (2 * F16_TILE * 16)))

  // ---- this lane's two lists (column groups 0 / 1: patterns m_lane, m_lane + 32; it sees the rows
  // 4 (lane >> 5) + {0..3} + 8 j of every 32-row group of its wave's 128 rows).  Their home is the
  // scratch: entry j of list (workgroup, wave, cg) at [((wg * 8 + wave) * 2 + cg) * KMAX + j][lane].
  const int m_lane = rb * F16_TILE + wc * 64 + (lane & 31);
  float *home_s = ls_scores + (((size_t)blockIdx.x * 8 + wv) * 2) * KMAX * 64 + lane;
  int *home_i = ls_idx + (((size_t)blockIdx.x * 8 + wv) * 2) * KMAX * 64 + lane;
#pragma unroll
  for (int j = 0; j < 2 * KMAX; ++j) {
    home_s[j * 64] = -INFINITY;
    home_i[j * 64] = INT_MAX;
  }
  float last0 = -INFINITY, last1 = -INFINITY;  // the lists' last entries: all the main loop keeps of them
  float ub0 = INFINITY, ub1 = INFINITY;
  int ubi0 = -1, ubi1 = -1;
  if (BOUNDED) {
    ub0 = a.bound_score[m_lane];
    ubi0 = a.bound_idx[m_lane];
    ub1 = a.bound_score[m_lane + 32];
    ubi1 = a.bound_idx[m_lane + 32];
  }
  const unsigned *line0 = a.gthr + (size_t)m_lane * BOUND_SLOTS;
  const unsigned *line1 = line0 + 32 * BOUND_SLOTS;
  const int list_id = sp * 4 + wr * 2 + (lane >> 5);
  const int my_slot = list_id & (BOUND_SLOTS - 1);
  const int bound_rank = a.bound_rank;
  const bool bound_grouped = a.bound_grouped != 0;
  float g0 = -INFINITY, g1 = -INFINITY;

  // ---- dictionary tiles are handed out dynamically as in match.hip: t0 = tile being computed, t1 / t2
  // the next two (loads run two steps ahead); a workgroup's first three tiles are fixed
  int t0 = sp, t1 = sp + a.nsplit, t2 = sp + 2 * a.nsplit;
  if (t0 < n_tiles) {
    const int last_tile = n_tiles - 1;
    int ld_pos = 0, ld_step = 0, ld_stage = 0;
    int fetched = 0, tp = 0;
    const char *gd = nullptr, *ge = nullptr;
#define KPDI16_CURSOR_SET()                                                          \
  {                                                                                  \
    int t_ = ld_pos == 0 ? t0 : (ld_pos == 1 ? t1 : t2);                             \
    t_ = t_ < last_tile ? t_ : last_tile; /* past the end: harmless re-load */       \
    gd = dict_base + (size_t)t_ * tile_bytes + (size_t)ld_step * BLOCK16;            \
    ge = exp_base + (size_t)ld_step * BLOCK16;                                       \
  }
#define KPDI16_CURSOR_ADVANCE()                                \
  {                                                            \
    if (++ld_step == nsteps) {                                 \
      ld_step = 0;                                             \
      ++ld_pos;                                                \
    }                                                          \
    ld_stage = ld_stage == NSTAGE16 - 1 ? 0 : ld_stage + 1;    \
  }
    // ---- prologue: steps 0 and 1 in flight, then landed and visible
    KPDI16_CURSOR_SET();
#pragma unroll
    for (int i = 0; i < 6; ++i) issue_piece16(gd, ge, smem + ld_stage * STAGE16, wv, i, goff);
    KPDI16_CURSOR_ADVANCE();
    KPDI16_CURSOR_SET();
#pragma unroll
    for (int i = 0; i < 6; ++i) issue_piece16(gd, ge, smem + ld_stage * STAGE16, wv, i, goff);
    KPDI16_CURSOR_ADVANCE();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // fragments of the three k-steps of a step, each in its own registers (static indices)
    f32x4 fa[KSTEPS16][4], fb[KSTEPS16][2];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) fa[0][rt] = KPDI_FA(smem, rt, 0);
#pragma unroll
    for (int cg = 0; cg < 2; ++cg) fb[0][cg] = KPDI_FB(smem, cg, 0);

    int stage = 0;
#pragma clang loop unroll(disable)
    for (;;) {  // dictionary tiles
      f32x16 acc0[4], acc1[4];  // column group 0 / 1
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[rt][r] = acc1[rt][r] = 0.f;
#pragma clang loop unroll(disable)
      for (int step = 0; step < nsteps; ++step) {
        const char *ls = smem + stage * STAGE16;
        const int nstage = stage == NSTAGE16 - 1 ? 0 : stage + 1;
        const char *ls_next = smem + nstage * STAGE16;
        if (step == 0 && tid == 0) fetched = (int)atomicAdd(tile_ctr, 1u);
        if (step == nsteps - 1) {  // landed by the next wait, used in the epilogue
          g0 = shared_bound<KMAX>(line0, bound_grouped);
          g1 = shared_bound<KMAX>(line1, bound_grouped);
        }
        KPDI16_CURSOR_SET();
        char *ld_base = smem + ld_stage * STAGE16;
#pragma unroll
        for (int ks = 0; ks < KSTEPS16; ++ks) {
          if (ks == 1) {
            // ---- the step's only synchronisation point: this wave's pieces of step + 1 (issued during
            // the previous step) have landed; after the barrier step + 1 is complete in LDS and every
            // wave is past the previous step, whose stage is refilled below
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (step == 0 && tid == 0) ctrl[4 + tp] = fetched;
            __syncthreads();
          }
          const int nk = ks == KSTEPS16 - 1 ? 0 : ks + 1;         // fragments read during this k-step
          const char *src = ks == KSTEPS16 - 1 ? ls_next : ls;    // ... of the next step for the last one
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) {
            mfma16(acc0[rt], fa[ks][rt], fb[ks][0]);
            mfma16(acc1[rt], fa[ks][rt], fb[ks][1]);
            // in the shadow of these MFMAs: a fragment of the next k-step and, after the barrier,
            // this wave's 6 LDS-DMA pieces of the step two ahead
            fa[nk][rt] = KPDI_FA(src, rt, nk);
            if (rt == 1) fb[nk][0] = KPDI_FB(src, 0, nk);
            if (rt == 3) fb[nk][1] = KPDI_FB(src, 1, nk);
            if (ks >= 1 && rt < 3) issue_piece16(gd, ge, ld_base, wv, (ks - 1) * 3 + rt, goff);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        KPDI16_CURSOR_ADVANCE();
        stage = nstage;
      }  // steps
      // the last MFMAs (8 passes) must have written the accumulators before they are read
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc0[rt]), "+a"(acc1[rt]));
      {
        // ---- epilogue of the tile: a list is only brought into registers when a candidate reaches it
        const int row0 = t0 * F16_TILE + wr * 128 + 4 * (lane >> 5);
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) {
          f32x16(&acc)[4] = cg == 0 ? acc0 : acc1;
          const float gthr = cg == 0 ? g0 : g1;
          float &last = cg == 0 ? last0 : last1;
          if (any_candidate(acc, fmaxf(gthr, next_up(last)))) {
            float best[KMAX];
            int bidx[KMAX];
            float *hs = home_s + cg * KMAX * 64;
            int *hi = home_i + cg * KMAX * 64;
#pragma unroll
            for (int j = 0; j < KMAX; ++j) {
              best[j] = hs[j * 64];
              bidx[j] = hi[j * 64];
            }
            float pub = best[0];  // entry bound_rank - 1 before the scan
#pragma unroll
            for (int j = 1; j < KMAX; ++j) pub = j == bound_rank - 1 ? best[j] : pub;
            scan16<KMAX, BOUNDED>(acc, best, bidx, gthr, cg == 0 ? ub0 : ub1, cg == 0 ? ubi0 : ubi1, row0, n_valid,
                                  idx_base);
#pragma unroll
            for (int j = 0; j < KMAX; ++j) {
              hs[j * 64] = best[j];
              hi[j * 64] = bidx[j];
            }
            last = best[KMAX - 1];
            float now = best[0];
#pragma unroll
            for (int j = 1; j < KMAX; ++j) now = j == bound_rank - 1 ? best[j] : now;
            if (now > pub)  // publish the list entry the shared bound is built from, if it rose
              __hip_atomic_fetch_max(const_cast<unsigned *>(cg == 0 ? line0 : line1) + my_slot, score_key(now),
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        t0 = t1;
        t1 = t2;
        t2 = __builtin_amdgcn_readfirstlane(ctrl[4 + tp]);  // published at this tile's first barrier
        tp ^= 1;
        --ld_pos;
        if (t0 >= n_tiles) break;
      }
    }
  }

  // ---- lists -> [m_pad][4 * nsplit][KMAX] for the merge kernel
  {
    const int lists = 4 * a.nsplit;
#pragma unroll
    for (int cg = 0; cg < 2; ++cg) {
      const size_t o = ((size_t)(m_lane + 32 * cg) * lists + (size_t)list_id) * KMAX;
#pragma unroll
      for (int j = 0; j < KMAX; ++j) {
        a.part_scores[o + j] = home_s[(cg * KMAX + j) * 64];
        a.part_idx[o + j] = home_i[(cg * KMAX + j) * 64];
      }
    }
  }
}

size_t match16_scratch_bytes(int grid, int list_len) { return (size_t)grid * 8 * 2 * list_len * 64 * sizeof(float); }

template <int KMAX, bool BOUNDED>
static hipError_t launch16_t(const MatchArgs &args, int grid, void *scratch, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)match16_kernel<KMAX, BOUNDED>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS16 + 32);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  // scratch: scores of all lists, then their indices
  float *ls = (float *)scratch;
  int *li = (int *)((char *)scratch + match16_scratch_bytes(grid, KMAX));
  hipLaunchKernelGGL((match16_kernel<KMAX, BOUNDED>), dim3(grid), dim3(MATCH16_THREADS), LDS16 + 32, s, args, ls, li);
  return hipGetLastError();
}

hipError_t launch_match16(const MatchLaunch &a, void *list_scratch, hipStream_t s) {
  if (a.operand_form != 2 || a.row_tiles != 4 || !list_scratch) return hipErrorInvalidValue;
  MatchArgs g;
  g.dict = a.dict;
  g.exp = a.exp;
  g.kpad = a.kpad;
  g.n_tiles = a.n_tiles;
  g.n_valid = a.n_valid;
  g.nsplit = a.nsplit;
  g.idx_base = a.idx_base;
  g.row_first = a.row_first;
  g.row_base = 0;
  g.part_scores = a.part_scores;
  g.part_idx = a.part_idx;
  g.bound_score = a.bound_score;
  g.bound_idx = a.bound_idx;
  g.gthr = a.gthr;
  g.bound_rank = a.bound_rank;
  g.bound_grouped = a.bound_grouped;
  g.tile_ctr = a.tile_ctr;
  g.tile_groups = 1;
  const int grid = a.rows * a.nsplit;
  const bool bounded = a.bound_score != nullptr;
  switch (a.list_len) {
    case 1: return bounded ? launch16_t<1, true>(g, grid, list_scratch, s) : launch16_t<1, false>(g, grid, list_scratch, s);
    case 8: return bounded ? launch16_t<8, true>(g, grid, list_scratch, s) : launch16_t<8, false>(g, grid, list_scratch, s);
    case 20:
      return bounded ? launch16_t<20, true>(g, grid, list_scratch, s) : launch16_t<20, false>(g, grid, list_scratch, s);
    case 32:
      return bounded ? launch16_t<32, true>(g, grid, list_scratch, s) : launch16_t<32, false>(g, grid, list_scratch, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace kpdi
