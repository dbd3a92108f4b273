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
//    (tile, step) block is 24 KB contiguous, stored PLANE-major: [3 k-steps][256 rows][16 pixels].  A
//    lane's MFMA fragment (row l & 31 of a 32-row group, pixels 8 (l >> 5) .. + 7 of the k-step) is one
//    16-byte ds_read_b128 at plane * 8192 + row * 32 + half * 16, the two halves swapped for rows with
//    bit 3 set: within every 16-lane group of the instruction the 16 reads hit 16 distinct bank quads
//    (SQ_LDS_BANK_CONFLICT = 0), and a block is copied verbatim by 24 lane-linear 1 KB LDS-DMA pieces.
//  * LDS = ring of three 48 KB stages (dictionary block + experimental block), filled two steps ahead by
//    6 pieces per wave and step; one barrier per step (after its first k-step), as in match.hip.
//
// Algorithmic work per launch: 2 * M * n_chunk * K flops (K = kept pixels).
#include "match_device.h"
#include <stdlib.h>

namespace kpdi {

// Geometry of a variant.  Both share the operand layout (tiles of 256 patterns, steps of 48 pixels = 3 k-steps,
// a (tile, step) block of 24 KB) and the 256 x 256 workgroup tile over a ring of three 48 KB stages:
//   WAVES = 8: two waves per SIMD, wave tile 128 x 64  = 8 accumulators (128 architectural VGPRs), 6 fragment
//              reads per 8 MFMAs, 6 LDS-DMA pieces per wave and step;
//   WAVES = 4: one wave per SIMD, wave tile 128 x 128 = 16 accumulators (256 AGPRs), 8 fragment reads per 16
//              MFMAs (0.5 instead of 0.75 per MFMA: the LDS is what bounds the 8-wave form), 12 pieces per wave.
template <int WAVES>
struct Geo {
  static constexpr int DT = F16_TILE;                  // dictionary patterns per tile
  static constexpr int BK = F16_STEP;                  // pixels per step
  static constexpr int KS = BK / 16;                   // MFMA k-steps per step
  static constexpr int DBLOCK = DT * BK * 2;           // dictionary (tile, step) block, bytes
  static constexpr int EBLOCK = F16_TILE * BK * 2;     // experimental (tile, step) block, bytes
  static constexpr int STAGE = DBLOCK + EBLOCK;
  static constexpr int NSTAGE = 3;
  static constexpr int LDS = NSTAGE * STAGE;           // 144 KB (+ 32 B control words)
  static constexpr int NCG = WAVES == 8 ? 2 : 4;       // column groups (32 experimental patterns) of a wave tile
  static constexpr int WCOLS = 32 * NCG;               // experimental patterns of a wave tile
  static constexpr int WC = F16_TILE / WCOLS;          // waves side by side
  static constexpr int PIECES = STAGE / 1024 / WAVES;  // LDS-DMA pieces per wave and step: 6 / 12
  static constexpr int DPIECES = DBLOCK / 1024 / WAVES;  // ... of which dictionary: 3 / 6
  static constexpr bool ACC_A = WAVES == 4;            // accumulators in AGPRs
};

// acc += A x B for 16 pixels, A and B = 8 f16 per lane (one 16-byte LDS read).  The operands come from
// ds_read (lgkmcnt), never from a VALU write: no hazard padding needed inside the asm statement.
// 8-wave form: accumulators in architectural VGPRs (bare MFMA loop 1.41 vs 1.74 ms in AGPRs, and the
// epilogue reads them without v_accvgpr_read); 4-wave form: 256 accumulation registers = the AGPR file.
//
// The asm statement hides the MFMA from the compiler's hazard recogniser: whatever the compiler places behind it
// that touches the accumulator (a register copy, a spill) would read it before the matrix pipe has written it
// (8 passes: 11 wait states).  KPDI16_MFMA_NOPS puts those wait states into the statement itself; the shipped
// build instead keeps every accumulator in its registers throughout the loop, which tools/check_mfma_loops.py
// (run by the test suite) verifies on the generated code of every instantiation.
#ifdef KPDI16_MFMA_NOPS
#define KPDI16_MFMA_TAIL "\n\ts_nop 11"
#else
#define KPDI16_MFMA_TAIL
#endif
template <bool ACC_A>
__device__ __forceinline__ void mfma16(f32x16 &c, const f32x4 &a, const f32x4 &b) {
  if (ACC_A)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" KPDI16_MFMA_TAIL : "+a"(c) : "v"(a), "v"(b));
  else
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" KPDI16_MFMA_TAIL : "+v"(c) : "v"(a), "v"(b));
}

// float32 form: one v_mfma_f32_32x32x2_f32 per element j of the two 16-byte fragments (pixels j of lanes 0-31 and
// 4 + j of lanes 32-63 of an 8-pixel plane) - exact f32 products, the arithmetic of match.hip
__device__ __forceinline__ void mfma32(f32x16 &c, float a, float b) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// One of a wave's 1 KB LDS-DMA pieces of a stage (i < DPIECES: dictionary block, else experimental
// block); piece q = wv + WAVES * i' of its block.  `gd` / `ge`: wave-uniform block addresses.
template <int WAVES>
__device__ __forceinline__ void issue_piece16(const char *gd, const char *ge, char *stage_base, int wv, int i,
                                              unsigned goff) {
  typedef Geo<WAVES> G;
  const bool is_exp = i >= G::DPIECES;
  const int q = wv + WAVES * (is_exp ? i - G::DPIECES : i);
  __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)(is_exp ? ge : gd), 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(
      rsrc, (__attribute__((address_space(3))) void *)(stage_base + (is_exp ? G::DBLOCK : 0) + q * 1024), 16,
      (int)goff, q * 1024, 0, 0);
}

// Entry j of a list in its scratch home: `base` is wave-uniform; accesses are grouped in chunks of 16
// entries (4 KB = the immediate-offset range of a global access with a scalar base) whose base pointer is
// made opaque to the optimiser: otherwise it materialises one 64-bit address register PER ENTRY, hoists
// them out of the tile loop and spills them (1.2 KB of scratch per lane, a third of the kernel's time).
template <typename T>
__device__ __forceinline__ T *chunk_base(T *base, int chunk) {
  T *p = base + chunk * 16 * 64;
  asm volatile("" : "+s"(p));
  return p;
}

// One column group's 4 accumulators (128 dictionary rows x 32 patterns; 64 candidates per lane by
// increasing dictionary index) into the lane's list: first a screen with plain compares (bit r of
// `hot` = some lane of register r reaches the threshold), then only those registers go through the
// loop with the scalar register index (match.hip: scan_tile, FORM = 2).
// LEX: candidates may arrive in any order of dictionary index (the permuted tile order): thresholds are non-strict
// against a list's last entry and the insertion decides by (score, index).  !LEX (the 32-entry lists of the 8-wave
// form, which have no register to spare for it - tools/check_mfma_loops.py): natural tile order, strict thresholds,
// list_insert's arrival-order tie rule.
template <int KMAX, bool BOUNDED, bool F32 = false, bool LEX = true>
__device__ __forceinline__ void scan16(f32x16 (&acc)[4], float (&best)[KMAX], int (&best_idx)[KMAX], float gthr,
                                       float ub, int ub_idx, int row0, int n_valid, int idx_base, int rt_n = 4) {
  constexpr float unscale = F32 ? 1.f : 0x1p-24f;  // float16 operands are stored scaled by 2^12 each
  // (non-strict against the list's last entry: tiles arrive in a permuted order, so an equal score with a LOWER index
  // may still come - the insertion decides by (score, index))
  float thr = fmaxf(gthr, LEX ? best[KMAX - 1] : next_up(best[KMAX - 1]));
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    if (F32 && rt >= rt_n) continue;  // a partial unit of the float32 form's tail
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
      bool ok = lrow < n_valid && v >= thr && (!LEX || ranks_before(v, idx, best[KMAX - 1], best_idx[KMAX - 1]));
      if (BOUNDED) ok = ok && (v < ub || (v == ub && idx > ub_idx));
      if (ok) {
        if (LEX) {
          list_insert_lex<KMAX>(best, best_idx, v, idx);
          thr = fmaxf(gthr, best[KMAX - 1]);
        } else {
          list_insert<KMAX>(best, best_idx, v, idx);
          thr = fmaxf(gthr, next_up(best[KMAX - 1]));
        }
      }
    }
  }
}

// Candidate buffers (per lane and column group, in the scratch behind the lists).  Grouped form of the shared
// bound (>= 32 lists per pattern, every list publishes its BEST entry - which needs no sorted list): the
// buffer takes everything that passes the bound, including all 64 candidates of the first tile, and the sorted
// list is built ONCE, at the end, from the buffered candidates that pass the FINAL bound (a handful): no list is
// loaded, updated or stored inside the tile loop unless a buffer overflows (adversarial data: long runs of
// equal scores).  After the first tile the bound sits near the 3 % quantile (the minimum over 32 slots of the
// best of 128 candidates), then rises like 1 / tiles: ~7 further candidates per lane over a whole launch.
// Plain form (few lists; a list publishes its j-th best, j > 1): the list is needed, the buffer holds 8.
#ifndef KPDI16_CAP
#define KPDI16_CAP 96
#endif
constexpr int CAND_CAP = KPDI16_CAP;  // a multiple of 16
constexpr int CAND_CAP_PLAIN = 8;

template <int KMAX, bool BOUNDED, int WAVES, bool F32 = false>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void match16_kernel(MatchArgs a, float *ls_scores, int *ls_idx) {
  static_assert(!F32 || WAVES == 4, "the float32 form runs one wave per SIMD");
  typedef Geo<WAVES> G;
  constexpr int NCG = G::NCG;
#ifdef KPDI16_NO_LEX  // (developer build: round 5's order and tie handling in every instantiation)
  constexpr bool LEX = false;
#else
  constexpr bool LEX = !(KMAX == 32 && WAVES == 8);  // (scan16 above)
#endif
  constexpr int BLOCK16 = G::DBLOCK, STAGE16 = G::STAGE, NSTAGE16 = G::NSTAGE, KSTEPS16 = G::KS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef KPDI16_TIME_PHASES  // developer build: cycles of a launch's phases (tools/probes/one_step.py; profiles/r06_launch_phases.txt)
  const unsigned long long ph_t0 = __builtin_readcyclecounter();
  unsigned long long ph_prologue = 0, ph_first_loop = 0, ph_first_epi = 0, ph_loop_end = 0, ph_epi_t0 = 0;
  unsigned long long ph_fs_bound = 0, ph_fs_loop = 0;
  int ph_fs_iters = 0, ph_fs_inserts = 0;
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0 .. WAVES - 1
  const int wr = wv / G::WC, wc = wv % G::WC;
  int sp, rb;
  block_rb_sp(a, blockIdx.x, &rb, &sp);
  if (rb >= a.rows) return;  // (a workgroup of the padding of the XCD grid: whole workgroup, before any barrier)
  rb += a.row_first;
  const int n_tiles = a.n_tiles, n_valid = a.n_valid, idx_base = a.idx_base;
  const int nsteps = (2 * a.kpad) / G::BK;
  const size_t tile_bytes = (size_t)nsteps * G::DBLOCK;   // one dictionary tile, all steps
  const size_t etile_bytes = (size_t)nsteps * G::EBLOCK;  // one 256-pattern experimental tile
  const unsigned goff = (unsigned)lane * 16u;
  const char *exp_base = (const char *)a.exp + (size_t)rb * etile_bytes;
  const char *dict_base = (const char *)a.dict;

  // LDS -> MFMA fragments: lane l reads row (l & 31) of a 32-row group in plane ks: the 16-byte half
  // (l >> 5) of the row's 32 bytes, halves swapped for rows with bit 3 set (prep_device.h: half_slot)
  const unsigned half_off = (unsigned)((((lane >> 5) ^ (lane >> 3)) & 1) * 16);
  // (float32 form: the rows of this wave within the tile change with the unit - fa_base below)
  const unsigned fa_lane = (unsigned)((lane & 31) * 32) + half_off;
  unsigned fa_off = (unsigned)(wr * 128 * 32) + fa_lane;
  const unsigned fb_off = (unsigned)(BLOCK16 + (wc * G::WCOLS + (lane & 31)) * 32) + half_off;
#define KPDI_FA(base, rt, ks) (*(const f32x4 *)((base) + fa_off + (rt) * 1024 + (ks) * (G::DT * 32)))
#define KPDI_FB(base, cg, ks) (*(const f32x4 *)((base) + fb_off + (cg) * 1024 + (ks) * (F16_TILE * 32)))

  // ---- this lane's NCG lists (column group cg: pattern m_lane + 32 cg; it sees the rows
  // 4 (lane >> 5) + {0..3} + 8 j of every 32-row group of its wave's 128 rows).  Their home is the
  // scratch: entry j of list (workgroup, wave, cg) at [((wg * WAVES + wave) * NCG + cg) * KMAX + j][lane].
  const int m_lane = rb * F16_TILE + wc * G::WCOLS + (lane & 31);
  // (wave-uniform base pointers + a 32-bit lane offset: scalar-base addressing, no per-entry 64-bit
  // address registers - hoisted out of the tile loop they were spilled, 900 bytes per lane)
  float *home_s = ls_scores + (((size_t)blockIdx.x * WAVES + wv) * NCG) * KMAX * 64;
  int *home_i = ls_idx + (((size_t)blockIdx.x * WAVES + wv) * NCG) * KMAX * 64;
  const unsigned ulane = (unsigned)lane;
  // A list's home holds nothing until the list has been BUILT (a full candidate buffer, a first tile without a bound:
  // wave-uniform events) - bit cg of `built` says so, and an unbuilt list is the empty list wherever it is wanted.  (The
  // homes used to be initialised here: 160 stores per lane, 42 MB from the whole chip at once at the start of every
  // launch, and as much to load at its end - part of what a launch of this kernel costs beyond its tiles.)
  // (the 32-entry lists of the 8-wave form - !LEX - keep the initialised homes: that instantiation has no register to
  // spare, tools/check_mfma_loops.py)
  unsigned built = LEX ? 0u : ~0u;
  if (!LEX) {
#pragma unroll
    for (int c = 0; c < (NCG * KMAX + 15) / 16; ++c) {
      float *ps = chunk_base(home_s, c);
      int *pi = chunk_base(home_i, c);
#pragma unroll
      for (int j = 16 * c; j < NCG * KMAX && j < 16 * c + 16; ++j) {
        ps[(j - 16 * c) * 64 + ulane] = -INFINITY;
        pi[(j - 16 * c) * 64 + ulane] = INT_MAX;
      }
    }
  }
  // candidate buffers behind the lists: [((wg * WAVES + wave) * NCG + cg) * CAND_CAP + slot][lane]
  const size_t n_list_entries = (size_t)gridDim.x * WAVES * NCG * KMAX * 64;
  float *buf_s = ls_scores + n_list_entries + (((size_t)blockIdx.x * WAVES + wv) * NCG) * CAND_CAP * 64;
  int *buf_i = ls_idx + n_list_entries + (((size_t)blockIdx.x * WAVES + wv) * NCG) * CAND_CAP * 64;
  // per list, all the main loop keeps in registers: its last entry, the number of buffered candidates,
  // what it has published into the shared bound
  float last[NCG], pub[NCG], g[NCG];
  int cnt[NCG];
#pragma unroll
  for (int cg = 0; cg < NCG; ++cg) {
    last[cg] = pub[cg] = g[cg] = -INFINITY;
    cnt[cg] = 0;
  }
  const unsigned *line0 = a.gthr + (size_t)m_lane * BOUND_SLOTS;  // column group cg: + 32 cg BOUND_SLOTS
  const int list_id = sp * 4 + wr * 2 + (lane >> 5);
  const int my_slot = list_id & (BOUND_SLOTS - 1);
  const int bound_rank = a.bound_rank;
  const bool bound_grouped = a.bound_grouped != 0;
  const int cap = bound_rank == 1 ? CAND_CAP : CAND_CAP_PLAIN;

  // ---- dictionary tiles: t0 = tile being computed, t1 / t2 the next two (loads run two steps ahead).
  // STATIC hand-out: split sp takes the tiles sp, sp + nsplit, sp + 2 nsplit ...  Block b runs on XCD
  // b % 8 and nsplit is a multiple of 8 whenever the chip is full, so the workgroups of ALL row blocks
  // with the same split share an XCD and walk the SAME dictionary tiles at the same pace: a dictionary
  // block crosses the fabric once and is served to the other row blocks by that XCD's L2 (with the
  // dynamic hand-out of match.hip every row block re-fetched it: 16 x the dictionary per launch, which
  // the f32 kernel's 1.2 TB/s tolerates and this kernel's 6+ TB/s did not).
  // timing-only ablations: the same block every step (always an L2 hit)
#if defined(KPDI16_E_FIXED) && defined(KPDI16_D_FIXED)
#define KPDI16_ABLATE_ADDR ge = exp_base; gd = dict_base;
#elif defined(KPDI16_E_FIXED)
#define KPDI16_ABLATE_ADDR ge = exp_base;
#elif defined(KPDI16_D_FIXED)
#define KPDI16_ABLATE_ADDR gd = dict_base;
#else
#define KPDI16_ABLATE_ADDR
#endif
  // Units of work.  Whole tiles everywhere except in the float32 form's tail: the tiles from a.tail_first on are cut
  // into units of 256 >> tail_shift dictionary rows (unit u >= tail_first: tile tail_first + ((u - tail_first) >>
  // tail_shift), rows from ((u - tail_first) & (2^tail_shift - 1)) * (256 >> tail_shift)), in which a wave runs
  // 4 >> tail_shift of its four 32-row groups: the last round costs a half or a quarter of a tile-time.
  const int tail_first = F32 ? a.tail_first : n_tiles, tail_shift = F32 ? a.tail_shift : 0;
  const int n_units = tail_first + ((n_tiles - tail_first) << tail_shift);
#define KPDI16_UNIT_TILE(u) ((u) < tail_first ? (u) : tail_first + (((u) - tail_first) >> tail_shift))
#define KPDI16_UNIT_ROW(u) ((u) < tail_first ? 0 : ((((u) - tail_first) & ((1 << tail_shift) - 1)) * (F16_TILE >> tail_shift)))
#define KPDI16_UNIT_RT(u) ((u) < tail_first ? 4 : (4 >> tail_shift))
  // ORDER of this workgroup's units: the whole-tile rounds [0, perm_rounds) in a low-discrepancy walk over the
  // dictionary (round j: tile sp + nsplit * ((j * perm_stride) mod perm_rounds); the same for every row block, so the
  // workgroups of an XCD still stream the same tiles together), then the remaining rounds in natural order.  A
  // dictionary in the order a sampler emits it - or sorted by score, the hostile case - then looks to the shared bound
  // like a shuffled one: after the first rounds it holds samples from all over the dictionary, and a tile whose every
  // row beats everything seen so far happens O(log rounds) times instead of every round (bench.py:
  // extra.structured_config2.dictionary_sorted_ascending: match 0.76 -> of the f32 peak with the natural order).
  const int perm_rounds = a.perm_rounds, perm_stride = a.perm_stride;
  int seq_round = 0, seq_pos = 0;
  auto next_unit = [&]() {
    int u;
    if (seq_round < perm_rounds) {
      u = sp + a.nsplit * seq_pos;
      seq_pos += perm_stride;
      if (seq_pos >= perm_rounds) seq_pos -= perm_rounds;
    } else {
      u = sp + a.nsplit * seq_round;
    }
    ++seq_round;
    return u;
  };
  int t0 = next_unit(), t1 = next_unit(), t2 = next_unit();
  if (t0 < n_units) {
    const int last_tile = n_tiles - 1;
    int ld_pos = 0, ld_step = 0, ld_stage = 0;
    const char *gd = nullptr, *ge = nullptr;
#define KPDI16_CURSOR_SET()                                                          \
  {                                                                                  \
    int t_ = ld_pos == 0 ? t0 : (ld_pos == 1 ? t1 : t2);                             \
    t_ = KPDI16_UNIT_TILE(t_);                                                       \
    t_ = t_ < last_tile ? t_ : last_tile; /* past the end: harmless re-load */       \
    gd = dict_base + (size_t)t_ * tile_bytes + (size_t)ld_step * G::DBLOCK;          \
    ge = exp_base + (size_t)ld_step * G::EBLOCK;                                     \
    KPDI16_ABLATE_ADDR                                                               \
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
    for (int i = 0; i < G::PIECES; ++i) issue_piece16<WAVES>(gd, ge, smem + ld_stage * STAGE16, wv, i, goff);
    KPDI16_CURSOR_ADVANCE();
    KPDI16_CURSOR_SET();
#pragma unroll
    for (int i = 0; i < G::PIECES; ++i) issue_piece16<WAVES>(gd, ge, smem + ld_stage * STAGE16, wv, i, goff);
    KPDI16_CURSOR_ADVANCE();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef KPDI16_TIME_PHASES
    ph_prologue = __builtin_readcyclecounter() - ph_t0;
#endif

    // rows of this wave in a unit: 32 * rt_n consecutive rows from unit row + wr * 32 * rt_n
    auto fa_base = [&](int u) { return (unsigned)((KPDI16_UNIT_ROW(u) + wr * 32 * KPDI16_UNIT_RT(u)) * 32) + fa_lane; };
    if (F32) fa_off = fa_base(t0);
    // fragments of the three k-steps of a step, each in its own registers (static indices)
    f32x4 fa[KSTEPS16][4], fb[KSTEPS16][NCG];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) fa[0][rt] = KPDI_FA(smem, rt, 0);
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) fb[0][cg] = KPDI_FB(smem, cg, 0);

    int stage = 0;
    int tiles_done = 0, refresh_at = 0;
#ifdef KPDI16_EPI_STATS  // developer build: per tile, what the epilogue of ONE wave did (tools/probes/one_step.py; profiles/r05_f32_tile_time.txt)
    int st_hot = 0, st_iter = 0, st_cand = 0;
    unsigned long long st_t0 = 0;
#endif
#ifdef KPDI16_EPI_FINE  // developer build: cycles of the three parts of a block's epilogue (same tools, same record)
    unsigned long long fine_screen = 0, fine_pm = 0, fine_loop = 0;
    int fine_blocks = 0;
#endif
#ifdef KPDI16_TIME_EPI  // developer build (tools/build_variant.sh + tools/probes/one_step.py): where the cycles between tiles go
    unsigned long long epi_cycles = 0, epi_drain = 0;
    const unsigned long long kern_t0 = __builtin_readcyclecounter();
#endif
#pragma clang loop unroll(disable)
    for (;;) {  // dictionary tiles (units)
      const int rt_n = F32 ? KPDI16_UNIT_RT(t0) : 4;                        // row groups of this wave in this unit
      f32x16 acc[NCG][4];  // [column group][32-row group]
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[cg][rt][r] = 0.f;
#pragma clang loop unroll(disable)
      for (int step = 0; step < nsteps; ++step) {
        const char *ls = smem + stage * STAGE16;
        const int nstage = stage == NSTAGE16 - 1 ? 0 : stage + 1;
        const char *ls_next = smem + nstage * STAGE16;
        if (step == nsteps - 1 && tiles_done >= refresh_at) {
          // the shared bound of this wave's patterns, for the epilogue: all lines in flight, then reduced (a wait
          // of one memory latency that nothing hides - the waves run in lockstep); refreshed after the tiles
          // 0, 1, 2, 4, 7, 11, 17, 26 ... of a workgroup: the bound rises like the logarithm of the candidates
          // seen, and a bound that is a few tiles old lets 1.5 x as many of the ~0.1 candidates per lane pass
          BoundHalf raw[NCG];
#pragma unroll
          for (int cg = 0; cg < NCG; ++cg) bound_load_half(raw[cg], line0 + 32 * cg * BOUND_SLOTS, lane >> 5);
#pragma unroll
          for (int cg = 0; cg < NCG; ++cg) g[cg] = bound_reduce_half<KMAX>(raw[cg], bound_grouped, lane >> 5);
          refresh_at = tiles_done + 1 + (tiles_done >> 1);
        }
        KPDI16_CURSOR_SET();
        char *ld_base = smem + ld_stage * STAGE16;
#pragma unroll
        for (int ks = 0; ks < KSTEPS16; ++ks) {
          if (ks == 1) {
            // ---- the step's only synchronisation point: this wave's pieces of step + 1 (issued during
            // the previous step) have landed; after the barrier step + 1 is complete in LDS and every
            // wave is past the previous step, whose stage is refilled below
#ifndef KPDI16_NO_BARRIER
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#endif
          }
          const int nk = ks == KSTEPS16 - 1 ? 0 : ks + 1;         // fragments read during this k-step
          const char *src = ks == KSTEPS16 - 1 ? ls_next : ls;    // ... of the next step for the last one
          if (F32 && ks == KSTEPS16 - 1 && step == nsteps - 1) fa_off = fa_base(t1);  // those are the next unit's rows
          if (F32) {
            // ---- float32 form: 64 MFMAs of 64 pipe cycles per k-step; behind every four of them one of the 8
            // fragment reads of the next k-step, then (after the barrier) this wave's 6 LDS-DMA pieces
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int rt = 0; rt < 4; ++rt) {
                if (rt < rt_n) {  // (a partial unit of the tail runs 2 or 1 of its row groups; the schedule stays)
#pragma unroll
                  for (int cg = 0; cg < NCG; ++cg) mfma32(acc[cg][rt], fa[ks][rt][j], fb[ks][cg][j]);
                }
                const int slot = 4 * j + rt;
#ifndef KPDI16_NO_READS
                if (slot < 4) fa[nk][slot] = KPDI_FA(src, slot, nk);
                else if (slot < 8) fb[nk][slot - 4] = KPDI_FB(src, slot - 4, nk);
#endif
#ifndef KPDI16_NO_DMA
                if (ks >= 1 && slot >= 8 && slot < 8 + 6) issue_piece16<WAVES>(gd, ge, ld_base, wv, (ks - 1) * 6 + slot - 8, goff);
#endif
                __builtin_amdgcn_sched_barrier(0);
              }
          } else
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) mfma16<G::ACC_A>(acc[cg][rt], fa[ks][rt], fb[ks][cg]);
            // in the shadow of these MFMAs: fragments of the next k-step and, after the barrier,
            // this wave's LDS-DMA pieces of the step two ahead
#ifndef KPDI16_NO_READS
            if (NCG == 4) {
              fa[nk][rt] = KPDI_FA(src, rt, nk);
              fb[nk][rt] = KPDI_FB(src, rt, nk);
            } else {
              fa[nk][rt] = KPDI_FA(src, rt, nk);
              if (rt == 1) fb[nk][0] = KPDI_FB(src, 0, nk);
              if (rt == 3) fb[nk][1] = KPDI_FB(src, 1, nk);
            }
#endif
#ifndef KPDI16_NO_DMA
            if (WAVES == 8) {  // 3 pieces in each of the k-steps 1 and 2
              if (ks >= 1 && rt < 3) issue_piece16<WAVES>(gd, ge, ld_base, wv, (ks - 1) * 3 + rt, goff);
            } else {           // 6 in each
              if (ks >= 1) {
                issue_piece16<WAVES>(gd, ge, ld_base, wv, (ks - 1) * 6 + rt, goff);
                if (rt < 2) issue_piece16<WAVES>(gd, ge, ld_base, wv, (ks - 1) * 6 + 4 + rt, goff);
              }
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        KPDI16_CURSOR_ADVANCE();
        stage = nstage;
      }  // steps
#ifdef KPDI16_TIME_EPI
      const unsigned long long epi_t0 = __builtin_readcyclecounter();
#endif
#ifdef KPDI16_TIME_PHASES
      ph_epi_t0 = __builtin_readcyclecounter();
      if (tiles_done == 0) ph_first_loop = ph_epi_t0 - ph_t0 - ph_prologue;
#endif
      // the last MFMAs (8 passes) must have written the accumulators before they are read
      // (the pipe retires MFMAs in order: ONE wait covers all of them; the empty statements only tie every accumulator to
      // this point - 16 x 24 wait states per tile were spent here before)
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          if (cg == 0 && rt == 0) {
            if (G::ACC_A)
              asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc[cg][rt]));
            else
              asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[cg][rt]));
          } else {
            if (G::ACC_A)
              asm volatile("" : "+a"(acc[cg][rt]));
            else
              asm volatile("" : "+v"(acc[cg][rt]));
          }
        }
#ifdef KPDI16_TIME_EPI
      epi_drain += __builtin_readcyclecounter() - epi_t0;
#endif
#ifdef KPDI16_EPI_STATS
      st_hot = st_iter = st_cand = 0;
      st_t0 = __builtin_readcyclecounter();
#endif
      {
        // ---- epilogue of the tile.  Steady state (per column group): the 64 accumulator registers are
        // compared with the pre-scaled threshold (a v_max3 tree per 16 registers first) and the few
        // candidates that pass are APPENDED to the lane's candidate buffer in the scratch - the sorted list
        // itself is only loaded, updated and stored when a buffer is full (the first tiles of a launch,
        // rarely afterwards) and at the end.  The waves of a workgroup run in lockstep (one barrier per
        // step), so every cycle spent here is lost on the matrix pipe: a list update per tile cost 23 %.
        const int row0 = F32 ? KPDI16_UNIT_TILE(t0) * G::DT + KPDI16_UNIT_ROW(t0) + wr * 32 * rt_n + 4 * (lane >> 5)
                             : t0 * G::DT + wr * 128 + 4 * (lane >> 5);
        // ---- the FIRST tile of a launch has no bound to screen with.  Building every lane's list from all 64 candidates
        // of each column group (scan16 below) is what a launch of this kernel costs beyond its tiles: ~360 000 cycles =
        // 0.15 ms (profiles/r04_epilogue_ab.txt).  A bound needs no list, though - only every list's BEST entry (grouped
        // form): each lane publishes the maximum of its 64 candidates right away, the workgroups of a row block do so
        // within microseconds of each other (one launch, equal work), and a short poll later the bound stands near the
        // 3 % quantile: ~2 of the 64 candidates pass and take the steady-state append path.  If the bound is not
        // complete after the poll (a workgroup of the row block lags, or is not resident) the list is built directly, as before.
        bool first_fast = false;
#ifndef KPDI16_FIRST_TILE_DIRECT  // (developer build: rounds 3's first tile)
        if (tiles_done == 0 && !BOUNDED && bound_rank == 1 && bound_grouped &&
            (F32 ? KPDI16_UNIT_TILE(t0) : t0) * G::DT + G::DT <= n_valid) {
#pragma unroll
          for (int cg = 0; cg < NCG; ++cg) {
            float m = -INFINITY;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
              if (F32 && rt >= rt_n) continue;
#pragma unroll
              for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[cg][rt][r]);
            }
            m = m * (F32 ? 1.f : 0x1p-24f) + 0.f;
            if (m > pub[cg]) {
              pub[cg] = m;
              __hip_atomic_fetch_max(const_cast<unsigned *>(line0) + 32 * cg * BOUND_SLOTS + my_slot, score_key(m),
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
#pragma unroll 1
          for (int it = 0; it < 48 && !first_fast; ++it) {
            BoundHalf raw[NCG];
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) bound_load_half(raw[cg], line0 + 32 * cg * BOUND_SLOTS, lane >> 5);
            bool ready = true;
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) {
              g[cg] = bound_reduce_half<KMAX>(raw[cg], bound_grouped, lane >> 5);
              ready = ready && g[cg] > -INFINITY;
            }
            first_fast = __builtin_amdgcn_ballot_w64(!ready) == 0;
            if (!first_fast) __builtin_amdgcn_s_sleep(24);
          }
        }
#endif
        // index of a built list's last entry (a candidate that TIES with it passes only with a lower index: tiles
        // arrive in a permuted order); lists are rarely built before the end - then nothing is loaded
        int lidx[NCG];
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
          lidx[cg] = INT_MAX;
          if (LEX && ((built >> cg) & 1)) lidx[cg] = chunk_base(home_i + cg * KMAX * 64, (KMAX - 1) / 16)[((KMAX - 1) % 16) * 64 + ulane];
        }
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
#ifdef KPDI16_NO_EPILOGUE  // (the MFMAs are asm volatile: they stay)
          continue;
#endif
#ifdef KPDI16_SCREEN_NONE  // the bound is loaded and reduced, nothing is screened
          if (g[cg] == 12345.f) cnt[cg] = 1;
          continue;
#endif
          constexpr float unscale = F32 ? 1.f : 0x1p-24f;
          // BOUNDED: the pass admits what ranks strictly after (ub, ubi); read here, not carried through the
          // MFMA loop in registers (there is none to spare: tools/check_mfma_loops.py)
          float ub_cg = INFINITY;
          int ubi_cg = -1;
          if (BOUNDED) {
            ub_cg = a.bound_score[m_lane + 32 * cg];
            ubi_cg = a.bound_idx[m_lane + 32 * cg];
          }
          const float thr = fmaxf(g[cg], LEX ? last[cg] : next_up(last[cg]));
          const float thr_raw = F32 ? thr : thr * 0x1p24f;  // exact: the float16 form's accumulators hold 2^24 * score
          float *bs = buf_s + cg * CAND_CAP * 64;
          int *bi = buf_i + cg * CAND_CAP * 64;
          unsigned *line = const_cast<unsigned *>(line0) + 32 * cg * BOUND_SLOTS;
          int c = cnt[cg];
          // The FIRST tile of a launch has no bound yet: every one of its 64 candidates per lane would be appended -
          // 134 MB of stores from the whole chip at once, and as much to read back at the end: ~0.13 of the 0.20 ms a
          // launch of the float32 form cost beyond its tiles (tools/tile_ramp_probe.py on the ablation builds).  It goes
          // the way of a full buffer instead: the list is built from the accumulators directly (scan16: 20 entries stay,
          // 40 stores), and its last entry screens the lane's next tiles beside the shared bound.
#ifdef KPDI16_FIRST_TILE_APPENDS  // (developer build: round 2's behaviour)
          const bool first_tile = false;
#else
          const bool first_tile = tiles_done == 0 && !first_fast;
#endif
          bool overflow = first_tile;
          float mx = -INFINITY;
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) {
            if (first_tile) continue;
            if (F32 && rt >= rt_n) continue;
#ifdef KPDI16_EPI_FINE
            const unsigned long long f0 = __builtin_readcyclecounter();
#endif
            float m = acc[cg][rt][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[cg][rt][r]);
#ifdef KPDI16_SCAN_NEVER  // timing-only ablation: the screen (maxima + ballot) runs, no candidate is ever taken
            if (__builtin_amdgcn_ballot_w64(m >= thr_raw) != 0xdeadbeefull) continue;
#endif
            if (__builtin_amdgcn_ballot_w64(m >= thr_raw) == 0) continue;  // wave-uniform
            // (the bounded 32-entry instantiation of the 8-wave form has no register to spare for the mask form below - it
            // would reload a fragment from scratch inside the MFMA loop, tools/check_mfma_loops.py - and keeps round 3's
            // form: one exec-masked block per accumulator register; KPDI16_APPEND_PER_REGISTER forces it everywhere, A/B)
#ifdef KPDI16_APPEND_PER_REGISTER
            constexpr bool PER_REGISTER = true;
#else
            constexpr bool PER_REGISTER = BOUNDED && KMAX == 32 && WAVES == 8;
#endif
            if (PER_REGISTER) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int lrow = row0 + rt * 32 + (r & 3) + 8 * (r >> 2);
              const float v = acc[cg][rt][r] * unscale + 0.f;  // -0 -> +0 so that ties compare as the merge does
              const int idx = idx_base + lrow;
              bool ok = acc[cg][rt][r] >= thr_raw && lrow < n_valid && (!LEX || v > last[cg] || idx < lidx[cg]);
              if (BOUNDED) ok = ok && (v < ub_cg || (v == ub_cg && idx > ubi_cg));
              if (ok) {
                if (c < cap) {
                  bs[c * 64 + ulane] = v;
                  bi[c * 64 + ulane] = idx;
                  ++c;
                  mx = fmaxf(mx, v);
                } else {
                  overflow = true;
                }
              }
            }
            } else {
            // ---- some lane of this block of 16 registers holds a candidate (typically ONE lane, one register).  Which of
            // a lane's registers pass is folded into a per-lane bit mask with vector instructions only (compare -> select ->
            // or: a per-register `if` costs a vector-compare -> scalar-and -> save-exec -> branch chain per register, ~150
            // cycles of scalar dependency stalls each, 16 x 16 times per tile: 2 % of the kernel, profiles/r03_epilogue_cycles.txt);
            // then ONE loop, as long as any lane has a bit left: lowest bit -> register (a 16-way select) -> append.  Bits
            // are taken in ascending register order = ascending dictionary index, like the per-register form.
#ifdef KPDI16_EPI_FINE
            const unsigned long long f1 = __builtin_readcyclecounter();
#endif
            unsigned pm = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) pm |= acc[cg][rt][r] >= thr_raw ? (1u << r) : 0u;
#ifdef KPDI16_EPI_FINE
            asm volatile("" : "+v"(pm));
            const unsigned long long f2 = __builtin_readcyclecounter();
#endif
#ifdef KPDI16_EPI_STATS
            ++st_hot;
#endif
#pragma unroll 1
            while (__builtin_amdgcn_ballot_w64(pm != 0) != 0) {
#ifdef KPDI16_EPI_STATS
              ++st_iter;
              st_cand += __builtin_popcountll(__builtin_amdgcn_ballot_w64(pm != 0));
#endif
              if (pm != 0) {
                const int r = __builtin_ctz(pm);
                pm &= pm - 1;
                float raw = acc[cg][rt][0];
#pragma unroll
                for (int q = 1; q < 16; ++q) raw = r == q ? acc[cg][rt][q] : raw;
                const int lrow = row0 + rt * 32 + (r & 3) + 8 * (r >> 2);
                const float v = raw * unscale + 0.f;  // -0 -> +0 so that ties compare as the merge does
                const int idx = idx_base + lrow;
                bool ok = lrow < n_valid && (!LEX || v > last[cg] || idx < lidx[cg]);
                if (BOUNDED) ok = ok && (v < ub_cg || (v == ub_cg && idx > ubi_cg));
                if (ok) {
                  if (c < cap) {
#ifndef KPDI16_NO_APPEND_STORES
                    bs[c * 64 + ulane] = v;
                    bi[c * 64 + ulane] = idx;
#endif
                    ++c;
                    mx = fmaxf(mx, v);
                  } else {
                    overflow = true;
                  }
                }
              }
            }
#ifdef KPDI16_EPI_FINE
            {
              const unsigned long long f3 = __builtin_readcyclecounter();
              fine_screen += f1 - f0;
              fine_pm += f2 - f1;
              fine_loop += f3 - f2;
              ++fine_blocks;
            }
#endif
            }
          }
          if (__builtin_amdgcn_ballot_w64(overflow) == 0) {
            cnt[cg] = c;
            // grouped form of the shared bound (every list publishes its best entry): the best buffered
            // candidate counts as well
            if (bound_rank == 1 && mx > pub[cg]) {
              pub[cg] = mx;
              __hip_atomic_fetch_max(line + my_slot, score_key(mx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          } else {
            // ---- a buffer is full: list <- buffered candidates of the earlier tiles (in arrival order), then
            // this tile's accumulators the direct way (what was appended from this tile above is dropped)
            float best[KMAX];
            int bidx[KMAX];
            float *hs = home_s + cg * KMAX * 64;
            int *hi = home_i + cg * KMAX * 64;
            if (!LEX || ((built >> cg) & 1)) {
#pragma unroll
              for (int q = 0; q < (KMAX + 15) / 16; ++q) {
                const float *ps = chunk_base(hs, q);
                const int *pi = chunk_base(hi, q);
#pragma unroll
                for (int j = 16 * q; j < KMAX && j < 16 * q + 16; ++j) {
                  best[j] = ps[(j - 16 * q) * 64 + ulane];
                  bidx[j] = pi[(j - 16 * q) * 64 + ulane];
                }
              }
            } else {  // (never built: the empty list)
#pragma unroll
              for (int j = 0; j < KMAX; ++j) {
                best[j] = -INFINITY;
                bidx[j] = INT_MAX;
              }
              if (LEX) built |= 1u << cg;
            }
#pragma unroll 1
            for (int i = 0; __builtin_amdgcn_ballot_w64(i < cnt[cg]) != 0; ++i) {
              if (i < cnt[cg]) {
                const float v = bs[i * 64 + ulane];
                const int id = bi[i * 64 + ulane];
                if (LEX) {
                  if (ranks_before(v, id, best[KMAX - 1], bidx[KMAX - 1])) list_insert_lex<KMAX>(best, bidx, v, id);
                } else if (v > best[KMAX - 1]) {
                  list_insert<KMAX>(best, bidx, v, id);
                }
              }
            }
            if (LEX && a.epi_stats) {  // (developer counters, profiling level 3: what was appended before this list was built)
              const unsigned app = wave_sum_u32((unsigned)cnt[cg]);
              if (lane == 0) {
                atomicAdd(a.epi_stats + 1, (unsigned long long)app);
                atomicAdd(a.epi_stats + (first_tile ? 3 : 2), 1ull);
              }
            }
            cnt[cg] = 0;
            scan16<KMAX, BOUNDED, F32, LEX>(acc[cg], best, bidx, g[cg], ub_cg, ubi_cg, row0, n_valid, idx_base, rt_n);
#pragma unroll
            for (int q = 0; q < (KMAX + 15) / 16; ++q) {
              float *ps = chunk_base(hs, q);
              int *pi = chunk_base(hi, q);
#pragma unroll
              for (int j = 16 * q; j < KMAX && j < 16 * q + 16; ++j) {
                ps[(j - 16 * q) * 64 + ulane] = best[j];
                pi[(j - 16 * q) * 64 + ulane] = bidx[j];
              }
            }
            last[cg] = best[KMAX - 1];
            float now = best[0];  // the entry the shared bound is built from
#pragma unroll
            for (int j = 1; j < KMAX; ++j) now = j == bound_rank - 1 ? best[j] : now;
            if (now > pub[cg]) {
              pub[cg] = now;
              __hip_atomic_fetch_max(line + my_slot, score_key(now), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
#ifdef KPDI16_TIME_EPI
        epi_cycles += __builtin_readcyclecounter() - epi_t0;
#endif
#ifdef KPDI16_EPI_STATS
        {
          const unsigned long long dt = __builtin_readcyclecounter() - st_t0;
          if (blockIdx.x == 100 && lane == 0 && wv == 0)
            printf("tile %d: %llu cycles, %d hot blocks of 16, %d loop iterations, %d candidates of the wave, first_fast %d\n", tiles_done, dt,
                   st_hot, st_iter, st_cand, (int)first_fast);
        }
#endif
#ifdef KPDI16_TIME_PHASES
        if (tiles_done == 0) ph_first_epi = __builtin_readcyclecounter() - ph_epi_t0;
        ph_loop_end = __builtin_readcyclecounter() - ph_t0;
#endif
        ++tiles_done;
        t0 = t1;
        t1 = t2;
        t2 = next_unit();
        --ld_pos;
        if (t0 >= n_units) break;
      }
    }
#ifdef KPDI16_EPI_FINE
    if (blockIdx.x == 100 && lane == 0)
      printf("wave %d: %d hot blocks: screen %llu, mask %llu, loop %llu cycles per hot block\n", wv, fine_blocks, fine_screen / fine_blocks,
             fine_pm / fine_blocks, fine_loop / fine_blocks);
#endif
#ifdef KPDI16_TIME_EPI
    if ((blockIdx.x == 0 || blockIdx.x == 100) && lane == 0)
      printf("block %d wave %d: %d tiles, %llu cycles between tiles (%llu of them waiting for the last MFMAs) of %llu in the tile loop "
             "(%.2f %%)\n", (int)blockIdx.x, wv, tiles_done, epi_cycles, epi_drain, __builtin_readcyclecounter() - kern_t0,
             100.0 * epi_cycles / (double)(__builtin_readcyclecounter() - kern_t0));
#endif
  }

  // ---- the buffered candidates that pass the FINAL shared bound join their lists (in arrival order = by
  // increasing dictionary index), then lists -> [m_pad][lists][KMAX] for the merge kernel
  {
    const int lists = 4 * a.nsplit;
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) {
      const size_t ol = (size_t)(m_lane + 32 * cg) * lists + (size_t)list_id, o = ol * KMAX;
#ifdef KPDI16_SKIP_FINAL  // (developer build, TIMING ONLY - results are wrong: what a launch costs without its final stage)
      if (LEX) {
        a.part_cnt[ol] = 0;
        continue;
      }
#endif
      // a candidate below the bound has KMAX better ones somewhere among the pattern's lists
#ifdef KPDI16_TIME_PHASES
      const unsigned long long fs0 = __builtin_readcyclecounter();
#endif
      // (any bound that ever stood is valid; the one this wave refreshed last - during its last tiles - is nearly the
      // final one and costs nothing: four dependent loads from memory here were 10 us of every launch)
      float tf = g[cg];
#ifdef KPDI16_FINAL_BOUND_LOAD  // (developer build: the final stage reads the bound from memory, as before round 6)
      tf = -INFINITY;
#endif
      if (__builtin_amdgcn_ballot_w64(!(tf > -INFINITY)) != 0) tf = shared_bound<KMAX>(line0 + 32 * cg * BOUND_SLOTS, bound_grouped);
#ifdef KPDI16_TIME_PHASES
      asm volatile("" : "+v"(tf));
      const unsigned long long fs1 = __builtin_readcyclecounter();
      ph_fs_bound += fs1 - fs0;
#endif
#ifdef KPDI16_NO_DIRECT  // (developer build: every list through the sorted path)
      constexpr bool DIRECT = false;
#else
      // (the one-wave-per-SIMD forms only.  With this path compiled in, the 8-wave float16 kernel took 8 % LONGER as a whole -
      // 6.3 -> 6.85 ms at K = 14 400, 2.72 -> 2.97 at K = 3600 - although its tile loop and its final stage count the same
      // cycles and its hot loop is instruction for instruction the same (A/B of developer builds on one box, alternating:
      // profiles/r06_f16_ab.txt; why is not established - not the scratch size, which is the same either way).  Its final
      // stage is 20 k cycles of 5 M: there is nothing to gain there for it anyway.)
#ifdef KPDI16_DIRECT_ALL  // (developer build: the A/B of profiles/r06_f16_ab.txt)
      constexpr bool DIRECT = LEX;
#else
      constexpr bool DIRECT = LEX && WAVES == 4;
#endif
#endif
      if (DIRECT && !((built >> cg) & 1)) {
        // ---- the usual case: this list was never built.  The merge kernel takes a partial list as a SET of candidates
        // (merge.hip ranks by key, whatever the order), so the buffered candidates that reach the final bound - a handful
        // of a lane's ~10 - go straight to the lane's partial list, in arrival order, unsorted, and "no entry" behind
        // them: 16 predicated stores per batch instead of one 20-entry sorted insertion per survivor (those insertions, 6
        // per column group at ~2500 cycles each, were most of what a launch spent behind its last tile:
        // profiles/r06_launch_phases.txt).  A lane with more than KMAX survivors (adversarial data): the sorted path below.
#ifdef KPDI16_TIME_PHASES
        const unsigned long long fs2 = __builtin_readcyclecounter();
#endif
        int n = 0;
#pragma unroll 1
        for (int base = 0; __builtin_amdgcn_ballot_w64(base < cnt[cg]) != 0; base += 16) {
          const float *ps = chunk_base(buf_s + (cg * CAND_CAP + base) * 64, 0);
          const int *pi = chunk_base(buf_i + (cg * CAND_CAP + base) * 64, 0);
          f32x16 vs;
          int ids[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const bool in = base + e < cnt[cg];
            vs[e] = in ? ps[e * 64 + ulane] : -INFINITY;
            ids[e] = in ? pi[e * 64 + ulane] : INT_MAX;
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            if (vs[e] >= tf && ids[e] != INT_MAX) {
              if (n < KMAX) {
                a.part_scores[o + n] = vs[e];
                a.part_idx[o + n] = ids[e];
              }
              ++n;
            }
          }
        }
#ifdef KPDI16_TIME_PHASES
        ph_fs_loop += __builtin_readcyclecounter() - fs2;
        ++ph_fs_iters;
#endif
        if (__builtin_amdgcn_ballot_w64(n > KMAX) == 0) {
          a.part_cnt[ol] = n;  // (the merge takes the first n entries of this list: nothing is stored behind them)
          if (a.epi_stats) {
            const unsigned app = wave_sum_u32((unsigned)cnt[cg]);
            if (lane == 0) {
              atomicAdd(a.epi_stats, 64ull);
              atomicAdd(a.epi_stats + 1, (unsigned long long)app);
            }
          }
          continue;
        }
      }
      float best[KMAX];
      int bidx[KMAX];
      if (!LEX || ((built >> cg) & 1)) {
#pragma unroll
        for (int q = 0; q < (KMAX + 15) / 16; ++q) {
          const float *ps = chunk_base(home_s + cg * KMAX * 64, q);
          const int *pi = chunk_base(home_i + cg * KMAX * 64, q);
#pragma unroll
          for (int j = 16 * q; j < KMAX && j < 16 * q + 16; ++j) {
            best[j] = ps[(j - 16 * q) * 64 + ulane];
            bidx[j] = pi[(j - 16 * q) * 64 + ulane];
          }
        }
      } else {  // (never built: the empty list + the buffered candidates below)
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
          best[j] = -INFINITY;
          bidx[j] = INT_MAX;
        }
      }
#pragma unroll 1
      for (int base = 0; __builtin_amdgcn_ballot_w64(base < cnt[cg]) != 0; base += 16) {
        const float *ps = chunk_base(buf_s + (cg * CAND_CAP + base) * 64, 0);  // 16 entries in flight, not one
        const int *pi = chunk_base(buf_i + (cg * CAND_CAP + base) * 64, 0);
        f32x16 vs;
        int ids[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const bool in = base + e < cnt[cg];
          vs[e] = in ? ps[e * 64 + ulane] : -INFINITY;
          ids[e] = in ? pi[e * 64 + ulane] : INT_MAX;
        }
        // Which of a lane's 16 candidates may still enter its list: a bit mask per lane; then ONE loop, as long as any
        // lane has a bit left, in which every lane inserts ITS next candidate (lowest bit = arrival order -> a 16-way
        // select).  A handful of a lane's ~10 buffered candidates reach the final bound, so the loop runs 3 - 5 times;
        // walking the 16 slots wave-wide (one insertion per slot in which ANY lane had a candidate - nearly all of them)
        // ran 16+ insertions of ~250 instructions per column group: most of the 55 us a launch spent behind its last tile
        // (profiles/r06_launch_phases.txt).
        unsigned pm = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e)
          pm |= (vs[e] >= tf && (LEX ? ranks_before(vs[e], ids[e], best[KMAX - 1], bidx[KMAX - 1]) : vs[e] > best[KMAX - 1])) ? (1u << e) : 0u;
#ifdef KPDI16_TIME_PHASES
        asm volatile("" : "+v"(pm));
        const unsigned long long fs2 = __builtin_readcyclecounter();
        ++ph_fs_iters;
#endif
#pragma unroll 1
        while (__builtin_amdgcn_ballot_w64(pm != 0) != 0) {
#ifdef KPDI16_TIME_PHASES
          ++ph_fs_inserts;
#endif
          if (pm != 0) {
            const int e = __builtin_ctz(pm);
            pm &= pm - 1;
            float v = vs[0];
            int id = ids[0];
#pragma unroll
            for (int q = 1; q < 16; ++q) {
              v = e == q ? vs[q] : v;
              id = e == q ? ids[q] : id;
            }
            if (LEX) {
              if (ranks_before(v, id, best[KMAX - 1], bidx[KMAX - 1])) list_insert_lex<KMAX>(best, bidx, v, id);
            } else if (v > best[KMAX - 1]) {
              list_insert<KMAX>(best, bidx, v, id);
            }
          }
        }
#ifdef KPDI16_TIME_PHASES
        ph_fs_loop += __builtin_readcyclecounter() - fs2;
#endif
      }
      if (LEX && a.epi_stats) {
        const unsigned app = wave_sum_u32((unsigned)cnt[cg]);
        if (lane == 0) {
          atomicAdd(a.epi_stats, 64ull);
          atomicAdd(a.epi_stats + 1, (unsigned long long)app);
        }
      }
      a.part_cnt[ol] = KMAX;  // (the sorted path writes the whole list, "no entry" tail included)
#pragma unroll
      for (int j = 0; j < KMAX; ++j) {
        a.part_scores[o + j] = best[j];
        a.part_idx[o + j] = bidx[j];
      }
    }
  }
#ifdef KPDI16_TIME_PHASES
  {
    const unsigned long long end = __builtin_readcyclecounter() - ph_t0;
    if ((blockIdx.x == 0 || blockIdx.x == 100 || blockIdx.x == 255) && lane == 0 && wv == 0)
      printf("block %d: prologue %llu, first tile's steps %llu, first epilogue %llu, tile loop ends at %llu, final stage %llu, kernel %llu cycles "
             "(shader cycles); final stage: bound loads %llu, insert loops %llu (%d batches, %d insertions)\n", (int)blockIdx.x, ph_prologue,
             ph_first_loop, ph_first_epi, ph_loop_end, end - ph_loop_end, end, ph_fs_bound, ph_fs_loop, ph_fs_iters, ph_fs_inserts);
  }
#endif
}

// floats (and as many ints) the kernel keeps per launch: the lists and the candidate buffers behind them
static size_t scratch16_entries(int grid, int waves, int list_len) {
  (void)waves;  // waves x column groups per wave = 16 lists per lane position in both forms
  return (size_t)grid * 16 * (list_len + CAND_CAP) * 64;
}
size_t match16_scratch_bytes(int grid, int waves, int list_len) {
  return scratch16_entries(grid, waves, list_len) * (sizeof(float) + sizeof(int));
}

template <int KMAX, bool BOUNDED, int WAVES, bool F32 = false>
static hipError_t launch16_t(const MatchArgs &args_in, int grid, void *scratch, hipStream_t s) {
  // (the attribute belongs to the function ON A DEVICE: remembered per device, not per process - a second context
  // on another GPU would otherwise launch a 144 KB kernel without it)
  static unsigned long long attr_set = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (!((attr_set >> (dev & 63)) & 1ull)) {
    hipError_t e = hipFuncSetAttribute((const void *)match16_kernel<KMAX, BOUNDED, WAVES, F32>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Geo<WAVES>::LDS + 32);
    if (e != hipSuccess) return e;
    attr_set |= 1ull << (dev & 63);
  }
  MatchArgs args = args_in;
  if (KMAX == 32 && WAVES == 8) args.perm_rounds = 0;  // (!LEX instantiations: natural order, match16_kernel)
#ifdef KPDI16_NO_LEX
  args.perm_rounds = 0;
#endif
  // scratch: scores of all lists, then their indices
  float *ls = (float *)scratch;
  int *li = (int *)(ls + scratch16_entries(grid, WAVES, KMAX));
  hipLaunchKernelGGL((match16_kernel<KMAX, BOUNDED, WAVES, F32>), dim3(grid), dim3(64 * WAVES), Geo<WAVES>::LDS + 32, s,
                     args, ls, li);
  return hipGetLastError();
}

template <int WAVES>
static hipError_t launch16_w(const MatchLaunch &a, const MatchArgs &g, void *scratch, hipStream_t s) {
  const int grid = g.rows_grid * a.nsplit;
  const bool bounded = a.bound_score != nullptr;
  switch (a.list_len) {
    case 1: return bounded ? launch16_t<1, true, WAVES>(g, grid, scratch, s) : launch16_t<1, false, WAVES>(g, grid, scratch, s);
    case 8: return bounded ? launch16_t<8, true, WAVES>(g, grid, scratch, s) : launch16_t<8, false, WAVES>(g, grid, scratch, s);
    case 20:
      return bounded ? launch16_t<20, true, WAVES>(g, grid, scratch, s) : launch16_t<20, false, WAVES>(g, grid, scratch, s);
    case 32:
      // (the bounded 32-entry form of the one-wave-per-SIMD variant does not fit its registers - an accumulator
      // would live in scratch, tools/check_mfma_loops.py - so that launch runs the 8-wave kernel: same operands, same lists)
      return bounded ? launch16_t<32, true, 8>(g, grid, scratch, s) : launch16_t<32, false, WAVES>(g, grid, scratch, s);
    default: return hipErrorInvalidValue;
  }
}

// the float32 form (operand_form 3): exact f32 products on the one-wave-per-SIMD kernel
static hipError_t launch16_f32(const MatchLaunch &a, const MatchArgs &g, void *scratch, hipStream_t s) {
  const int grid = g.rows_grid * a.nsplit;
  const bool bounded = a.bound_score != nullptr;
  switch (a.list_len) {
    case 1:
      return bounded ? launch16_t<1, true, 4, true>(g, grid, scratch, s) : launch16_t<1, false, 4, true>(g, grid, scratch, s);
    case 8:
      return bounded ? launch16_t<8, true, 4, true>(g, grid, scratch, s) : launch16_t<8, false, 4, true>(g, grid, scratch, s);
    case 20:
      return bounded ? launch16_t<20, true, 4, true>(g, grid, scratch, s) : launch16_t<20, false, 4, true>(g, grid, scratch, s);
    case 32:
      // (bounded passes of this form rank 20 entries at a time, api.hip: the bounded 32-entry instantiation does not
      // fit its registers - tools/check_mfma_loops.py)
      return bounded ? hipErrorInvalidValue : launch16_t<32, false, 4, true>(g, grid, scratch, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_match16(const MatchLaunch &a, int waves, void *list_scratch, hipStream_t s) {
  if ((a.operand_form != 2 && a.operand_form != 3) || a.row_tiles != 4 || !list_scratch) return hipErrorInvalidValue;
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
  g.part_cnt = a.part_cnt;
  if (!g.part_cnt) return hipErrorInvalidValue;
  g.bound_score = a.bound_score;
  g.bound_idx = a.bound_idx;
  g.gthr = a.gthr;
  g.bound_rank = a.bound_rank;
  g.bound_grouped = a.bound_grouped;
  g.tile_ctr = a.tile_ctr;
  g.tile_groups = 1;
  g.fixed_draws = 1 << 30;
  g.perm_rounds = a.perm_rounds;
  g.perm_stride = a.perm_stride;
  g.epi_stats = a.epi_stats;
  g.tail_first = a.operand_form == 3 && a.tail_shift > 0 ? a.tail_first : a.n_tiles;
  g.tail_shift = a.operand_form == 3 ? a.tail_shift : 0;
  g.xcd_rows = a.xcd_rows;
  g.xcd_splits = a.xcd_splits;
  g.rows = a.rows;
  g.rows_grid = a.xcd_rows > 0 && a.rows_grid > a.rows ? a.rows_grid : a.rows;
  if (a.operand_form == 3) return launch16_f32(a, g, list_scratch, s);
  return waves == 4 ? launch16_w<4>(a, g, list_scratch, s) : launch16_w<8>(a, g, list_scratch, s);
}

}  // namespace kpdi
