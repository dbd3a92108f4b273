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
//  * ONE WAVE PER SIMD.  Measured on MI355X (tools/ubench/mfma_f32.hip):
//    v_mfma_f32_32x32x2_f32 sustains 99 % of the 157.3 TFLOP/s f32 peak from one wave
//    per SIMD, but only 66 % when two waves share a SIMD's matrix pipe.  So a
//    workgroup is 4 waves = one per SIMD, one workgroup per CU, up to 512 VGPRs per
//    lane, and every latency is hidden inside the wave's own instruction stream
//    (memory instructions are issued in the shadow of MFMAs already in the pipe).
//  * Operands are the PREPARED matrices (prep.hip), f32, zero padded, both K-major
//    (an NT GEMM): dictionary Y = MFMA A operand (accumulator rows), experimental X =
//    B operand (columns).  Both are stored tile/slab-blocked: a (128 patterns x 32
//    pixels) block is 16 KB contiguous and already in LDS order incl. bank swizzle
//    (kernels.h: prepared_offset), so a slab is a sequential burst of lane-linear
//    1 KB global_load_lds pieces.
//  * Workgroup tile = 128 dictionary x 256 experimental patterns; wave w owns columns
//    [64w, 64w+64) x all 128 rows = 8 accumulators (128 VGPRs).  With 32x32x2 the
//    accumulator column is lane&31: every lane owns ONE experimental pattern per 32-wide
//    column group and sees 16 dictionary candidates per accumulator in its registers,
//    so top-k is a lane-local streaming insertion into two sorted register lists.
//  * The workgroup is persistent: it belongs to one block of 256 experimental patterns
//    and draws dictionary tiles (ascending) from that block's counter (dynamic balance).
//  * Step = one 32-pixel slab = 128 MFMAs per wave (8192 pipe cycles).  LDS is a ring
//    of three 48 KB stages (16 KB dictionary slab + 32 KB experimental slab) filled by
//    global_load_lds_dwordx4 two slabs ahead (144 KB: the CU's LDS is there to be used).
//    ONE barrier per step, in the MIDDLE of the step: after it the next slab is
//    complete in LDS, so its first fragments are read during the second half of the
//    current step and no wave ever waits on LDS or memory at a step boundary.
//  * ds_read_b128 hands a lane 4 consecutive pixels of its row; MFMA j of a group
//    takes element j for both operands (lanes 0-31 pixel j, lanes 32-63 pixel 4+j):
//    a permutation of the summation order only.  24 LDS reads (16 A + 8 B) per 128 MFMAs.
//  * FORM = 1 is the opt-in split-f16 form (KPDI_COMPUTE_F16X2) of the same skeleton:
//    the 16-byte slots hold 8 float16 (high halves in slots 0-3 of a row-slab, low halves
//    in 4-7), a slab is 2 steps of 16 pixels with three v_mfma_f32_32x32x16_f16 per
//    accumulator (hi.hi + hi.lo + lo.hi), and the epilogue rescales by 2^-24.
//  * The opt-in plain float16 form (KPDI_COMPUTE_F16) has a kernel of its own: match16.hip.
//
// Algorithmic work per launch: 2 * M * n_chunk * K flops (K = kept pixels).
#include "match_device.h"
#include <stdlib.h>

namespace kpdi {

constexpr int SLAB_BYTES = TILE_DICT * TILE_K * 4;  // 128 patterns x 32 pixels: 16 KB
constexpr int STAGE_BYTES = 3 * SLAB_BYTES;          // dictionary slab + 2 experimental slabs
constexpr int NSTAGE = 3;
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;      // 144 KB ring (+ 32 B control words)

// One accumulator column group (32 patterns) of the epilogue: 64 candidates per lane by
// increasing dictionary index.  Steady state is ONE compare + branch per accumulator
// register: a candidate has to reach thr = max(shared bound, next float above the list's
// last entry), and only a register in which some lane does takes the exact path (valid
// row, multi-pass bound, insertion).  The register index r is a scalar loop counter
// (relative VGPR addressing), so there is one copy of the insertion code per accumulator.
template <int KMAX, bool BOUNDED, int FORM, int ROWT>
__device__ __forceinline__ void scan_tile(f32x16 (&acc)[4], float (&best)[KMAX], int (&best_idx)[KMAX],
                                          float gthr, float ub, int ub_idx, int row0, int n_valid,
                                          int idx_base) {
  // float16 operands (split form) are stored scaled by 2^12 each: the accumulators hold 2^24 * score
  constexpr float unscale = FORM != 0 ? 0x1p-24f : 1.f;
  // v > best[KMAX-1]  <=>  v >= nextafter(best[KMAX-1], +inf)   (scores are finite)
  float thr = fmaxf(gthr, next_up(best[KMAX - 1]));
#pragma unroll
  for (int rt = 0; rt < ROWT; ++rt) {
#pragma unroll 1
    for (int r = 0; r < 16; ++r) {
      const float v = acc[rt][r] * unscale + 0.f;  // -0 -> +0 so that ties compare as the merge does
      if (__builtin_amdgcn_ballot_w64(v >= thr) != 0) {
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
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
  }
}

// acc += A x B with the accumulator pinned to the accumulation registers ("a" class).
// Written as asm because with > 256 live registers the register allocator otherwise keeps
// parts of the accumulators in VGPRs and shuttles them (hundreds of v_accvgpr moves per
// step); accumulators in architectural VGPRs measured 4 % slower (they compete with the LDS
// returns and VALU for the VGPR ports).  `s_nop 1` covers the VALU-write -> MFMA-operand
// hazard, which hipcc does not pad for instructions inside an asm statement.
__device__ __forceinline__ void mfma_acc(f32x16 &c, float a, float b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// split-f16 form: acc += A x B for 16 pixels, A and B = 8 f16 per lane (one 16-byte LDS slot)
__device__ __forceinline__ void mfma_acc_h(f32x16 &c, const f32x4 &a, const f32x4 &b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// One of a wave's twelve 1 KB LDS-DMA pieces of a stage: 0-3 dictionary slab, 4-7 / 8-11
// the two experimental slabs.
// KPDI_DMA_BUFFER = 1 (default): `buffer_load_dwordx4 ... offen lds` - `gd` / `ge` are wave-uniform
// slab addresses (scalar registers, one buffer descriptor each), the lane's 16-byte offset is the
// ONE VGPR operand of every piece, and piece / slab / tile offsets travel in the scalar offset.
// KPDI_DMA_BUFFER = 0 keeps the `global_load_lds_dwordx4` form (a 64-bit per-lane address per
// piece) for comparison: with one wave per SIMD the issue of a piece blocks the wave's only
// instruction stream, and the global form did so for ~40 cycles - config 2 measured 21.58 ms
// (86.9 % of the f32 MFMA peak) against 20.90 ms (89.7 %) in the buffer form, which is what the
// timing-only ablation without any pieces had given (89.4 %).
#ifndef KPDI_DMA_BUFFER
#define KPDI_DMA_BUFFER 1
#endif
#if KPDI_DMA_BUFFER
__device__ __forceinline__ void issue_piece(const char *gd, const char *ge, size_t tile_bytes, char *stage_base,
                                            int wv, int p, unsigned goff) {
  const int c = p & 3, part = p >> 2;
  const int kb = (wv + 4 * c) * 1024;
  __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)(part == 0 ? gd : ge), 0, 0x7fffffff, 0x00020000);
  const int soffset = part == 0 ? kb : (int)((size_t)(part - 1) * tile_bytes) + kb;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(
      rsrc, (__attribute__((address_space(3))) void *)(stage_base + part * SLAB_BYTES + kb), 16, (int)goff, soffset,
      0, 0);
}
#define KPDI_GOFF_ARG , goff
#define KPDI_LANE_BASE 0u
#else
// `gd`/`ge` already include the lane's 16-byte offset.
__device__ __forceinline__ void issue_piece(const char *gd, const char *ge, size_t tile_bytes, char *stage_base,
                                            int wv, int p) {
  const int c = p & 3, part = p >> 2;
  const char *g = (part == 0 ? gd : ge + (size_t)(part - 1) * tile_bytes) + (wv + 4 * c) * 1024;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                   (__attribute__((address_space(3))) void *)(stage_base + part * SLAB_BYTES +
                                                                              (wv + 4 * c) * 1024),
                                   16, 0, 0);
}
#define KPDI_GOFF_ARG
#define KPDI_LANE_BASE goff
#endif

// ROWT = row tiles of 32 dictionary patterns per unit of work: 4 = a whole 128-pattern tile; 1 = the
// TAIL form (f32 only): the last n_tiles % nsplit tiles of a row block are handed out as quarter
// tiles, so that a launch whose tile count is a small non-multiple of its workgroups (a rank's
// share of a dictionary sharded over several GPUs) ends within a quarter tile-time of its even share.
// A quarter of a 16 KB slab block is its contiguous 4 KB piece q (rows 32q .. 32q + 31: slots
// 256q .. 256q + 255, same swizzle), i.e. ONE LDS-DMA piece per wave; everything else is the tile
// code with the row-tile loops cut to one.
template <int KMAX, bool BOUNDED, int FORM, int ROWT = 4>
__global__ __launch_bounds__(MATCH_THREADS, 1) void match_topk_kernel(MatchArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int sp, rb;
  block_rb_sp(a, blockIdx.x, &rb, &sp);
  if (rb >= a.rows) return;  // (a workgroup of the padding of the XCD grid: whole workgroup, before any barrier)
  rb += a.row_first;
  // kernel arguments into locals (nothing below takes the address of `a`)
  const float *a_dict = a.dict;
  const int n_tiles = a.n_tiles, n_valid = a.n_valid, idx_base = a.idx_base;
  const int row_base = a.row_base;
  unsigned *gthr_arr = a.gthr;
  const int kpad = a.kpad;
  const int nslab = kpad / TILE_K;
  // Tiles: a workgroup's first `fixed_draws` are fixed (sp, sp + nsplit, ...: the workgroups an XCD hosts walk
  // the same dictionary tiles at the same pace, see block_rb_sp), the last ones are drawn from the row block's
  // counter, which evens out the speeds at the end of the launch.
  unsigned *tile_ctr = a.tile_ctr + rb;
  const int fixed_draws = a.fixed_draws;
  int drawn = 3;  // t0, t1, t2 below are draws 0, 1, 2
  volatile int *ctrl = (volatile int *)(smem + LDS_BYTES);  // control words behind the ring

  const unsigned goff = (unsigned)lane * 16u;
  const size_t tile_bytes = (size_t)nslab * SLAB_BYTES;  // one 128-pattern dictionary tile, all slabs
  // the workgroup's 256 experimental patterns = prepared tiles 2*rb and 2*rb+1
  const char *exp_base = (const char *)a.exp + (size_t)rb * 2 * tile_bytes + KPDI_LANE_BASE;
  // wave wv's two column groups inside a stage: rows wv*64 + c*32 + (lane&31) of the 256
  const unsigned exp_frag = SLAB_BYTES + (wv >> 1) * SLAB_BYTES + ((wv & 1) * 2) * 4096;

  // LDS -> MFMA A fragments: lane l reads row (l&31) of a 32-row tile, pixel quad kg*2 + (l>>5)
  unsigned frag[4];
  {
    const int lr = lane & 31;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const int w = (((lane & 1) << 3) | (kg * 2 + (lane >> 5))) ^ ((lane >> 1) & 7);
      frag[kg] = (unsigned)((lr >> 1) * 256 + w * 16);
    }
  }

  // per-lane running best lists: pattern rb*256 + wv*64 + c*32 + (lane&31), c = 0, 1
  float best0[KMAX], best1[KMAX];
  int bidx0[KMAX], bidx1[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    best0[j] = best1[j] = -INFINITY;
    bidx0[j] = bidx1[j] = INT_MAX;
  }
  const int m_lane = rb * TILE_EXP + wv * 64 + (lane & 31);  // column group 0; group 1 = +32
  float ub0 = INFINITY, ub1 = INFINITY;
  int ubi0 = -1, ubi1 = -1;
  if (BOUNDED) {
    ub0 = a.bound_score[m_lane];
    ubi0 = a.bound_idx[m_lane];
    ub1 = a.bound_score[m_lane + 32];
    ubi1 = a.bound_idx[m_lane + 32];
  }
  // shared rejection bound (see shared_bound above): this lane's two pattern lines and slot
  const unsigned *line0 = gthr_arr + (size_t)m_lane * BOUND_SLOTS;
  const unsigned *line1 = line0 + 32 * BOUND_SLOTS;
  const int my_slot = (sp * 2 + (lane >> 5)) & (BOUND_SLOTS - 1);
  const int bound_rank = a.bound_rank;
  const bool bound_grouped = a.bound_grouped != 0;
  float g0 = -INFINITY, g1 = -INFINITY;


  // ---- dictionary tiles are handed out dynamically; t0 = tile being computed, t1/t2 the
  // next two (the loads run two slabs ahead, which can reach two tiles ahead)
  int t0, t1, t2;
  // The first three tiles of a workgroup are fixed (draws r, r + G, r + 2G for the r-th of the G
  // workgroups sharing the counter, which starts at 3G): workgroups do not start at the same
  // time, and when the early ones drew their look-ahead tiles from the counter a launch with
  // <= 3 tiles per workgroup took about one tile-time more than its share (4096 x 2048 / 4096 /
  // 6144 patterns: 1.00 / 1.48 / 1.96 ms before, 0.57 / 1.06 / 1.59 ms now).  Everything after
  // is drawn while the previous tile is computed.
  t0 = sp;
  t1 = sp + a.nsplit;
  t2 = sp + 2 * a.nsplit;
  if (t0 >= n_tiles) goto write_out;

  {
    const int last_tile = n_tiles - 1;
    // load cursor: the slab that is fetched next (two ahead of the one being computed)
    int ld_pos = 0, ld_slab = 0, ld_stage = 0;
    int fetched = 0;  // thread 0: tile number drawn during the current tile
    int tp = 0;       // tile parity: the drawn number is handed over through ctrl[4 + tp]
    const char *gd = nullptr, *ge = nullptr;
    const char *dict_base = (const char *)a_dict + KPDI_LANE_BASE;
    // (plain macros instead of lambdas: by-reference captures put the whole state in scratch)
#define KPDI_CURSOR_SET()                                                                        \
  {                                                                                              \
    int t_ = ld_pos == 0 ? t0 : (ld_pos == 1 ? t1 : t2);                                         \
    t_ = t_ < last_tile ? t_ : last_tile; /* past the end: harmless re-load, no branch */        \
    if (ROWT == 4) {                                                                             \
      gd = dict_base + (size_t)t_ * tile_bytes + (size_t)ld_slab * SLAB_BYTES;                   \
    } else { /* unit t_ = 32 rows from row_base + 32 t_: block (row >> 7), quarter (row >> 5) & 3 */ \
      const int row_ = row_base + 32 * t_;                                                       \
      gd = dict_base + (size_t)(row_ >> 7) * tile_bytes + (size_t)ld_slab * SLAB_BYTES + ((row_ >> 5) & 3) * 4096; \
    }                                                                                            \
    ge = exp_base + (size_t)ld_slab * SLAB_BYTES;                                                \
  }
#define KPDI_CURSOR_ADVANCE()                                  \
  {                                                            \
    if (++ld_slab == nslab) {                                  \
      ld_slab = 0;                                             \
      ++ld_pos;                                                \
    }                                                          \
    ld_stage = ld_stage == NSTAGE - 1 ? 0 : ld_stage + 1;      \
  }
    // ---- prologue: slabs 0 and 1 in flight, then everything landed and visible
    KPDI_CURSOR_SET();
    // (ROWT = 1: the dictionary part of a slab is its one 4 KB quarter = piece 0 of every wave)
#define KPDI_PIECE_LIVE(p) (ROWT == 4 || (p) == 0 || (p) >= 4)
#pragma unroll
    for (int p = 0; p < 12; ++p)
      if (KPDI_PIECE_LIVE(p)) issue_piece(gd, ge, tile_bytes, smem + ld_stage * STAGE_BYTES, wv, p KPDI_GOFF_ARG);
    KPDI_CURSOR_ADVANCE();
    KPDI_CURSOR_SET();
#pragma unroll
    for (int p = 0; p < 12; ++p)
      if (KPDI_PIECE_LIVE(p)) issue_piece(gd, ge, tile_bytes, smem + ld_stage * STAGE_BYTES, wv, p KPDI_GOFF_ARG);
    KPDI_CURSOR_ADVANCE();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // A (dictionary) and B (experimental) fragments, double buffered by pixel group
    // (split-f16: by 16-pixel step; [.][0..3] = the high halves of the 4 row tiles / 2 column
    // groups, [.][4..7] / [.][2..3] the low halves)
    constexpr bool SPLIT = FORM == 1;
    f32x4 fa[2][SPLIT ? 8 : 4], fb[2][SPLIT ? 4 : 2];
#pragma unroll
    for (int rt = 0; rt < ROWT; ++rt) fa[0][rt] = *(const f32x4 *)(smem + rt * 4096 + frag[0]);
#pragma unroll
    for (int c = 0; c < 2; ++c) fb[0][c] = *(const f32x4 *)(smem + exp_frag + c * 4096 + frag[0]);
    if (SPLIT) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) fa[0][4 + rt] = *(const f32x4 *)(smem + rt * 4096 + frag[2]);
#pragma unroll
      for (int c = 0; c < 2; ++c) fb[0][2 + c] = *(const f32x4 *)(smem + exp_frag + c * 4096 + frag[2]);
    }

    int stage = 0;
#pragma clang loop unroll(disable)
    for (;;) {  // dictionary tiles
      // accumulators are born here and die in this tile's epilogue: inside the slab loop
      // they are only ever touched by MFMAs (keeps them in the accumulation registers
      // instead of being shuttled between register classes every step)
      f32x16 acc0[4], acc1[4];
#pragma unroll
      for (int rt = 0; rt < ROWT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[rt][r] = acc1[rt][r] = 0.f;
#pragma clang loop unroll(disable)
      for (int slab = 0; slab < nslab; ++slab) {
      const char *ls = smem + stage * STAGE_BYTES;
      const int nstage = stage == NSTAGE - 1 ? 0 : stage + 1;
      const char *ls_next = smem + nstage * STAGE_BYTES;
      if (slab == 0) {
        if (tid == 0) fetched = drawn < fixed_draws ? sp + drawn * a.nsplit : (int)atomicAdd(tile_ctr, 1u);
        ++drawn;
      }
      if (slab == nslab - 1) {  // landed by the mid-step wait, used in the epilogue
        g0 = shared_bound<KMAX>(line0, bound_grouped);
        g1 = shared_bound<KMAX>(line1, bound_grouped);
      }
      KPDI_CURSOR_SET();
      char *ld_base = smem + ld_stage * STAGE_BYTES;

      if (SPLIT) {
        // ---- split-f16 slab: 2 steps of 16 pixels; per step and accumulator three MFMAs
        // hi.hi + hi.lo + lo.hi (the lo.lo term is below 2^-22 of the product).  The 16-byte
        // LDS slots hold 8 f16: slots 0-3 of a row = high halves of pixels 8q..8q+7, slots
        // 4-7 the low halves, so the fragment addresses are the f32 kernel's frag[0..3].
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const int cur = st;
          if (st == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (slab == 0 && tid == 0) ctrl[4 + tp] = fetched;
            __syncthreads();
          }
          // fragments of the next step: step 1 of this slab, or step 0 of the next slab
          const char *src_hi = st == 0 ? ls + frag[1] : ls_next + frag[0];
          const char *src_lo = st == 0 ? ls + frag[3] : ls_next + frag[2];
#pragma unroll
          for (int g = 0; g < 3; ++g) {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
              const f32x4 &av = fa[cur][g == 2 ? 4 + rt : rt];
              mfma_acc_h(acc0[rt], av, fb[cur][g == 1 ? 2 : 0]);
              mfma_acc_h(acc1[rt], av, fb[cur][g == 1 ? 3 : 1]);
              // in the shadow of these MFMAs: one fragment load and (second half of the slab,
              // after the barrier) one LDS-DMA piece of the slab two steps ahead
              const int slot = g * 4 + rt;  // 0..11
              if (slot < 4) fa[cur ^ 1][slot] = *(const f32x4 *)(src_hi + slot * 4096);
              else if (slot < 8) fa[cur ^ 1][slot] = *(const f32x4 *)(src_lo + (slot - 4) * 4096);
              else if (slot < 10) fb[cur ^ 1][slot - 8] = *(const f32x4 *)(src_hi + exp_frag + (slot - 8) * 4096);
              else fb[cur ^ 1][slot - 8] = *(const f32x4 *)(src_lo + exp_frag + (slot - 10) * 4096);
              if (st == 1) issue_piece(gd, ge, tile_bytes, ld_base, wv, slot KPDI_GOFF_ARG);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      } else {
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        const int cur = kg & 1;
        if (kg == 2) {
          // ---- the step's only synchronisation point.  Everything this wave issued during
          // the second half of the previous step has landed (its pieces of slab+1, the tile
          // counter); after the barrier (a) slab+1 is complete in LDS, (b) every wave is past
          // the previous step, whose stage is refilled below.
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (slab == 0 && tid == 0) ctrl[4 + tp] = fetched;
          __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int rt = 0; rt < ROWT; ++rt) {
            mfma_acc(acc0[rt], fa[cur][rt][j], fb[cur][0][j]);
            mfma_acc(acc1[rt], fa[cur][rt][j], fb[cur][1][j]);
          }
          // in the shadow of these 8 MFMAs (512 pipe cycles):
          // ... the fragments of the next pixel group (for kg = 3: of the next slab)
          if (j < 2) {
            const char *src = kg < 3 ? ls + frag[kg + 1] : ls_next + frag[0];
            if (2 * j < ROWT) fa[cur ^ 1][2 * j] = *(const f32x4 *)(src + (2 * j) * 4096);
            if (2 * j + 1 < ROWT) fa[cur ^ 1][2 * j + 1] = *(const f32x4 *)(src + (2 * j + 1) * 4096);
            fb[cur ^ 1][j] = *(const f32x4 *)(src + exp_frag + j * 4096);
          }
          // ... and the slab two steps ahead: this wave's 12 LDS-DMA pieces
          if (kg == 2) {
            if (KPDI_PIECE_LIVE(2 * j)) issue_piece(gd, ge, tile_bytes, ld_base, wv, 2 * j KPDI_GOFF_ARG);
            if (KPDI_PIECE_LIVE(2 * j + 1)) issue_piece(gd, ge, tile_bytes, ld_base, wv, 2 * j + 1 KPDI_GOFF_ARG);
          }
          if (kg == 3) issue_piece(gd, ge, tile_bytes, ld_base, wv, 8 + j KPDI_GOFF_ARG);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      }
      KPDI_CURSOR_ADVANCE();
      stage = nstage;
      }  // slabs
      // the last MFMAs (16 passes) must have written the accumulators before they are read
#pragma unroll
      for (int rt = 0; rt < ROWT; ++rt)
        asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc0[rt]), "+a"(acc1[rt]));
      {
        // ---- epilogue of the tile
        const int row0 = (ROWT == 4 ? t0 * TILE_DICT : row_base + 32 * t0) + 4 * (lane >> 5);
        float pub0 = best0[0], pub1 = best1[0];  // entry bound_rank-1 before the scan
#pragma unroll
        for (int j = 1; j < KMAX; ++j) {
          pub0 = j == bound_rank - 1 ? best0[j] : pub0;
          pub1 = j == bound_rank - 1 ? best1[j] : pub1;
        }
        scan_tile<KMAX, BOUNDED, FORM, ROWT>(acc0, best0, bidx0, g0, ub0, ubi0, row0, n_valid, idx_base);
        scan_tile<KMAX, BOUNDED, FORM, ROWT>(acc1, best1, bidx1, g1, ub1, ubi1, row0, n_valid, idx_base);
        // publish the list entry the bound is built from, if it rose
        float now0 = best0[0], now1 = best1[0];
#pragma unroll
        for (int j = 1; j < KMAX; ++j) {
          now0 = j == bound_rank - 1 ? best0[j] : now0;
          now1 = j == bound_rank - 1 ? best1[j] : now1;
        }
        if (now0 > pub0)
          __hip_atomic_fetch_max(const_cast<unsigned *>(line0) + my_slot, score_key(now0), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
        if (now1 > pub1)
          __hip_atomic_fetch_max(const_cast<unsigned *>(line1) + my_slot, score_key(now1), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
        t0 = t1;
        t1 = t2;
        // published at this tile's first barrier; the slot alternates so that a wave still in
        // its epilogue cannot see the next tile's number
        t2 = __builtin_amdgcn_readfirstlane(ctrl[4 + tp]);
        tp ^= 1;
        --ld_pos;
        if (t0 >= n_tiles) break;
      }
    }
  }

write_out : {
  const int lists = 2 * a.nsplit;
  const size_t o0 = ((size_t)m_lane * lists + (size_t)(sp * 2 + (lane >> 5))) * KMAX;
  const size_t o1 = ((size_t)(m_lane + 32) * lists + (size_t)(sp * 2 + (lane >> 5))) * KMAX;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    a.part_scores[o0 + j] = best0[j];
    a.part_idx[o0 + j] = bidx0[j];
    a.part_scores[o1 + j] = best1[j];
    a.part_idx[o1 + j] = bidx1[j];
  }
}
}

int match_list_len(int k) {
  if (k <= 1) return 1;
  if (k <= 8) return 8;
  if (k <= 20) return 20;
  return 32;
}

int match_blocks_per_cu() { return 1; }

template <int KMAX, bool BOUNDED, int FORM, int ROWT = 4>
static hipError_t launch_t(const MatchArgs &args, int grid, hipStream_t s) {
  // (the attribute belongs to the function ON A DEVICE: remembered per device, not per process - a second context
  // on another GPU would otherwise launch a 144 KB kernel without it)
  static unsigned long long attr_set = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (!((attr_set >> (dev & 63)) & 1ull)) {
    hipError_t e = hipFuncSetAttribute((const void *)match_topk_kernel<KMAX, BOUNDED, FORM, ROWT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 32);
    if (e != hipSuccess) return e;
    attr_set |= 1ull << (dev & 63);
  }
  hipLaunchKernelGGL((match_topk_kernel<KMAX, BOUNDED, FORM, ROWT>), dim3(grid), dim3(MATCH_THREADS), LDS_BYTES + 32,
                     s, args);
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
  g.idx_base = a.idx_base;
  g.row_first = a.row_first;
  g.part_scores = a.part_scores;
  g.part_idx = a.part_idx;
  g.bound_score = a.bound_score;
  g.bound_idx = a.bound_idx;
  g.gthr = a.gthr;
  g.bound_rank = a.bound_rank;
  g.bound_grouped = a.bound_grouped;
  g.tile_ctr = a.tile_ctr;
  g.tile_groups = 1;
  g.fixed_draws = a.fixed_draws < 3 ? 3 : a.fixed_draws;
  g.tail_first = a.n_tiles;
  g.tail_shift = 0;
  g.part_cnt = nullptr;
  g.perm_rounds = 0;  // (match.hip hands its tiles out in rising order: list_insert's tie rule relies on it)
  g.perm_stride = 1;
  g.epi_stats = nullptr;
  g.xcd_rows = a.xcd_rows;
  g.xcd_splits = a.xcd_splits;
  g.rows = a.rows;
  g.rows_grid = a.xcd_rows > 0 && a.rows_grid > a.rows ? a.rows_grid : a.rows;
  g.row_base = a.row_base;
  const int grid = g.rows_grid * a.nsplit;
  const bool bounded = a.bound_score != nullptr;
  if (a.operand_form == 2) return hipErrorInvalidValue;  // the float16 form has its own kernel: launch_match16
  if (a.row_tiles == 1) {  // tail form: f32, single pass
    if (a.operand_form != 0 || bounded || (a.row_base & 31)) return hipErrorInvalidValue;
    switch (a.list_len) {
      case 1: return launch_t<1, false, 0, 1>(g, grid, s);
      case 8: return launch_t<8, false, 0, 1>(g, grid, s);
      case 20: return launch_t<20, false, 0, 1>(g, grid, s);
      case 32: return launch_t<32, false, 0, 1>(g, grid, s);
      default: return hipErrorInvalidValue;
    }
  }
#define KPDI_CASE(K)                                                                             \
  case K:                                                                                        \
    if (a.operand_form == 1) return bounded ? launch_t<K, true, 1>(g, grid, s) : launch_t<K, false, 1>(g, grid, s); \
    return bounded ? launch_t<K, true, 0>(g, grid, s) : launch_t<K, false, 0>(g, grid, s);
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
