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
//    and experimental X (m_pad x kpad), f32, row-major with the pixel axis
//    contiguous, zero-padded.  Both are "K-major", i.e. an NT GEMM.
//  * The dictionary is the MFMA A operand (rows of the accumulator tile), the
//    experimental patterns are the B operand (columns).  With
//    v_mfma_f32_32x32x2_f32 the accumulator column is lane&31, so every lane owns
//    ONE experimental pattern per 32x32 tile and sees 16 dictionary candidates
//    for it in its registers: top-k becomes a lane-local streaming insertion with
//    no cross-lane traffic.
//  * Workgroup = 4 waves, tile = 128 dictionary x 128 experimental patterns;
//    wave w owns experimental columns [32w, 32w+32) and all 128 dictionary rows
//    (4 accumulator tiles = 64 VGPRs).  A workgroup is persistent over a
//    contiguous range of dictionary tiles ("split") and keeps its lanes' sorted
//    best-KMAX lists in registers for the whole sweep.
//  * HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip), two
//    32 KB stages; one barrier per 32-pixel slab.  The LDS image is lane-linear
//    (hardware rule), so the bank swizzle is applied to the per-lane SOURCE
//    address and again on the ds_read_b128 fragment reads: 16-byte slot
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

namespace kpdi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SLAB_BYTES = TILE_DICT * TILE_K * 4;  // 16 KB per operand per stage
constexpr int STAGE_BYTES = 2 * SLAB_BYTES;          // dictionary slab + experimental slab
constexpr int LDS_BYTES = 2 * STAGE_BYTES;           // double buffered: 64 KB -> 2 workgroups / CU

struct MatchArgs {
  const float *dict;
  const float *exp;
  int kpad, n_tiles, n_valid, nsplit, idx_base;
  float *part_scores;
  int *part_idx;
  const float *bound_score;
  const int *bound_idx;
};

// Insert (v, idx) into a descending sorted list; precondition v > s[KMAX-1].
// Equal scores keep arrival order (candidates arrive by increasing dictionary
// index), which is the engine's tie rule: lower dictionary index first.
// new s[j] = median(s[j-1], s[j], v) because s[j-1] >= s[j].
template <int KMAX>
__device__ __forceinline__ void list_insert(float (&s)[KMAX], int (&id)[KMAX], float v, int idx) {
#pragma unroll
  for (int j = KMAX - 1; j >= 1; --j) {
    const bool below = v > s[j];       // v ranks above entry j
    const bool below1 = v > s[j - 1];  // v ranks above entry j-1 as well -> shift
    id[j] = below ? (below1 ? id[j - 1] : idx) : id[j];
    s[j] = __builtin_amdgcn_fmed3f(s[j - 1], s[j], v);
  }
  id[0] = (v > s[0]) ? idx : id[0];
  s[0] = fmaxf(s[0], v);
}

template <int KMAX, bool BOUNDED>
__global__ __launch_bounds__(MATCH_THREADS, 2) void match_topk_kernel(MatchArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sp = blockIdx.x % a.nsplit;
  const int rb = blockIdx.x / a.nsplit;
  const int t0 = (int)(((int64_t)sp * a.n_tiles) / a.nsplit);
  const int t1 = (int)(((int64_t)(sp + 1) * a.n_tiles) / a.nsplit);
  const int kpad = a.kpad;
  const int nslab = kpad / TILE_K;
  const int nsteps = (t1 - t0) * nslab;

  // ---- global -> LDS staging: wave wv copies 1 KB pieces {wv, wv+4, wv+8, wv+12}
  // of each operand's 16 KB slab.  Piece c covers rows 8c..8c+7; LDS slot p (16 B
  // units) of the slab holds row = 2*(p>>4) + (w>>3), pixel quad kq = w&7 where
  // w = (p&15) ^ ((p>>4)&7).
  unsigned goff[4];  // byte offset of this lane's 16 B inside the operand tile (without k0)
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int piece = wv + 4 * c;
    const int rp = piece * 4 + (lane >> 4);
    const int w = (lane & 15) ^ (rp & 7);
    const int row = rp * 2 + (w >> 3);
    const int kq = w & 7;
    goff[c] = (unsigned)(row * kpad + kq * 4) * 4u;
  }
  const char *exp_base = (const char *)(a.exp + (size_t)rb * TILE_EXP * kpad);
  const size_t dict_tile_bytes = (size_t)TILE_DICT * kpad * 4;

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
  const unsigned exp_frag_base = SLAB_BYTES + wv * 4096;

  // ---- per-lane running best lists for experimental pattern rb*128 + wv*32 + (lane&31)
  float best[KMAX];
  int best_idx[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    best[j] = -INFINITY;
    best_idx[j] = INT_MAX;
  }
  float ub = INFINITY;
  int ub_idx = -1;
  if (BOUNDED) {
    const int m = rb * TILE_EXP + wv * 32 + (lane & 31);
    ub = a.bound_score[m];
    ub_idx = a.bound_idx[m];
  }

  f32x16 acc[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;

  if (nsteps <= 0) goto write_out;

  {
    // next slab to fetch
    int ld_tile = t0, ld_slab = 0;
    auto issue = [&](int stage) {
      const char *gd = (const char *)a.dict + (size_t)ld_tile * dict_tile_bytes + (size_t)ld_slab * (TILE_K * 4);
      const char *ge = exp_base + (size_t)ld_slab * (TILE_K * 4);
      char *ls = smem + stage * STAGE_BYTES;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(gd + goff[c]),
            (__attribute__((address_space(3))) void *)(ls + (wv + 4 * c) * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(ge + goff[c]),
            (__attribute__((address_space(3))) void *)(ls + SLAB_BYTES + (wv + 4 * c) * 1024), 16, 0, 0);
      }
      if (++ld_slab == nslab) {
        ld_slab = 0;
        ++ld_tile;
      }
    };

    issue(0);
    int tile = t0, slab = 0;
    for (int s = 0; s < nsteps; ++s) {
      const int stage = s & 1;
      // slab s has landed (this wave's pieces: vmcnt; the other waves': barrier) and
      // every wave is done reading the other stage (it was computed on in step s-1)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (s + 1 < nsteps) issue(stage ^ 1);

      const char *ls = smem + stage * STAGE_BYTES;
      // fragments of pixel group kg+1 are fetched while the 16 MFMAs of group kg run
      f32x4 fa[2][4], fb[2];
      fb[0] = *(const f32x4 *)(ls + exp_frag_base + frag[0]);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) fa[0][rt] = *(const f32x4 *)(ls + rt * 4096 + frag[0]);
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        const int cur = kg & 1;
        if (kg < 3) {
          fb[cur ^ 1] = *(const f32x4 *)(ls + exp_frag_base + frag[kg + 1]);
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) fa[cur ^ 1][rt] = *(const f32x4 *)(ls + rt * 4096 + frag[kg + 1]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int rt = 0; rt < 4; ++rt)
            acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][rt][j], fb[cur][j], acc[rt], 0, 0, 0);
      }

      if (++slab == nslab) {
        // ---- epilogue: 64 candidates per lane, by increasing dictionary index.
        // The register index r is a (scalar) loop counter: the accumulator element is
        // fetched with relative VGPR addressing, so there are 4 copies of the insertion
        // code instead of 64.
        const int row0 = tile * TILE_DICT + 4 * (lane >> 5);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
#pragma unroll 1
          for (int r = 0; r < 16; ++r) {
            const int lrow = row0 + rt * 32 + (r & 3) + 8 * (r >> 2);
            const float v = acc[rt][r] + 0.f;  // -0 -> +0 so that ties compare as the merge does
            const int idx = a.idx_base + lrow;
            bool ok = lrow < a.n_valid;
            if (BOUNDED) ok = ok && (v < ub || (v == ub && idx > ub_idx));
            if (ok && v > best[KMAX - 1]) list_insert<KMAX>(best, best_idx, v, idx);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
        }
        slab = 0;
        ++tile;
      }
    }
  }

write_out : {
  const int m = rb * TILE_EXP + wv * 32 + (lane & 31);
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
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((match_topk_kernel<KMAX, BOUNDED>), dim3(grid), dim3(MATCH_THREADS), LDS_BYTES, s, args);
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
  g.part_scores = a.part_scores;
  g.part_idx = a.part_idx;
  g.bound_score = a.bound_score;
  g.bound_idx = a.bound_idx;
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
