// tailgemm.hip - the LAST PARTIAL ROUND of a sweep on the float32 form of match16.hip, as a kernel of its own.
//
// match16.hip hands a row block's dictionary tiles (256 patterns) out to `nsplit` persistent workgroups in whole rounds.
// What is left after the last whole round - n_tiles % nsplit tiles - used to run inside that kernel as halves / quarters
// of a tile; when it is ONE tile per 16 workgroups (one rank's share of configs[1] at N = 8: 48.8 tiles over 16 splits)
// a quarter of the workgroups then worked for a quarter of a tile-time while the others idled: 0.31 tile-times for
// 0.05 of work.  Units of a sixteenth of a tile inside that kernel did not help (profiles/r06_fine_tail_units.txt): its
// ring of three 48 KB stages gives a load ONE step to land, and a step of a small unit (768 matrix-pipe cycles) is
// shorter than the latency of an LDS-DMA piece under load (~4000 cycles) - and the guards cost the main loop 5 %.
//
// So the left-over rows get what small units need - a DEEP pipeline - here:
//   * workgroup = 32 dictionary rows x 128 experimental patterns (wave w: one 32 x 32 accumulator against column group w),
//     so 212 left-over rows x 4096 patterns are 7 x 32 = 224 workgroups: the whole chip, one round;
//   * a step = the 24 pixels of one (tile, step) block of the plane-major operands (prep_device.h: half_slot, form 3): a wave
//     loads what IT needs - 3 x 1 KB of its column group + 3 x 1 KB of the dictionary rows - into its own part of LDS, six
//     lane-linear LDS-DMA pieces per step.  The dictionary pieces are thereby loaded four times per workgroup (from L2:
//     9 KB per step) - and NO wave ever waits for another: no barrier in the kernel (one per step cost a quarter of the
//     matrix pipe's time; without it 212 rows x 4096 patterns take 60-63 us, profiles/r06_share8_kernel_stats.csv);
//   * a ring of SIX stages of 4 x 6 KB, loads running FIVE steps ahead (5 x 768 cycles of matrix work hide the latency),
//     one `s_waitcnt vmcnt` per step, the fragments of step s + 1 read while the MFMAs of step s run;
//   * NO fused top-k: the 32 x 128 scores go to a small matrix S[row][pattern] (212 x 4096 floats = 3.5 MB); by then the
//     main kernel has finished and the shared rejection bound of every pattern is FINAL, so tail_select_kernel (16
//     patterns x 16 row classes per workgroup, coalesced over patterns, candidates collected in LDS) passes a handful of
//     the rows to a sorted list that joins the merge as a source of its own (sweep.hip).
// Arithmetic: the same `v_mfma_f32_32x32x2_f32` sequence per accumulator as match16.hip (plane by plane, element j of the
// two 16-byte fragments feeds MFMA j), so every score is bit for bit what the main kernel would have produced - results
// do not depend on which kernel took a row (tests/test_gpu_engine.py: the two f32 kernels agree; chunking invariance).
//
// Reference semantics: SimilarityMetric.match + argtopk / topk of the chunk's last rows
// (indexing/_dictionary_indexing.py:193-203).  Algorithmic work: 2 * M * rows * K flops.
#include "match_device.h"

namespace kpdi {

constexpr int TG_STAGES = 6, TG_AHEAD = 5, TG_WAVE = 6144, TG_STAGE = 4 * TG_WAVE, TG_BLOCK = F16_TILE * F16_STEP * 2;  // 24 KB (tile, step) block
constexpr int TG_PIECES = 6;  // LDS-DMA pieces per wave and step
constexpr int TG_COLS = 128;  // experimental patterns per workgroup

struct TailGemmArgs {
  const char *dict, *exp;
  int nsteps;        // kpad / 24
  int tile_first;    // dictionary tile of row group 0
  int col_blocks;    // m_pad / 128
  int m_pad;
  float *scores;     // [row_groups * 32][m_pad]
};

__global__ __launch_bounds__(256, 1) void tail_gemm_kernel(TailGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // workgroups that share a column block are `col_blocks` apart: the same XCD (block b runs on XCD b % 8; used for speed
  // only) whenever col_blocks is a multiple of 8 - its experimental tile then crosses the fabric once
  const int cb = blockIdx.x % a.col_blocks, g = blockIdx.x / a.col_blocks;
  const int nsteps = a.nsteps;
  const char *dtile = a.dict + (size_t)(a.tile_first + (g >> 3)) * nsteps * TG_BLOCK + (size_t)(g & 7) * 1024;
  const char *etile = a.exp + (size_t)(cb >> 1) * nsteps * TG_BLOCK + (size_t)(((cb & 1) << 2) + wv) * 1024;
  const unsigned goff = (unsigned)lane * 16u;
  char *wbase = smem + wv * TG_WAVE;  // this wave's 6 KB of every stage: [3 planes of its column group][3 planes of the dictionary rows]

  // piece i of this wave's six 1 KB pieces of step s into stage `st`: i < 3: experimental plane i, else dictionary plane i - 3
  auto issue = [&](int i, int s, int st) {
    char *base = wbase + st * TG_STAGE;
    const char *src = (i < 3 ? etile : dtile) + (size_t)s * TG_BLOCK;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(base + i * 1024), 16, (int)goff,
                                             (i < 3 ? i : i - 3) * 8192, 0, 0);
  };
  // lane l reads row (l & 31) of a 32-row group: the 16-byte half (l >> 5) of the row's 32 bytes of a plane, halves
  // swapped for rows with bit 3 set (match16.hip: fa_lane)
  const unsigned fl = (unsigned)((lane & 31) * 32) + (unsigned)((((lane >> 5) ^ (lane >> 3)) & 1) * 16);
  f32x4 fa[3], fb[3], na[3], nb[3];
  // fragment read i of a stage: i even = dictionary plane i / 2, odd = experimental plane i / 2
  auto read = [&](int i, int st, f32x4 (&A)[3], f32x4 (&B)[3]) {
    const char *base = wbase + st * TG_STAGE;
    if (i & 1)
      B[i >> 1] = *(const f32x4 *)(base + (i >> 1) * 1024 + fl);
    else
      A[i >> 1] = *(const f32x4 *)(base + (3 + (i >> 1)) * 1024 + fl);
  };

  const int last = nsteps - 1;
#pragma unroll
  for (int s = 0; s < TG_AHEAD; ++s)
#pragma unroll
    for (int i = 0; i < TG_PIECES; ++i) issue(i, s < last ? s : last, s);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TG_PIECES * (TG_AHEAD - 1)) : "memory");  // step 0 has landed (this wave's own pieces: no barrier)
#pragma unroll
  for (int i = 0; i < 6; ++i) read(i, 0, fa, fb);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int st_cur = 0;
#pragma clang loop unroll(disable)
  for (int s = 0; s < nsteps; ++s) {
    const int st_next = st_cur == TG_STAGES - 1 ? 0 : st_cur + 1;
    const int st_prev = st_cur == 0 ? TG_STAGES - 1 : st_cur - 1;
    const int sl = s + TG_AHEAD < last ? s + TG_AHEAD : last;  // past the end the last step is loaded again - harmless, and the
                                                               // wait counts stay what they are
    // The step's 12 MFMAs (one accumulator: a dependent chain, 64 cycles each) with everything else in their shadow:
    //   behind MFMA 0..5: this wave's six pieces of step s + AHEAD into the stage of step s - 1, whose fragments it read
    //     during iteration s - 2 (its own part of LDS: program order is all the synchronisation there is);
    //   behind MFMA 5: its pieces of step s + 1 have landed (all but the newest AHEAD - 1 steps' pieces);
    //   behind MFMA 6..11: the six fragment reads of step s + 1.
#pragma unroll
    for (int slot = 0; slot < 12; ++slot) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot >> 2][slot & 3], fb[slot >> 2][slot & 3], acc, 0, 0, 0);
      if (slot < TG_PIECES) issue(slot, sl, st_prev);
      if (slot == TG_PIECES - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TG_PIECES * (TG_AHEAD - 1)) : "memory");
      if (slot >= 6) read(slot - 6, st_next, na, nb);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      fa[p] = na[p];
      fb[p] = nb[p];
    }
    st_cur = st_next;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the harmless re-loads: nothing of this kernel may land after it ends)
  // accumulator register r of lane l: dictionary row (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the group, pattern l & 31
  float *out = a.scores + (size_t)(g * 32 + 4 * (lane >> 5)) * a.m_pad + (size_t)cb * TG_COLS + wv * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) out[(size_t)((r & 3) + 8 * (r >> 2)) * a.m_pad] = acc[r];
}

hipError_t launch_tail_gemm(const TailGemmLaunch &a, hipStream_t s) {
  if (a.row_groups <= 0) return hipSuccess;
  if (a.kpad % (F16_STEP / 2) != 0 || a.m_pad % F16_TILE != 0) return hipErrorInvalidValue;
  static unsigned long long attr_set = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (!((attr_set >> (dev & 63)) & 1ull)) {
    hipError_t e = hipFuncSetAttribute((const void *)tail_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       TG_STAGES * TG_STAGE);
    if (e != hipSuccess) return e;
    attr_set |= 1ull << (dev & 63);
  }
  TailGemmArgs g;
  g.dict = (const char *)a.dict;
  g.exp = (const char *)a.exp;
  g.nsteps = a.kpad / (F16_STEP / 2);
  g.tile_first = a.tile_first;
  g.col_blocks = a.m_pad / TG_COLS;
  g.m_pad = a.m_pad;
  g.scores = a.scores;
  hipLaunchKernelGGL(tail_gemm_kernel, dim3(a.row_groups * g.col_blocks), dim3(256), TG_STAGES * TG_STAGE, s, g);
  return hipGetLastError();
}

// ---- S[row][pattern] -> one sorted list per pattern: the rows that reach the pattern's (final) shared bound
struct TailSelectArgs {
  const float *scores;
  int rows, m, m_pad, idx_first;
  const unsigned *gthr;
  int grouped;
  float *out_s;
  int *out_i;
};

// A workgroup = 16 patterns x 16 interleaved row classes (thread t: pattern t & 15, rows t >> 4, + 16, ...), eight loads in
// flight per thread - a pattern's rows as ONE dependent load chain took 80 us for 3.5 MB.  Rows that reach the bound
// (typically one or two per pattern) are appended to the pattern's candidate array in LDS; then one thread per pattern
// sorts them into the list by (score, index).  More than TS_CAP of them (a first chunk with no bound yet and a long
// tail; adversarial data): that thread walks the pattern's rows itself - slow, exact.  Out: [m][KMAX], one list per pattern.
constexpr int TS_PAT = 16, TS_CLS = 16, TS_CAP = 64;
template <int KMAX>
__global__ __launch_bounds__(TS_PAT * TS_CLS) void tail_select_kernel(TailSelectArgs a) {
  __shared__ float cs[TS_PAT][TS_CAP];
  __shared__ int ci[TS_PAT][TS_CAP];
  __shared__ int cn[TS_PAT];
  const int pl = threadIdx.x & (TS_PAT - 1), q = threadIdx.x / TS_PAT;
  const int m = blockIdx.x * TS_PAT + pl;
  if (threadIdx.x < TS_PAT) cn[threadIdx.x] = 0;
  __syncthreads();
  const bool live = m < a.m;
  // a score below the bound has KMAX better ones among the pattern's lists (match_device.h: shared_bound)
  const float t = live ? shared_bound<KMAX>(a.gthr + (size_t)m * BOUND_SLOTS, a.grouped != 0) : INFINITY;
  const float *p = a.scores + (live ? m : 0);
  for (int r0 = q; r0 < a.rows; r0 += 8 * TS_CLS) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = r0 + e * TS_CLS;
      v[e] = r < a.rows ? p[(size_t)r * a.m_pad] + 0.f : -INFINITY;  // -0 -> +0 so that ties compare as the merge does
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (live && v[e] >= t && v[e] > -INFINITY) {
        const int slot = atomicAdd(&cn[pl], 1);
        if (slot < TS_CAP) {
          cs[pl][slot] = v[e];
          ci[pl][slot] = a.idx_first + r0 + e * TS_CLS;
        }
      }
  }
  __syncthreads();
  if (threadIdx.x >= TS_PAT || !live) return;
  float best[KMAX];
  int bidx[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    best[j] = -INFINITY;
    bidx[j] = INT_MAX;
  }
  const int n = cn[pl];
  if (n <= TS_CAP) {
    for (int i = 0; i < n; ++i) {  // (any order of arrival: the position is decided by (score, index))
      const float v = cs[pl][i];
      const int id = ci[pl][i];
      if (ranks_before(v, id, best[KMAX - 1], bidx[KMAX - 1])) list_insert_lex<KMAX>(best, bidx, v, id);
    }
  } else {
    for (int r = 0; r < a.rows; ++r) {  // rows by rising dictionary index: list_insert's tie rule
      const float v = p[(size_t)r * a.m_pad] + 0.f;
      if (v >= t && v > best[KMAX - 1]) list_insert<KMAX>(best, bidx, v, a.idx_first + r);
    }
  }
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    a.out_s[(size_t)m * KMAX + j] = best[j];
    a.out_i[(size_t)m * KMAX + j] = bidx[j];
  }
}

int tail_select_lists() { return 1; }

hipError_t launch_tail_select(const TailSelectLaunch &a, hipStream_t s) {
  if (a.m <= 0) return hipSuccess;
  TailSelectArgs g;
  g.scores = a.scores;
  g.rows = a.rows;
  g.m = a.m;
  g.m_pad = a.m_pad;
  g.idx_first = a.idx_first;
  g.gthr = a.gthr;
  g.grouped = a.bound_grouped;
  g.out_s = a.out_scores;
  g.out_i = a.out_idx;
  const dim3 grid((a.m + TS_PAT - 1) / TS_PAT), block(TS_PAT * TS_CLS);
  switch (a.list_len) {
    case 1: hipLaunchKernelGGL(tail_select_kernel<1>, grid, block, 0, s, g); break;
    case 8: hipLaunchKernelGGL(tail_select_kernel<8>, grid, block, 0, s, g); break;
    case 20: hipLaunchKernelGGL(tail_select_kernel<20>, grid, block, 0, s, g); break;
    case 32: hipLaunchKernelGGL(tail_select_kernel<32>, grid, block, 0, s, g); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace kpdi
