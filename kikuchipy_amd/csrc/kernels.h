// Internal launcher prototypes shared by the .hip translation units of libkpdi.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace kpdi {

// ---- tile geometry of the match kernel (match.hip) -------------------------
constexpr int TILE_DICT = 128;  // dictionary patterns per tile (MFMA A operand, rows)
constexpr int TILE_EXP = 256;   // experimental patterns per workgroup tile (MFMA B operand, columns)
constexpr int TILE_K = 32;      // pixels per LDS slab
constexpr int MATCH_THREADS = 256;
constexpr int KMAX_LIMIT = 32;  // longest register-resident list of one pass

// ---- the float16 form (KPDI_COMPUTE_F16) has its own kernel and layout (match16.hip, prep_device.h: half_slot)
// Two variants of ONE geometry (256 x 256 workgroup tiles, operand steps of 48 pixels): 8 waves = two per SIMD (default),
// 4 waves = one per SIMD with a 128 x 128 wave tile (KPDI_F16_WAVES=4; also the skeleton of the f32 form).  Operands and
// lists are the same for both: a change of the variant does not change the prepared layout.
constexpr int F16_TILE = 256;     // experimental patterns per tile; dictionary patterns per tile of the 8-wave variant
constexpr int F16_STEP = 48;      // pixels (float16) per LDS step of the 8-wave variant
struct F16Geometry {
  int waves;      // 8 or 4
  int dict_tile;  // dictionary patterns per tile: 32 * 4 * (waves / 4)
  int step;       // pixels per step
};
inline F16Geometry f16_geometry(int waves) { return F16Geometry{waves == 4 ? 4 : 8, F16_TILE, F16_STEP}; }

inline int round_up(int64_t v, int64_t m) { return (int)(((v + m - 1) / m) * m); }

// ---- layout of a PREPARED pattern matrix (what prep.hip writes and match.hip reads).
// Rows (patterns) are grouped in tiles of 128, the pixel axis in slabs of TILE_K = 32.
// One (tile, slab) block = 128 x 32 floats = 16 KB is CONTIGUOUS in memory and is
// already the LDS image of match.hip (bank swizzle included), so a slab is fetched with
// 16 lane-linear 1 KB global_load_lds pieces = one sequential 16 KB burst:
//   block(tile, slab) at float offset (tile * nslab + slab) * 4096
//   inside: 16-byte slot p = (row>>1)*16 + (((row&1)<<3 | kq) ^ ((row>>1)&7)), kq = (c&31)>>2
// Returns the float offset of element (row r, padded pixel c).
__host__ __device__ inline size_t prepared_offset(int r, int c, int nslab) {
  const int tile = r >> 7, row = r & 127, slab = c >> 5, kq = (c >> 2) & 7;
  const int rp = row >> 1;
  const int slot = rp * 16 + ((((row & 1) << 3) | kq) ^ (rp & 7));
  return ((size_t)tile * nslab + slab) * 4096 + (size_t)slot * 4 + (c & 3);
}

// list length (template instantiation) used for a pass that needs `k` entries
int match_list_len(int k);

struct MatchLaunch {
  const float *dict;  // (n_dict_pad, kpad) prepared, zero rows beyond n_valid
  const float *exp;   // (m_pad, kpad) prepared, zero rows beyond M
  int kpad;           // multiple of TILE_K
  int n_tiles;        // n_dict_pad / TILE_DICT
  int n_valid;        // valid dictionary patterns in this chunk
  int m_pad;          // multiple of TILE_EXP
  int nsplit;         // dictionary splits (lists per pattern = 2 * nsplit)
  int row_first;      // first 256-pattern row block of this launch
  int rows;           // row blocks of this launch (grid = rows * nsplit)
  int idx_base;       // dictionary index of chunk row 0
  int list_len;       // KMAX of the instantiation
  float *part_scores; // [m_pad][2*nsplit][list_len]
  int *part_idx;
  int *part_cnt = nullptr;  // match16.hip: [m_pad][4*nsplit] entries every list holds - only those are written (the "no entry"
                            // tails were 9 of 10 of the scattered stores all workgroups issue behind their last tile)
  // optional upper bound per experimental pattern (multi-pass for keep_n > 32):
  // only candidates strictly after (bound_score, bound_idx) in the ranking count
  const float *bound_score; // [m_pad] or nullptr
  const int *bound_idx;
  // [m_pad][BOUND_SLOTS] published list ranks (see match.hip: shared_bound); prepared by
  // launch_init_bound at the start of a sweep (and of every bounded pass)
  unsigned *gthr;
  int bound_rank, bound_grouped;  // from bound_plan()
  unsigned *tile_ctr;  // [m_pad / TILE_EXP] dynamic tile counters, set to fixed_draws * nsplit before every launch
  int tile_groups;     // (unused)
  int fixed_draws = 3; // tiles per workgroup that are fixed (sp + j * nsplit) before it draws from the counter
  int xcd_rows = 0, xcd_splits = 0;  // XCD grid over a launch's (row block, split) workgroups, 0 = plain mapping
  int rows_grid = 0;                 // rows rounded up to a multiple of xcd_rows (0 = rows): the launch has rows_grid * nsplit
                                     // workgroups, those of the row blocks [rows, rows_grid) leave at once
  int operand_form;    // 0 = f32, 1 = split-f16 (KPDI_COMPUTE_F16X2), 2 = f16 (KPDI_COMPUTE_F16), see match.hip
  int row_tiles = 4;   // 4: units of work = 128-pattern tiles; 1: the tail form - 32-pattern units (f32 only)
  int row_base = 0;    // tail form: dictionary row (of this chunk) of unit 0, a multiple of 32; `n_tiles` counts units
  int tail_first = 0, tail_shift = 0;  // float32 form of match16.hip: partial units for the tiles from tail_first on
  int perm_rounds = 0, perm_stride = 1;  // match16.hip: order of a workgroup's tiles (match_device.h: MatchArgs)
  unsigned long long *epi_stats = nullptr;  // match16.hip: 4 device counters of what the epilogues did, or nullptr
};
constexpr unsigned THRESHOLD_NONE = 0x007fffffu;  // key of -inf
constexpr int BOUND_SLOTS = 32;
// how the lists of one pattern share their rejection bound: lists = 2 * nsplit
inline void bound_plan(int lists, int list_len, int *rank, int *grouped, int *used_slots) {
  if (lists >= BOUND_SLOTS) {
    *rank = 1;
    *grouped = 1;
    *used_slots = BOUND_SLOTS;
  } else {
    *rank = (list_len + lists - 1) / lists;  // rank * lists >= list_len
    *grouped = 0;
    *used_slots = lists;
  }
}
// slots [0, used) <- key(-inf), slots [used, 32) <- key(+inf), for m_pad patterns
hipError_t launch_init_bound(unsigned *gthr, int m_pad, int used_slots, hipStream_t s);
hipError_t launch_match(const MatchLaunch &a, hipStream_t s);
// float16 form: tiles of F16_TILE dictionary patterns, lists per pattern = 4 * nsplit;
// `list_scratch`: match16_scratch_bytes(grid, list_len) bytes (the lists' home during the launch)
hipError_t launch_match16(const MatchLaunch &a, int waves, void *list_scratch, hipStream_t s);
size_t match16_scratch_bytes(int grid, int waves, int list_len);
int match_blocks_per_cu();

// ---- the last partial round of a float32 sweep on match16.hip's operands, as a kernel of its own (tailgemm.hip)
struct TailGemmLaunch {
  const float *dict;  // prepared dictionary chunk (operand form 3), whole matrix
  const float *exp;   // prepared experimental matrix (form 3)
  int kpad;           // floats per prepared row (a multiple of 24)
  int tile_first;     // dictionary tile (256 patterns) that holds row group 0
  int row_groups;     // 32-row groups to compute, from row tile_first * 256 on
  int m_pad;
  float *scores;      // out: [row_groups * 32][m_pad]
};
hipError_t launch_tail_gemm(const TailGemmLaunch &a, hipStream_t s);
struct TailSelectLaunch {
  const float *scores;  // [rows ...][m_pad] of launch_tail_gemm
  int rows;             // valid rows (dictionary patterns) of it
  int m, m_pad;
  int idx_first;        // dictionary index (or coalesced row) of row 0
  const unsigned *gthr; // the shared bound, final for this chunk
  int bound_grouped, list_len;
  float *out_scores;    // [m][tail_select_lists()][list_len]
  int *out_idx;
};
hipError_t launch_tail_select(const TailSelectLaunch &a, hipStream_t s);
int tail_select_lists();  // sorted lists per pattern it writes

// ---- pattern preparation (prep.hip): cast -> gather rows/pixels -> normalise --
struct PrepLaunch {
  const void *raw;     // (n_rows_in, npix) of `dtype`
  int dtype;
  int npix;            // sy*sx
  const int *row_map;  // [n_out] source row per output row, or nullptr (identity)
  const int *pix_map;  // [k] kept pixel indices, or nullptr (all pixels)
  const unsigned *quad_desc = nullptr;  // [ceil(k / 4)] gather_descriptors() of pix_map, or nullptr
  int k;               // kept pixels
  int kpad;
  int n_out;           // rows to produce
  int metric;          // KPDI_METRIC_*
  float *out;          // (>= n_out, kpad)
  int operand_form;    // 0 = f32; 1 = split-f16 (KPDI_COMPUTE_F16X2); 2 = f16 (KPDI_COMPUTE_F16): `kpad` then
                       // counts pairs of pixels (the row holds 2 * kpad float16)
  int f16_rows = F16_TILE, f16_step = F16_STEP;  // float16 form: patterns per tile / pixels per step of the layout
};
hipError_t launch_prep(const PrepLaunch &a, hipStream_t s);
// host: the signal mask's pixel map as one descriptor per 4 kept pixels for the gather kernels of prep.hip - the quad's
// pixels as (up to) two runs of consecutive detector pixels: bits 0-11 = detector pixel of element 0, bits 12-23 =
// detector pixel of element j MINUS j (so that element e >= j is the e-th float behind it), bits 24-26 = j (4: one run).
// Returns false when some quad needs more than two runs or npix > 4096 (descriptors unusable).
bool gather_descriptors(const int *pix_map, int k, int npix, std::vector<unsigned> *out);
// in place: prepared f32 rows [0, n_rows_pad) x kpad -> split-f16 form (KPDI_COMPUTE_F16X2): every
// 128-byte row-slab (32 pixels) becomes 4 slots of high halves + 4 slots of low halves of
// 2^12 * value, eight f16 pixels per 16-byte slot; n_rows_pad multiple of 128
hipError_t launch_split_f16(float *prepared, int n_rows_pad, int kpad, hipStream_t s);

// ---- top-k merge (merge.hip) -------------------------------------------------
// Several small dictionary chunks swept as ONE launch (sweep.hip: coalescing) sit row after row in one prepared matrix;
// the match kernel then ranks by the ROW of that matrix, and the merge translates rows to dictionary indices: segment t
// holds the rows [row0[t], row0[t + 1]) and a row r of it is dictionary pattern r + delta[t].  Rows and dictionary
// indices both increase from segment to segment, so "lower row first" among equal scores is "lower index first".
constexpr int INDEX_SEGMENTS = 16;
struct IndexSegments {
  int n = 0;                    // 0: the sources hold dictionary indices already
  int row0[INDEX_SEGMENTS];     // first row of segment t (INT_MAX: unused)
  int delta[INDEX_SEGMENTS];    // dictionary index of a row of segment t = row + delta[t]
};
struct MergeLaunch {
  int m;                // experimental patterns
  int k;                // entries to produce per pattern
  // up to 3 candidate sources, each [m][lists][len] (stride between patterns = lists*len)
  const float *src_scores[3];
  const int *src_idx[3];
  int src_lists[3];
  int src_len[3];
  int src_row_stride[3];  // elements between consecutive patterns
  int src_list_stride[3]; // elements between consecutive lists of one pattern
  const int *src_cnt[3] = {nullptr, nullptr, nullptr};  // [m][lists] valid entries per list (what lies behind them was never
                                                        // written), or nullptr = every entry is valid
  int n_src;
  float *out_scores;    // [m][out_stride] (must not alias a source)
  int *out_idx;
  int out_stride;
  int out_offset;       // first output column
  IndexSegments seg;    // seg.n > 0: the sources in `seg_sources` (bit j = source j) hold ROWS of a coalesced matrix
  unsigned seg_sources = 0;
};
hipError_t launch_merge(const MergeLaunch &a, hipStream_t s);
hipError_t launch_fill_topk(float *scores, int *idx, int64_t n, hipStream_t s);
hipError_t launch_fill_u32(unsigned *p, unsigned value, int64_t n, hipStream_t s);
// ONE launch for all the small initialisations in front of a sweep (running lists, shared bound, tile counters, the zero
// rows behind a partial tile): up to FILL_SEGMENTS word ranges, each filled with a constant, or (bound_used >= 0) with
// the shared bound's pattern - slot (i % BOUND_SLOTS) < bound_used ? key(-inf) : key(+inf)
constexpr int FILL_SEGMENTS = 8;
struct FillSegments {
  unsigned *p[FILL_SEGMENTS];
  unsigned long long words[FILL_SEGMENTS];
  unsigned value[FILL_SEGMENTS];
  int bound_used[FILL_SEGMENTS];
  int n = 0;
};
hipError_t launch_fill_segments(const FillSegments &f, hipStream_t s);
hipError_t launch_last_column(const float *scores, const int *idx, int m, int stride, int col,
                              float *bound_score, int *bound_idx, hipStream_t s);

// ---- float64 arithmetic: rescoring of screened candidates (rescore.hip) -------
struct RescoreLaunch {
  const void *exp_raw;   // the experimental patterns as handed over (after recorded pre-processing), [m_all][npix]
  int exp_dtype;
  const int *row_map;    // kept pattern -> source row (navigation mask), or nullptr
  const void *dict_raw;  // the raw dictionary chunk [n_chunk][npix]
  int dict_dtype;
  int64_t n_chunk, global_start;
  const int *pix_map;    // kept pixel -> detector pixel (signal mask), or nullptr
  int k, npix, metric;   // kept pixels, detector pixels, KPDI_METRIC_*
  int m;
  const float *cand_s;   // screened candidates [m][cand_stride], this pass: columns cand_offset .. + n_cand
  const int *cand_i;     // global dictionary indices (INT_MAX: none)
  int cand_stride, cand_offset, n_cand;
  double *cand_s64;      // out, same layout as cand_s
  unsigned *max_diff;    // bits of the largest |f32 score - f64 score| seen (atomic max)
};
hipError_t launch_rescore(const RescoreLaunch &a, hipStream_t s);

struct Merge64Launch {
  int m, k;
  const double *run_s;   // running best-k [m][k] or nullptr
  const int *run_i;
  const double *cand_s64;  // `lists` lists of `len` entries per pattern
  const int *cand_i;
  int lists, len;
  int64_t row_stride, list_stride;
  double *out_s;         // [m][k]; may alias run_s / run_i
  int *out_i;
  // certification (uncertified == nullptr: none): the f32 score of the last screened candidate of a pattern
  const float *cand_s32;
  int s32_stride, s32_col;
  int enumerated_all;    // every pattern of the chunk has been rescored
  const unsigned *max_diff;
  float eps_floor;
  int *uncertified;      // counter (atomic add)
};
hipError_t launch_merge64(const Merge64Launch &a, hipStream_t s);
hipError_t launch_fill_topk64(double *scores, int *idx, int64_t n, hipStream_t s);

// ---- pattern pre-processing (preproc.hip) ---------------------------------
// static background -> dynamic background -> (optionally) the metric's preparation of the
// patterns, as ONE kernel per pattern set when the detector fits the LDS (<= 19 200 pixels),
// else as streaming kernels (any size; the preparation then runs through launch_prep).
constexpr int CONV_R = 8;  // outputs per thread of the register-tiled 1-D correlations
struct PreLaunch {
  void *patterns; int dtype; int64_t n; int sy, sx;  // all patterns of the set, processed in place
  int do_static;
  const float *bg;   // sy*sx
  float bg_min, bg_max;
  int st_operation, scale_bg;
  int do_dynamic;
  const double *taps_padded;  // [ntaps + 2 * (CONV_R - 1)]: CONV_R - 1 zeros, the taps, CONV_R - 1 zeros
  int ntaps, centre;
  int reflect;       // 0 = nearest (edge replicate), 1 = scipy 'reflect'
  int dy_operation;
  float omin, omax;
  // fused preparation (PrepLaunch semantics); out_row[i] = prepared row of pattern i or -1
  // (navigation mask), nullptr = identity
  int do_prep;
  const int *out_row, *pix_map;
  int k, kpad, metric, operand_form;
  int f16_step = F16_STEP;  // float16 form: pixels per step of the layout (experimental tiles hold 256 patterns)
  float *out;
  float *scratch; size_t scratch_floats;  // streaming kernels, see preprocess_scratch_floats
};
// *prep_done tells whether the preparation was fused (else the caller runs launch_prep)
hipError_t launch_preprocess(const PreLaunch &a, bool *prep_done, hipStream_t s);
bool preprocess_fits_fused(int sy, int sx, int prepared_cols);
size_t preprocess_scratch_floats(int sy, int sx, int64_t n, int *grid_out);
size_t dtype_size(int dtype);

// ---- master-pattern projection (project.hip) --------------------------------
struct ProjectLaunch {
  const double *rotations;          // [n][4] unit quaternions
  int64_t n;
  const double *direction_cosines;  // [npix][3] (one PC for all patterns), unused with `pcs`
  int npix;
  const double *pcs;                // [n][3] one PC per pattern, or nullptr
  int nrows, ncols;                 // with `pcs`: detector shape (npix = nrows * ncols)
  double om[9];                     // with `pcs`: detector -> sample matrix
  const float *master_packed;       // pack_master_pattern() layout
  int npx, npy;
  int rescale;
  double out_min, out_max;
  int dtype_out;
  void *out;                        // [n][npix] of dtype_out
};
hipError_t launch_project(const ProjectLaunch &a, hipStream_t s);
// host: upper/lower [npy][npx] f32 -> [2][npy][npx + 1] float2 {m[r][c], m[r+1][c]}, edges repeated
size_t packed_master_floats(int npx, int npy);
void pack_master_pattern(const float *upper, const float *lower, int npx, int npy, float *out);

// ---- refinement (refine.hip) -------------------------------------------------
struct RefineLaunch {
  int mode, nvar, nfixed, n_starts;
  int64_t n_jobs;        // solve: patterns * starts; objective: evaluations
  const double *x0;      // [n_jobs][nvar]
  const double *fixed;   // [n_jobs][nfixed]
  const double *lower, *upper;  // [n_jobs][nvar] or nullptr
  int nrows, ncols, k;
  const unsigned *rowcol;       // [k] row << 16 | col of the kept pixels
  double om[9];
  const float *master_packed;
  int npx, npy;
  const float *patterns;        // [n][k] centred float32
  const double *sqnorm;         // [n]
  double xatol, fatol;
  int maxiter, maxfun;
  double *results;              // [n_jobs][REFINE_RESULT_STRIDE]: fun, nfev, nit, x[nvar]
};
constexpr int REFINE_RESULT_STRIDE = 9;
hipError_t launch_refine_prep(const void *raw, int dtype, int64_t n, int npix, const int *pix_map, int k, int rescale,
                              float *out, double *sqnorm, hipStream_t s);
hipError_t launch_refine_solve(const RefineLaunch &a, hipStream_t s);
hipError_t launch_refine_objective(const RefineLaunch &a, const int *pattern_index, double *out, hipStream_t s);
hipError_t launch_nelder_mead_selftest(int kind, int nvar, const double *x0, const double *lower, const double *upper,
                                       double xatol, double fatol, int maxiter, int maxfun, double *result,
                                       hipStream_t s);

// ---- orientation similarity map (osm.hip) -------------------------------------
// idx: [ny * nx][keep_n] int32 on the device; offsets: n_fp x (dy, dx) on the HOST
hipError_t launch_osm(const int *idx, int ny, int nx, int keep_n, int n_best, int from_n_best, const int *offsets,
                      int n_fp, int center_index, int normalize, float *out, hipStream_t s);

}  // namespace kpdi
