/* kpdi.h - C ABI of libkpdi.so, the MI355X (gfx950) dictionary-indexing engine.
 *
 * This is the drop-in boundary for ONE path of kikuchipy (reference =
 * /root/reference, paths below relative to its src/kikuchipy/):
 *
 *   EBSD.dictionary_indexing()            signals/ebsd.py:1827-1984
 *     -> _dictionary_indexing()           indexing/_dictionary_indexing.py:36-169
 *        -> SimilarityMetric.prepare_*()  indexing/similarity_metrics/_normalized_cross_correlation.py:88-159
 *                                         indexing/similarity_metrics/_normalized_dot_product.py:80-194
 *        -> SimilarityMetric.match()      ..._normalized_cross_correlation.py:161-183
 *        -> argtopk/topk + merge          indexing/_dictionary_indexing.py:172-203, :94-128
 *   EBSD.remove_static_background()       signals/ebsd.py:442-573   + pattern/_pattern.py:392-435, :484-509, :96-111
 *   EBSD.remove_dynamic_background()      signals/ebsd.py:575-696   + pattern/_pattern.py:438-481, :604-631,
 *                                         filters/fft_barnes.py:29-177
 *
 * The reference has no FFI for this path (it is pure Python + NumPy/Dask/Numba);
 * these entry points are what a ctypes binding inside a kikuchipy
 * `SimilarityMetric` subclass calls (INTEGRATION.md shows that binding).
 *
 * Conventions
 *  - plain C: pointers + sizes, no C++ or torch types; every function returns
 *    0 on success or a negative KPDI_E* code, and kpdi_last_error() returns
 *    the message for the calling thread's last failure.
 *  - one context (kpdi_ctx) = one GPU = one HIP stream; one OS thread drives a context.  Several GPUs from ONE
 *    thread of ONE process: a kpdi_group (below) - the kpdi_create(device_ids*, n_dev, &ctx) of SURVEY.md 8(b).
 *  - the caller owns all host buffers; the library owns all device buffers.
 *  - host inputs are never modified (tests/test_indexing/test_dictionary_indexing.py:41-43).
 *  - masks follow the reference: nonzero = EXCLUDED (similarity_metrics/_similarity_metric.py:51-58).
 *  - there is no CPU fallback: without a usable GPU kpdi_create() fails.
 *
 * Degenerate patterns
 *  A pattern whose normalisation is undefined is DEGENERATE:
 *    ncc - a CONSTANT pattern (a dead or saturated detector frame): all kept pixels EQUAL.  The test is exact (minimum
 *          == maximum of the pixels as read, after the cast to the compute dtype's float32) - there is no contrast floor:
 *          uint16 data around 60 000 counts with ONE pixel of 3600 off by one count is an ordinary pattern and correlates
 *          as in the reference (tests/test_gpu_degenerate.py::test_no_contrast_floor);
 *    ndp - all kept pixels zero;
 *    either metric - NaN or +-inf among the kept pixels, or a sum of squares that is not a positive finite float32
 *          (overflow; differences so small that their squares underflow).
 *  The reference divides 0 by 0 there (similarity_metrics/_normalized_cross_correlation.py:228-233,
 *  _normalized_dot_product.py:181-194): the prepared row is NaN, every score of it is NaN, and Dask's topk ranks NaN
 *  FIRST (dask/array/chunk.py:167-258) - a degenerate dictionary pattern becomes every experimental pattern's best
 *  match, a degenerate experimental pattern gets arbitrary indices with NaN scores.  SURVEY.md 8(a) puts that out
 *  of contract; THIS library's rule, on the experimental and on the dictionary side, in every arithmetic
 *  (KPDI_COMPUTE_*) and independent of chunking, tiling and sharding:
 *    a degenerate pattern is prepared as the ALL-ZERO row, so its score against every pattern is exactly +0.0
 *    ("no correlation") and ranks among the real scores like any other 0 - ties: lower dictionary index first.
 *  Hence a degenerate EXPERIMENTAL pattern comes back with scores 0 and the indices global_start .. of the lowest
 *  dictionary patterns pushed; a degenerate DICTIONARY pattern is selected only where fewer than keep_n real scores
 *  are positive; no other pattern's result changes (tests/test_gpu_degenerate.py, against oracle/kpdi_oracle.py which
 *  states the same rule).  Results never contain NaN.
 */
#ifndef KPDI_H
#define KPDI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kpdi_ctx kpdi_ctx;

/* error codes */
#define KPDI_OK 0
#define KPDI_EINVAL (-1)  /* bad argument / call order */
#define KPDI_EHIP (-2)    /* HIP runtime error */
#define KPDI_ENODEV (-3)  /* no usable GPU */
#define KPDI_ECOMM (-4)   /* RCCL error */
#define KPDI_ENOMEM (-5)
#define KPDI_ETIMEOUT (-6) /* a collective did not complete in time (kpdi_comm_selftest) */

/* similarity metric: "ncc" / "ndp" of EBSD._prepare_metric (signals/ebsd.py:3058-3061) */
#define KPDI_METRIC_NCC 0 /* zero-mean + L2 normalise, then dot product */
#define KPDI_METRIC_NDP 1 /* L2 normalise only, then dot product */

/* element type of a pattern array handed to the library */
#define KPDI_U8 0
#define KPDI_U16 1
#define KPDI_F32 2
#define KPDI_F64 3
#define KPDI_I8 4
#define KPDI_I16 5
#define KPDI_I32 6
#define KPDI_U32 7
#define KPDI_F16 8 /* IEEE half: dictionaries stored at half the bytes; not for background removal */

/* arithmetic of the match kernel */
#define KPDI_COMPUTE_F32 0 /* exact f32 MFMA (v_mfma_f32_32x32x2_f32), f32 accumulate: the default */
/* OPT-IN: every prepared value v is held as two f16, 2^12 v = hi + lo (22 significant bits),
 * and a product is hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 with f32 accumulation.
 * Scores differ from the f32 path by a few 1e-7 (the reference's own sgemm differs from exact
 * arithmetic by as much); inside the 1e-5 contract, not bit-identical to KPDI_COMPUTE_F32. */
#define KPDI_COMPUTE_F16X2 1
/* OPT-IN, REDUCED PRECISION (the "fp16 MFMA, fp32 accumulate" variant BASELINE.json configs[4]
 * names): every prepared value is ONE f16 (2^12 v rounded to 11 significant bits), one
 * v_mfma_f32_32x32x16_f16 per 16 pixels, f32 accumulation.  The rounding errors of the 2 K
 * operands of a score average out: measured against the f32 path at K = 3600, max 2e-5 / mean
 * 4e-6 - OUTSIDE the 1e-5 contract, and near-ties rank differently (1.3 % of the best-20 entries
 * of a random dictionary); a few 1e-4 for small K. */
#define KPDI_COMPUTE_F16 2
/* FLOAT64 ARITHMETIC (the reference's `dtype=float64`, _similarity_metric.py:244-253): scores are those of a
 * float64 evaluation (to ~1e-15; summation order differs from a dgemm's), returned by kpdi_finalize_f64.  The
 * KPDI_COMPUTE_F32 path screens keep_n + 12 candidates per pattern and chunk, csrc/rescore.hip rescores them in
 * double from the RAW patterns, keeps the best keep_n in double and checks that no unscreened candidate can
 * belong to them: an unscreened candidate's float32 score is at most the last screened one's, and its float64
 * score at most eps above its float32 score.  eps = the worst-case error of a K-term float32 dot product of unit
 * vectors, (K + 2) 2^-24 (2.1e-4 at K = 3600): a certificate that holds for ANY data - the default since round 5 (the
 * gap between the keep_n-th and the last screened score is an order of magnitude wider on ordinary data, so the proof
 * costs no extra pass: profiles/r05_f64_bounds.txt, also on an adversarial near-tie set).  KPDI_F64_EPS=statistical in
 * the environment selects round 4's bound instead: 8 x the largest |f32 - f64| difference seen among the rescored pairs
 * of the sweep, at least 1e-6 (about 1e5 samples per chunk, drawn from the best-scoring pairs) - fewer passes where
 * scores are closer than the worst case, no proof.  Where the check fails more screening passes run;
 * kpdi_counters.uncertified_patterns counts what is left (0 in every test and stress case).  Needs the
 * raw chunk: not available for resident (held) dictionaries. */
#define KPDI_COMPUTE_F64 3

/* background operations (`operation=` of remove_*_background) */
#define KPDI_OP_SUBTRACT 0
#define KPDI_OP_DIVIDE 1
/* `filter_domain=` of remove_dynamic_background (signals/ebsd.py:652-672) */
#define KPDI_DOMAIN_FREQUENCY 0 /* Barnes FFT filter == edge-replicating correlation */
#define KPDI_DOMAIN_SPATIAL 1   /* scipy.ndimage.gaussian_filter (reflect boundary) */

/* ---- library / device ---------------------------------------------------- */
const char *kpdi_version(void);
int kpdi_device_count(void);
/* message of the last failed call made by the CALLING THREAD (thread-local storage: contexts
 * driven from different threads never see each other's errors; valid until that thread's next call) */
const char *kpdi_last_error(void);

int kpdi_create(int device_id, kpdi_ctx **out);
int kpdi_destroy(kpdi_ctx *ctx);
int kpdi_synchronize(kpdi_ctx *ctx);

/* ---- problem set-up -------------------------------------------------------
 * Fixes what SimilarityMetric carries (similarity_metrics/_similarity_metric.py:69-86):
 * detector shape, signal mask (sy*sx bytes or NULL), metric and keep_n
 * (`keep_n = min(keep_n, dictionary_size)` is the caller's job,
 * indexing/_dictionary_indexing.py:67).  Resets the running top-k. */
int kpdi_set_problem(kpdi_ctx *ctx, int sy, int sx, const uint8_t *signal_mask,
                     int metric, int compute_dtype, int keep_n);

/* change keep_n only (forgets the running best-k, keeps everything else) */
int kpdi_set_keep_n(kpdi_ctx *ctx, int keep_n);

/* ---- experimental patterns (prepare_experimental, ..._cross_correlation.py:88-128)
 * `patterns`: m_all x sy x sx, C order, HOST memory.  `nav_mask`: m_all bytes or
 * NULL.  Uploads the raw patterns; they stay resident (and can be pre-processed
 * in place with kpdi_remove_*_background) until the first dictionary chunk is
 * pushed, when they are cast/masked/normalised once into the K-padded f32
 * matrix the match kernel reads. */
int kpdi_set_experimental(kpdi_ctx *ctx, const void *patterns, int dtype,
                          int64_t m_all, const uint8_t *nav_mask);
/* same, `patterns` already in DEVICE memory (copied device-to-device) */
int kpdi_set_experimental_dev(kpdi_ctx *ctx, const void *d_patterns, int dtype,
                              int64_t m_all, const uint8_t *nav_mask);
/* number of patterns that will be matched (nav mask applied) */
int64_t kpdi_n_experimental(kpdi_ctx *ctx);

/* ---- pre-processing of the resident experimental patterns -------------------
 * Both calls RECORD their step (arguments are validated and copied before they return); the
 * kernels run when the patterns are needed: fused with the metric's preparation of the
 * patterns into ONE kernel (static -> truncate -> dynamic -> truncate -> signal-mask gather ->
 * normalise -> tiled matrix; detectors up to 19 200 pixels, streaming kernels above that)
 * at the first dictionary chunk, or on their own in kpdi_get_experimental.  One static step
 * followed by one dynamic step fuse; any other sequence runs step by step.  Results are
 * identical either way: every step sees the truncated output of the previous one.
 * remove_static_background: pattern/_pattern.py:392-435 (+ :484-509, :96-111).
 * `static_bg`: sy*sx float32 (the caller did `static_bg.astype(float32)`,
 * signals/ebsd.py:547).  Output keeps the input dtype, truncated like `.astype`. */
int kpdi_remove_static_background(kpdi_ctx *ctx, const float *static_bg,
                                  int operation, int scale_bg);
/* remove_dynamic_background: pattern/_pattern.py:438-481; std <= 0 selects the
 * reference default sx/8 (signals/ebsd.py:648-649). */
int kpdi_remove_dynamic_background(kpdi_ctx *ctx, int operation, int filter_domain,
                                   double std, double truncate);
/* copy the (pre-processed) resident patterns back: m_all*sy*sx elements of the
 * dtype given to kpdi_set_experimental */
int kpdi_get_experimental(kpdi_ctx *ctx, void *patterns_out);

/* ---- dictionary sweep (_dictionary_indexing loop, indexing/_dictionary_indexing.py:94-128)
 * One call = one loop iteration: prepare_dictionary (cast, mask, normalise) +
 * match + top-k of the chunk + merge into the running best-k, all on the GPU.
 * `global_start` is the dictionary index of the chunk's first pattern
 * (`simulation_indices_i += start`, :118).  Chunks may arrive in any order and,
 * with several ranks, each rank pushes only its own shard.
 * The host-pointer form returns as soon as the upload has consumed `patterns` (the sweep
 * of the chunk is still running; it is ordered before every later call on the context);
 * uploads go through two staging buffers on a copy stream and overlap the sweep of the
 * previous piece / chunk - also with KPDI_COMPUTE_F64, whose look at a chunk's certification (and the extra screening
 * passes it may ask for) is left to the next call on the context.
 * SMALL CHUNKS ARE SWEPT TOGETHER.  The reference's call hands over `n_per_iteration` patterns per iteration - a tenth
 * of the dictionary in its tutorial, 3044 patterns - and one such chunk fills three quarters of ONE tile round of the
 * chip.  A pushed chunk of fewer than two tile rounds (8192 patterns against 4096 experimental ones) therefore only
 * joins the context's pending rows (its raw patterns are copied device-to-device); they are prepared and matched as
 * one launch set once three rounds' worth have come in, or when anything reads or ends the sweep (finalize,
 * synchronize, counters, export; a new experimental set, problem or keep_n forgets them with the running lists), and the
 * merge kernel translates rows of the coalesced matrix back to dictionary indices.  Chunks join in rising
 * dictionary order, one dtype, at most 16 per sweep; anything else sweeps the pending rows first.  The result never
 * depends on it (tests/test_gpu_engine.py); not in KPDI_COMPUTE_F64 (rescoring reads a chunk's own raw patterns);
 * KPDI_NO_COALESCE=1 switches it off.  Chunks handed over to be HELD (kpdi_hold_*) wait the same way and become one
 * resident chunk, eight rounds' worth at a time.  configs[1] pushed as 33 chunks of 3044: 26.6 -> 23.6 ms per call (one pass: 21.3; profiles/r05_group_chunks.txt). */
int kpdi_push_dictionary_chunk(kpdi_ctx *ctx, const void *patterns, int dtype,
                               int64_t n_chunk, int64_t global_start);
int kpdi_push_dictionary_chunk_dev(kpdi_ctx *ctx, const void *d_patterns, int dtype,
                                   int64_t n_chunk, int64_t global_start);
/* forget the running best-k (start another sweep over the same experimental set) */
int kpdi_reset_topk(kpdi_ctx *ctx);
/* scores: M x keep_n float32, descending; indices: M x keep_n int64 (the
 * reference's `simulation_indices` end up int64).  Ties: lower
 * dictionary index first.  With a communicator (kpdi_comm_init) every rank first
 * all-gathers the per-shard lists over RCCL and merges them, so all ranks get
 * the same, global result. */
int kpdi_finalize(kpdi_ctx *ctx, float *scores_out, int64_t *indices_out);
/* kpdi_finalize in two halves, for a caller that indexes map after map against one dictionary: kpdi_finalize_async queues
 * the (all-gather + merge over the ranks and the) result's device-to-host copies and returns a ticket at once;
 * kpdi_finalize_wait(ticket) blocks until they have landed and hands the result over.  Between the two the caller may
 * already queue the NEXT map (kpdi_set_experimental*, kpdi_push_*): the hand-over of a result - synchronisation, copies,
 * widening the indices, ~0.1 ms - then costs the GPU nothing.  At most two results may be pending; not available with
 * KPDI_COMPUTE_F64.  Same results as kpdi_finalize (= async + wait); the reference has no counterpart (its loop is
 * synchronous, indexing/_dictionary_indexing.py:94-128).  kpdi_pending_result_size: the number of (pattern, rank) entries
 * kpdi_finalize_wait will write for a ticket (m * keep_n at the time of the async call). */
int kpdi_finalize_async(kpdi_ctx *ctx, int *ticket);
int kpdi_finalize_wait(kpdi_ctx *ctx, int ticket, float *scores_out, int64_t *indices_out);
int kpdi_pending_result_size(kpdi_ctx *ctx, int ticket, int64_t *n);
/* The indices kpdi_finalize handed out last, as it left them in its page-locked staging buffer (int32, row-major
 * (m, keep_n)): valid until the next but one kpdi_finalize[_async] / kpdi_destroy of this context; *indices = NULL when there is none.
 * What a host layer compares a caller's array with before it tells kpdi_orientation_similarity_map to use the lists still
 * resident in HBM (indexing/_orientation_similarity_map.py:30-152 reads `simulation_indices` the caller may have edited). */
int kpdi_result_indices_i32(kpdi_ctx *ctx, const int32_t **indices, int64_t *n);
/* KPDI_COMPUTE_F64: the float64 scores (kpdi_finalize then returns them rounded to float32) */
int kpdi_finalize_f64(kpdi_ctx *ctx, double *scores_out, int64_t *indices_out);

/* ---- dictionary generation on the device ------------------
 * EBSDMasterPattern.get_patterns (signals/ebsd_master_pattern.py:95-330): project a
 * square-Lambert master pattern onto the detector, one simulated pattern per
 * rotation, without the dictionary ever existing in host memory.
 *
 * kpdi_set_master_pattern: the arrays `_get_master_pattern_arrays_from_energy`
 * returns (signals/_kikuchi_master_pattern.py:303-345): npy rows x npx columns per
 * hemisphere, dtype U8 / U16 / F32 (F64 is rounded to F32).  `lower` may be NULL
 * (hemisphere != "both": lower = upper). */
int kpdi_set_master_pattern(kpdi_ctx *ctx, const void *upper, const void *lower,
                            int dtype, int npx, int npy);
/* kpdi_set_detector: arguments of `_get_direction_cosines_for_fixed_pc`
 * (signals/util/_master_pattern.py:133-204) for the whole detector:
 * gnomonic_bounds = (x_min, x_max, y_min, y_max), om_detector_to_sample = 3x3
 * row-major.  The direction cosines (nrows*ncols x 3 f64) are computed on the host
 * in that function's operation order and kept on the device. */
int kpdi_set_detector(kpdi_ctx *ctx, const double *gnomonic_bounds, double pcz,
                      int nrows, int ncols, const double *om_detector_to_sample);
/* alternative: hand over direction cosines computed elsewhere (npix x 3 f64) */
int kpdi_set_direction_cosines(kpdi_ctx *ctx, const double *direction_cosines, int64_t npix);
/* the resident direction cosines, npix*3 doubles */
int kpdi_get_direction_cosines(kpdi_ctx *ctx, double *out);
/* `_project_patterns_from_master_pattern_with_fixed_pc` (:299-370): `rotations` =
 * n x 4 unit quaternions (a, b, c, d) f64; out = n x npix of `dtype_out` (F32, F64,
 * U8, U16; integers truncate like ndarray.astype) in HOST memory.  `rescale` != 0:
 * every pattern is min-max rescaled to [out_min, out_max] (pattern/_pattern.py:97-111). */
int kpdi_project_patterns(kpdi_ctx *ctx, const double *rotations, int64_t n, int rescale,
                          double out_min, double out_max, int dtype_out, void *out);
/* `_project_patterns_from_master_pattern_with_varying_pc` (:374-445) with
 * `_get_direction_cosines_for_varying_pc` (:216-295): pattern i is projected with its own
 * PC pcs[i] = (PCx, PCy, PCz), Bruker convention; the direction cosines are formed on the
 * device.  Needs only kpdi_set_master_pattern. */
int kpdi_project_patterns_varying_pc(kpdi_ctx *ctx, const double *rotations, const double *pcs,
                                     int64_t n, int nrows, int ncols,
                                     const double *om_detector_to_sample, int rescale,
                                     double out_min, double out_max, int dtype_out, void *out);
/* generate + match in one call - what the reference does when the dictionary is a lazy
 * `get_patterns(..., compute=False)` signal: the chunk is computed inside the indexing loop
 * (indexing/_dictionary_indexing.py:106-108).  The chunk of the dictionary belonging to
 * `rotations` is projected as float32 straight into device memory and swept like
 * kpdi_push_dictionary_chunk(..., KPDI_F32, n, global_start); the detector must have
 * sy*sx pixels (kpdi_set_problem). */
int kpdi_push_rotations_chunk(kpdi_ctx *ctx, const double *rotations, int64_t n,
                              int64_t global_start, int rescale, double out_min, double out_max);
/* the same with ONE PROJECTION CENTRE PER PATTERN - a lazy `get_patterns(rotations, detector)` whose detector holds a PC
 * for every rotation (signals/ebsd_master_pattern.py:236-241, :274-283: `nav_shape_det != (1,)`): pattern i of the chunk is
 * projected with pcs[i] = (PCx, PCy, PCz), Bruker convention, on the problem's sy x sx detector (the direction cosines are
 * formed on the device, as in kpdi_project_patterns_varying_pc), then the chunk is swept.  Needs kpdi_set_master_pattern;
 * no kpdi_set_detector. */
int kpdi_push_rotations_chunk_varying_pc(kpdi_ctx *ctx, const double *rotations, const double *pcs, int64_t n,
                                         int64_t global_start, const double *om_detector_to_sample, int rescale,
                                         double out_min, double out_max);

/* ---- resident dictionary: one dictionary, many maps -------------------------------
 * The reference prepares the dictionary again for every `dictionary_indexing()` call
 * (`metric.prepare_dictionary` inside the loop, indexing/_dictionary_indexing.py:106-110),
 * because host memory rarely holds it twice.  A prepared 60 x 60 dictionary of 100 000
 * patterns is 1.45 GB of the 288 GB of HBM, so a series of maps of the same phase and
 * detector (one `kpdi_set_problem`) can be indexed against chunks that are uploaded /
 * simulated and prepared ONCE: `hold` = prepare_dictionary into a buffer that stays,
 * `kpdi_sweep_held` = the match + top-k + merge of every held chunk against the current
 * experimental set (like pushing all of them again, results identical), then
 * kpdi_finalize.  kpdi_set_problem releases the held chunks (their layout depends on
 * shape, mask, metric and arithmetic); set_experimental / reset_topk / set_keep_n do not.
 * With several ranks each rank holds its own shard. */
int kpdi_hold_dictionary_chunk(kpdi_ctx *ctx, const void *patterns, int dtype, int64_t n_chunk,
                               int64_t global_start);
int kpdi_hold_dictionary_chunk_dev(kpdi_ctx *ctx, const void *d_patterns, int dtype,
                                   int64_t n_chunk, int64_t global_start);
/* simulated on the device like kpdi_push_rotations_chunk, then held */
int kpdi_hold_rotations_chunk(kpdi_ctx *ctx, const double *rotations, int64_t n,
                              int64_t global_start, int rescale, double out_min, double out_max);
int kpdi_sweep_held(kpdi_ctx *ctx);
int kpdi_release_held(kpdi_ctx *ctx);
/* patterns held and the device memory they occupy; either pointer may be NULL */
int kpdi_held_size(kpdi_ctx *ctx, int64_t *n_patterns, int64_t *n_bytes);

/* ---- refinement of orientations / projection centres ---------
 * EBSD.refine_orientation / refine_projection_center / refine_orientation_projection_center
 * with the default optimiser (scipy.optimize.minimize, Nelder-Mead): the chunk functions
 * `_refine_*_chunk_scipy` (indexing/_refinement/_refinement.py:441-503, :642-668, :779-809),
 * i.e. `_prepare_pattern` + the objective functions
 * (indexing/_refinement/_objective_functions.py:36-190) + the simplex search, all on the GPU:
 * one workgroup per (pattern, start) runs the whole optimisation.
 * The master pattern is the one given to kpdi_set_master_pattern (the caller passes the
 * float32 hemispheres of `_get_master_pattern_data`, _refinement.py:1288-1320). */
#define KPDI_REFINE_ORI 0     /* x = (phi1, Phi, phi2) [rad];            fixed = (PCx, PCy, PCz)   */
#define KPDI_REFINE_PC 1      /* x = (PCx, PCy, PCz) Bruker convention;  fixed = quaternion (a,b,c,d) */
#define KPDI_REFINE_ORI_PC 2  /* x = (phi1, Phi, phi2, PCx, PCy, PCz);   no fixed values            */
#define KPDI_REFINE_RESULT_STRIDE 9  /* result row: fun (= 1 - NCC), nfev, nit, x[0..nvar), padding */
/* Patterns to refine: `n` patterns of nrows x ncols `dtype` values in host memory.
 * signal_mask: nonzero = pixel NOT used, or NULL.  `rescale` != 0: intensities are rescaled
 * to [-1, 1] first (the reference does so exactly when the patterns are float32,
 * _refinement.py:956).  om_detector_to_sample: 3x3 row-major. */
int kpdi_refine_set_patterns(kpdi_ctx *ctx, const void *patterns, int dtype, int64_t n,
                             int nrows, int ncols, const uint8_t *signal_mask, int rescale,
                             const double *om_detector_to_sample);
/* the prepared patterns (n x k float32, centred) and their squared norms (n float64) */
int kpdi_refine_get_prepared(kpdi_ctx *ctx, float *patterns_out, double *sqnorm_out);
/* Objective values: out[e] = 1 - NCC(pattern[pattern_index[e]], projection(x[e], fixed[e])).
 * x: n_eval x nvar, fixed: n_eval x nfixed (nvar/nfixed = 3/3, 3/4, 6/0 for the three modes). */
int kpdi_refine_objective(kpdi_ctx *ctx, int mode, int64_t n_eval, const int32_t *pattern_index,
                          const double *x, const double *fixed, double *out);
/* Nelder-Mead from every start of every pattern.  x0 / fixed / lower / upper:
 * n_patterns x n_starts x (nvar | nfixed); lower/upper NULL = unbounded (no trust region).
 * maxiter / maxfev <= 0 select SciPy's defaults (200 * nvar each when both are unset, else
 * unlimited).  results: n_patterns x n_starts x KPDI_REFINE_RESULT_STRIDE doubles. */
int kpdi_refine_solve(kpdi_ctx *ctx, int mode, int64_t n_patterns, int n_starts,
                      const double *x0, const double *fixed, const double *lower,
                      const double *upper, double xatol, double fatol, int maxiter, int maxfev,
                      double *results);
/* The optimiser alone on an analytic f64 objective (kind 0: Rosenbrock, 1: weighted bowl),
 * nvar <= 6: lets a test compare the device's simplex path with SciPy's bit for bit.
 * result: fun, nfev, nit, x[0..nvar). */
int kpdi_nelder_mead_selftest(kpdi_ctx *ctx, int kind, int nvar, const double *x0,
                              const double *lower, const double *upper, double xatol,
                              double fatol, int maxiter, int maxfev, double *result);

/* ---- orientation similarity map ---------------------------------
 * kikuchipy.indexing.orientation_similarity_map
 * (indexing/_orientation_similarity_map.py:30-152).  simulation_indices: ny*nx x keep_n
 * int64 in host memory, or NULL = the best-k lists still resident from the last sweep
 * (kpdi_finalize; needs ny*nx == number of matched patterns).  footprint_offsets: n_fp
 * (dy, dx) pairs, the non-zero footprint elements in row-major order relative to the
 * footprint's centre (shape // 2); center_index as in the reference.  out: ny x nx x
 * (n_best - from_n_best + 1) float32, layer 0 = n_best; NaN where a point has no
 * neighbour. */
int kpdi_orientation_similarity_map(kpdi_ctx *ctx, const int64_t *simulation_indices, int ny,
                                    int nx, int keep_n, int n_best, int from_n_best,
                                    const int32_t *footprint_offsets, int n_fp, int center_index,
                                    int normalize, float *out);

/* ---- kikuchipy h5ebsd files --------------------------------------
 * What `kikuchipy.load("file.h5")` reads for this path
 * (io/plugins/kikuchipy_h5ebsd/_api.py:64-160, io/plugins/_h5ebsd.py:303-390): the scan's
 * header, `EBSD/Data/patterns` as (n_rows, n_columns, pattern_height, pattern_width) -
 * zero padded when the file holds fewer patterns -, `EBSD/Header/static_background` and
 * the projection centres.  The HDF5 C library is dlopen()ed (libhdf5.so from the
 * loader path, /opt/conda/lib, or $KPDI_HDF5_LIB); without it these calls fail.
 * `scan`: group name ("Scan 1"), NULL or "" = the first scan of the file. */
typedef struct kpdi_h5ebsd_info {
  char scan[64];                 /* the scan group that was read */
  int32_t ny, nx, sy, sx;        /* n_rows, n_columns, pattern_height, pattern_width */
  int32_t dtype;                 /* KPDI_* element type of the patterns */
  int32_t has_static_background; /* EBSD/Header/static_background of shape sy x sx present */
  int32_t static_background_dtype;
  int32_t binning;
  int64_t n_stored;              /* pattern values actually in the file (< ny*nx*sy*sx: padded) */
  int64_t n_pc;                  /* projection centres in the header (1 or ny*nx), 0 = none */
  double step_y, step_x, detector_pixel_size, sample_tilt, azimuth_angle, elevation_angle;
} kpdi_h5ebsd_info;
size_t kpdi_dtype_size(int dtype);
int kpdi_h5ebsd_info_read(const char *path, const char *scan, kpdi_h5ebsd_info *info);
int kpdi_h5ebsd_read_patterns(const char *path, const char *scan, void *out, size_t out_bytes);
int kpdi_h5ebsd_read_static_background(const char *path, const char *scan, void *out, size_t out_bytes);
/* out: n_pc x (PCx, PCy, PCz) float64 (Bruker convention, as stored) */
int kpdi_h5ebsd_read_pc(const char *path, const char *scan, double *out, int64_t n_pc);
/* read the scan's patterns through a pinned host buffer straight into the context's
 * experimental set (= kpdi_set_experimental of the file's contents) */
int kpdi_set_experimental_h5ebsd(kpdi_ctx *ctx, const char *path, const char *scan,
                                 const uint8_t *nav_mask);

/* ---- multi-GPU: dictionary sharded over ranks, one process per GPU -------- */
#define KPDI_UNIQUE_ID_BYTES 128
int kpdi_comm_unique_id(uint8_t *id_out /* KPDI_UNIQUE_ID_BYTES */);
int kpdi_comm_init(kpdi_ctx *ctx, int rank, int nranks, const uint8_t *id);
/* The fallback chain of a multi-process job (kikuchipy_amd/parallel.py: Communicator.attach; the sharding replaces the
 * chunk loop of indexing/_dictionary_indexing.py:100-128, whose merge :120-128 is what the gather feeds):
 *   kpdi_comm_selftest - ONE all-gather of n_bytes per rank, awaited for at most timeout_ms (KPDI_ETIMEOUT): a
 *       communicator that bootstrapped can still hang in its first collective;
 *   kpdi_comm_drop - forget (abort) the communicator, also while a kpdi_comm_init of this context hangs on another
 *       thread: finalize then returns this context's own lists, or what kpdi_import_lists gave it;
 *   kpdi_export_lists / kpdi_import_lists - the HOST-STAGED gather: every rank exports its own m x keep_n lists
 *       (scores float, or double in KPDI_COMPUTE_F64; indices int32), the caller's transport all-gathers them
 *       (M * keep_n * 8 bytes per rank: 0.66 MB at configs[1]), every rank imports the n_ranks lists (rank-major) and
 *       its next kpdi_finalize[_f64 / _async] merges them with the kernel - and the total order - of the RCCL path. */
int kpdi_comm_selftest(kpdi_ctx *ctx, int64_t n_bytes, int timeout_ms);
int kpdi_comm_drop(kpdi_ctx *ctx);
int kpdi_export_lists(kpdi_ctx *ctx, void *scores_out, int32_t *indices_out);
int kpdi_import_lists(kpdi_ctx *ctx, const void *scores_all, const int32_t *indices_all, int n_ranks);

/* ---- multi-GPU from ONE process: a group of contexts ----------------------------------------------------
 * The reference's call is one call in one interpreter (signals/ebsd.py:1827-1984; its loop over dictionary
 * chunks, indexing/_dictionary_indexing.py:100-128, never crosses a process boundary).  A kpdi_group keeps that
 * shape on a node with several MI355X: it owns one context per entry of `device_ids` and one host thread per
 * member, and its entry points are the per-context ones fanned out -
 *   set_problem / set_keep_n / set_experimental / remove_*_background / set_master_pattern / set_detector: every member
 *       (the experimental set is replicated from the caller's ONE host buffer);
 *   push_* / hold_*: a dictionary chunk is handed to the member(s) the assignment rule names (csrc/group_assign.h,
 *       kpdi_group_assign_chunk).  With the dictionary size announced (kpdi_group_set_dictionary_size) member i has
 *       a quota = the i-th of n_dev near-equal parts of the dictionary; a chunk goes to the member that has taken the
 *       fewest patterns so far, never beyond its quota - the rest spills to the next.  A single-pass call (one chunk =
 *       the dictionary) is thereby cut into the contiguous n_dev parts, while the chunks of a CHUNKED call - the
 *       reference's own shape: n_per_iteration patterns per iteration, _dictionary_indexing.py:100-128 - stay whole
 *       and go round the members (a tenth of the dictionary cut 8 ways would leave a member a fraction of one tile
 *       round per launch).  Size unknown: a chunk is cut into min(n_dev, n_chunk / (2 tile rounds)) >= 1 near-equal
 *       pieces for the least-loaded members.  A piece is pushed with global_start + its first row, so a global
 *       index is still chunk start + row (`simulation_indices_i += start`, :118).  Pushes only QUEUE the pieces on the
 *       members' host threads and return (at most 2 chunks wait per member): the next chunk can be fetched while
 *       the previous ones are uploaded / generated / swept.  A failure of queued work is reported by the next
 *       joining call (synchronize, finalize, any set_* call);
 *   finalize: the members' best-k lists are gathered and merged by the same (score desc, index asc) kernel used
 *       between chunks, and the caller gets ONE result - identical, bit for bit, to the single-context result.
 * Gather = KPDI_GATHER_RCCL: an in-process RCCL communicator (ncclCommInitAll - no sockets, no environment) and the
 * ncclAllGather of kpdi_finalize, or KPDI_GATHER_P2P: hipMemcpyPeerAsync of the members' M * keep_n * 8 bytes into
 * member 0 (xGMI between devices) - which also works when several members share ONE device (RCCL refuses duplicate
 * devices), so the whole multi-device code path runs on a 1-GPU box.  KPDI_GATHER_AUTO: $KPDI_GATHER = "rccl" | "p2p"
 * if set, else P2P when a device appears twice, else RCCL (falling back to P2P, with the reason kept for
 * kpdi_group_describe, if the communicator cannot be created or its first all-gather - a small one, run by
 * kpdi_group_create on every member at once under $KPDI_COMM_TIMEOUT seconds, default 60 - does not complete; with
 * KPDI_GATHER_RCCL asked for by name that is an error instead).  A group of one device gathers nothing.
 * Calls other than the chunk pushes are synchronous with respect to the members' host work (they return when every
 * member's call has) and, like the per-context calls, asynchronous with respect to the GPUs.  One thread at a time
 * drives a group.
 * The members stay reachable (kpdi_group_member) for the per-context calls that have no group form - device buffers,
 * counters, refinement; the caller must not use a member while a group call is running. */
typedef struct kpdi_group kpdi_group;
#define KPDI_GATHER_AUTO 0
#define KPDI_GATHER_RCCL 1
#define KPDI_GATHER_P2P 2
#define KPDI_GATHER_NONE 3 /* reported by kpdi_group_gather for a group of one device */
int kpdi_group_create(const int *device_ids, int n_dev, int gather, kpdi_group **out);
int kpdi_group_destroy(kpdi_group *g);
int kpdi_group_size(const kpdi_group *g);
int kpdi_group_gather(const kpdi_group *g);        /* the KPDI_GATHER_* mode in use */
const char *kpdi_group_describe(const kpdi_group *g); /* "8 devices [0,1,...], gather rccl" (+ why a fallback was taken) */
kpdi_ctx *kpdi_group_member(kpdi_group *g, int i); /* borrowed; NULL when i is out of range */
/* rows [*start, *end) of the i-th of n_dev near-equal contiguous parts of n rows (a member's quota of the dictionary) */
int kpdi_group_chunk_share(int64_t n_chunk, int i, int n_dev, int64_t *start, int64_t *end);
/* The assignment rule as a pure function (planning, tests; needs no GPU): the pieces of a chunk of n_chunk patterns
 * for a group of n_dev members that have taken load[i] patterns so far (updated), dictionary size n_total (0 =
 * unknown, then min_piece decides between the n_dev-way cut and a whole chunk).  Up to max_pieces pieces are written
 * (member, first row, rows - in row order), *n_pieces = how many there are. */
int kpdi_group_assign_chunk(int n_dev, int64_t n_total, int64_t min_piece, int64_t *load, int64_t n_chunk, int max_pieces,
                            int *member_out, int64_t *row0_out, int64_t *rows_out, int *n_pieces);
/* Dictionary patterns the coming sweep(s) will push in total (0 = unknown); starts a new assignment.  The members'
 * takes are also reset by set_problem / set_keep_n / set_experimental* / reset_topk (a new sweep). */
int kpdi_group_set_dictionary_size(kpdi_group *g, int64_t n_total);
int kpdi_group_synchronize(kpdi_group *g);
int kpdi_group_set_problem(kpdi_group *g, int sy, int sx, const uint8_t *signal_mask, int metric, int compute_dtype,
                           int keep_n);
int kpdi_group_set_keep_n(kpdi_group *g, int keep_n);
int kpdi_group_set_experimental(kpdi_group *g, const void *patterns, int dtype, int64_t m_all, const uint8_t *nav_mask);
/* d_patterns[i]: the set in the memory of member i's device (callers that keep their inputs resident) */
int kpdi_group_set_experimental_dev(kpdi_group *g, const void *const *d_patterns, int dtype, int64_t m_all,
                                    const uint8_t *nav_mask);
int64_t kpdi_group_n_experimental(kpdi_group *g);
int kpdi_group_remove_static_background(kpdi_group *g, const float *static_bg, int operation, int scale_bg);
int kpdi_group_remove_dynamic_background(kpdi_group *g, int operation, int filter_domain, double std, double truncate);
int kpdi_group_get_experimental(kpdi_group *g, void *patterns_out); /* member 0's (all members hold the same) */
/* returns when the members' uploads have consumed `patterns` (reference loop: the chunk is a temporary, :106-108) */
int kpdi_group_push_dictionary_chunk(kpdi_group *g, const void *patterns, int dtype, int64_t n_chunk,
                                     int64_t global_start);
/* the same, returning at once: `patterns` is BORROWED until kpdi_group_chunks_consumed reports a ticket >= *ticket
 * (tickets count up from 1), or until kpdi_group_synchronize / a finalize has returned */
int kpdi_group_push_dictionary_chunk_borrowed(kpdi_group *g, const void *patterns, int dtype, int64_t n_chunk,
                                              int64_t global_start, int64_t *ticket);
int kpdi_group_chunks_consumed(kpdi_group *g, int64_t *ticket); /* every borrowed chunk up to *ticket has been consumed */
/* explicit per-member chunks in device memory: member i sweeps n_chunk[i] patterns at d_patterns[i] (n_chunk[i] = 0:
 * nothing) whose first pattern has dictionary index global_start[i] */
int kpdi_group_push_dictionary_chunk_dev(kpdi_group *g, const void *const *d_patterns, int dtype, const int64_t *n_chunk,
                                         const int64_t *global_start);
int kpdi_group_set_master_pattern(kpdi_group *g, const void *upper, const void *lower, int dtype, int npx, int npy);
int kpdi_group_set_detector(kpdi_group *g, const double *gnomonic_bounds, double pcz, int nrows, int ncols,
                            const double *om_detector_to_sample);
int kpdi_group_push_rotations_chunk(kpdi_group *g, const double *rotations, int64_t n, int64_t global_start, int rescale,
                                    double out_min, double out_max);
int kpdi_group_hold_dictionary_chunk(kpdi_group *g, const void *patterns, int dtype, int64_t n_chunk,
                                     int64_t global_start);
int kpdi_group_hold_rotations_chunk(kpdi_group *g, const double *rotations, int64_t n, int64_t global_start, int rescale,
                                    double out_min, double out_max);
int kpdi_group_sweep_held(kpdi_group *g);
int kpdi_group_release_held(kpdi_group *g);
int kpdi_group_held_size(kpdi_group *g, int64_t *n_patterns, int64_t *n_bytes); /* summed over the members */
int kpdi_group_reset_topk(kpdi_group *g);
/* as kpdi_finalize / _f64 / _async / _wait / kpdi_pending_result_size, the result being the merge over all members */
int kpdi_group_finalize(kpdi_group *g, float *scores_out, int64_t *indices_out);
int kpdi_group_finalize_f64(kpdi_group *g, double *scores_out, int64_t *indices_out);
int kpdi_group_finalize_async(kpdi_group *g, int *ticket);
int kpdi_group_finalize_wait(kpdi_group *g, int ticket, float *scores_out, int64_t *indices_out);
int kpdi_group_pending_result_size(kpdi_group *g, int ticket, int64_t *n);
int kpdi_group_set_profiling(kpdi_group *g, int on);
int kpdi_group_reset_counters(kpdi_group *g);

/* ---- the launch planner, as data (csrc/plan.h; needs no GPU) -----------------
 * How a sweep of n_chunk dictionary patterns against m experimental patterns (k_kept pixels, keep_n) is laid on a
 * device of n_cu compute units: which f32 kernel (`form` -1 = the automatic choice, 0 = match.hip's 128-pattern
 * tiles, 3 = match16.hip's 256-pattern tiles), how many workgroups share a row block's dictionary tiles (nsplit), how
 * many row blocks a launch covers, each launch's XCD grid (and its padding), the tail plan.  The environment's
 * developer switches are read as kpdi_set_problem reads them. */
#define KPDI_PLAN_MAX_LAUNCHES 64
typedef struct kpdi_plan {
  int32_t form, tile;          /* kernel form (0 / 3) and its dictionary patterns per tile */
  int32_t row_blocks;          /* 256-pattern blocks of the experimental set */
  int32_t n_tiles, nsplit, rows_per_launch, launches;
  int64_t round_rows;          /* dictionary patterns one full round of the chip covers */
  int32_t n_main;              /* tiles of the main launch(es) */
  int32_t tail_tiles, tail_units, tail_nsplit, fixed_draws; /* match.hip: quarter-tile tail launch, fixed hand-out */
  int32_t tail_first, tail_shift;                           /* match16.hip f32: partial units from tile tail_first on */
  int32_t perm_rounds, perm_stride; /* match16.hip: round j < perm_rounds of split sp takes tile sp + nsplit * ((j * perm_stride)
                                       mod perm_rounds) - a low-discrepancy walk, so that the order of the dictionary
                                       (sampler order, sorted by score) does not matter to the fused top-k; 0 = natural order */
  int32_t tail_gemm_rows;           /* match16.hip f32: dictionary rows behind the whole rounds that tailgemm.hip takes as a kernel of
                                       its own (32 x 128 workgroups, scores -> select -> a merge source); then tail_shift = 0 */
  int32_t n_launch_desc;
  struct {
    int32_t row_first, rows, xcd_rows, xcd_splits, rows_grid;
  } launch[KPDI_PLAN_MAX_LAUNCHES];
} kpdi_plan;
int kpdi_plan_describe(int64_t m, int64_t n_chunk, int k_kept, int keep_n, int n_cu, int form, kpdi_plan *out);

/* ---- device buffers for callers that keep data resident (bench.py) -------- */
int kpdi_dev_alloc(kpdi_ctx *ctx, size_t bytes, void **d_out);
int kpdi_dev_free(kpdi_ctx *ctx, void *d_ptr);
int kpdi_h2d(kpdi_ctx *ctx, void *d_dst, const void *src, size_t bytes);
int kpdi_d2h(kpdi_ctx *ctx, void *dst, const void *d_src, size_t bytes);

/* ---- measurement ------------------------------------------------------------
 * kpdi_set_profiling(ctx, level): 0 off; 1 every phase of a sweep (preparation, match, merge, all-gather, ...) is
 * bracketed by HIP events on the context's stream; 2 only the match launches (and the all-gather) are - an event
 * record between two kernels leaves the GPU idle for ~6 us, 0.05 ms per step with level 1, which a 3 ms step
 * notices; 3 = level 1 + the fused top-k of match16.hip counts what its epilogues do (kpdi_counters.epi_*: a few
 * thousand atomics per launch, ~50 us - a developer level).  kpdi_get_counters() synchronises and sums the events. */
typedef struct kpdi_counters {
  double match_ms;      /* sum of match-kernel durations (HIP events) */
  int64_t match_launches;
  double match_flops;   /* 2 * M * n_chunk * K summed over launches (K = kept pixels) */
  double prep_ms;       /* dictionary + experimental normalisation kernels */
  double merge_ms;      /* top-k merge kernels */
  double h2d_bytes;     /* bytes copied host->device by push/set calls */
  int32_t match_grid;   /* workgroups of the last match launch */
  int32_t match_nsplit; /* dictionary splits of the last match launch */
  int32_t kpad;         /* padded reduction length */
  int32_t k_kept;       /* kept pixels K */
  double project_ms;    /* master-pattern projection kernels */
  double refine_ms;     /* refinement solve kernels */
  double preproc_ms;    /* background-removal kernels (incl. the fused preparation of the patterns) */
  int64_t preproc_launches;
  double rescore_ms;             /* KPDI_COMPUTE_F64: float64 rescoring + merge kernels */
  int64_t rescore_extra_passes;  /* ... screening passes beyond the first keep_n + 12 candidates of a chunk */
  int64_t uncertified_patterns;  /* ... (pattern, chunk) pairs whose best-k could not be certified; 0 in practice */
  int32_t match_form;            /* operand form of the last match launch: 0 match.hip f32, 1 split f16, 2 float16 (match16.hip),
                                    3 f32 on match16.hip's one-wave-per-SIMD kernel (chosen per sweep, KPDI_F32_WIDE forces) */
  int32_t comm_ranks;            /* ranks of the RCCL communicator (ncclCommCount), 0 = none attached */
  double comm_ms;                /* RCCL all-gather of the per-rank best-k lists inside kpdi_finalize (incl. waiting for the
                                    slowest rank to arrive) */
  double fixed_ms;               /* per-sweep bookkeeping kernels around the match: list / bound / counter initialisation */
  int32_t f64_certificate;       /* KPDI_COMPUTE_F64: 2 = worst-case bound (default), 1 = statistical bound (KPDI_F64_EPS=statistical
                                    at kpdi_set_problem); 0 = not float64 arithmetic.  `uncertified_patterns == 0` is a proof only for 2 */
  int32_t gather_ranks;          /* lists merged by the last finalize: RCCL ranks or peer-copied group members, 0 = this context's own only */
  int64_t coalesced_sweeps;      /* sweeps that took several small pushed chunks together (kpdi_push_dictionary_chunk: coalescing) */
  /* what the fused top-k of match16.hip did (collected at profiling level 1 only; per-lane lists: one per lane and 32
   * experimental patterns of a wave tile): lists that ran, candidates that passed the shared bound and were appended to
   * a lane's buffer, buffers that overflowed (the list is then built from the tile directly: the slow path), first
   * tiles that took that path because the bound was not complete in time */
  int64_t epi_lists;
  int64_t epi_appended;
  int64_t epi_overflows;
  int64_t epi_direct_first;
} kpdi_counters;
/* sizeof(kpdi_counters) as the LIBRARY was built: a binding whose struct differs must refuse to call kpdi_get_counters
 * (the struct has grown between versions; kpdi_version() changes with it) */
size_t kpdi_counters_size(void);
int kpdi_set_profiling(kpdi_ctx *ctx, int on);
int kpdi_get_counters(kpdi_ctx *ctx, kpdi_counters *out);
int kpdi_reset_counters(kpdi_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* KPDI_H */
