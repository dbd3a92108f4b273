"""CPU restatement (NumPy) of kikuchipy's dictionary-indexing hot path.

TEST INFRASTRUCTURE, NOT PRODUCT.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this module; `kikuchipy_amd/` never
does (tests/test_no_oracle_in_product.py enforces it).

Parity status: PINNED.  `oracle/gen_golden.py` ran the reference's own modules
(loaded unmodified from /root/reference by `oracle/ref_shim.py`) in the build
container and stored their inputs/outputs in `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every function here against those
vectors and against the hard-coded known answers of the reference's own test
suite (tests/test_signals/test_ebsd.py, tests/test_pattern/test_pattern.py,
tests/test_filters/test_fft_barnes.py, tests/test_indexing/*).

Every function cites the reference file:line it follows (paths relative to
/root/reference/src/kikuchipy unless they start with tests/).

Third-party arithmetic on the path that is NOT under /root/reference and is
restated from its published behaviour:
* dask.array.topk/argtopk (dask 2021.10.0, dask/array/chunk.py:167-258):
  k largest along the axis, returned in descending order; order among equal
  values unspecified.  Restated in `topk_desc` with the engine's documented
  tie rule (lower dictionary index first).
* numpy.einsum("ik,mk->im") -> BLAS sgemm: restated as `exp @ dic.T`.
* scipy.signal.windows.gaussian, scipy.fft.rfft2/irfft2/next_fast_len,
  scipy.ndimage.gaussian_filter: called directly (SciPy is importable both in
  the build container and on the GPU box).
* skimage.util.dtype.dtype_range: the table `DTYPE_RANGE` below.
"""

import numpy as np

# skimage/util/dtype.py `dtype_range` (used at signals/ebsd.py:523, :676)
DTYPE_RANGE = {
    np.dtype(np.uint8): (0, 255),
    np.dtype(np.uint16): (0, 65535),
    np.dtype(np.uint32): (0, 2**32 - 1),
    np.dtype(np.int8): (-128, 127),
    np.dtype(np.int16): (-32768, 32767),
    np.dtype(np.int32): (-(2**31), 2**31 - 1),
    np.dtype(np.float16): (-1, 1),
    np.dtype(np.float32): (-1, 1),
    np.dtype(np.float64): (-1, 1),
}

ALLOWED_DTYPES = (np.dtype(np.float32), np.dtype(np.float64))


# --------------------------------------------------------------------------
# a4/a5/a6: pattern preparation
# --------------------------------------------------------------------------
# Degenerate patterns.  Where a pattern's normalisation is undefined - zero variance for `ncc` (a constant pattern: a
# dead or saturated detector frame), all zeros for `ndp`, NaN / inf among the kept pixels for either - the reference
# divides 0 by 0 (`_normalized_cross_correlation.py:228-233`, `_normalized_dot_product.py:181-194`): the row is NaN,
# every score of it is NaN, and Dask's `topk` ranks NaN FIRST (`dask/array/chunk.py:167-258`: NaN sorts as the largest
# value) - a degenerate DICTIONARY pattern becomes everybody's best match, a degenerate experimental pattern gets
# arbitrary indices with NaN scores.  SURVEY.md 8(a) puts that out of contract and asks the engine to document what it
# does instead; the ENGINE'S RULE (include/kpdi.h "Degenerate patterns", csrc/prep_device.h: degenerate_pattern), which
# the oracle applies with `degenerate="zero"` (the default: it is the engine's checker): the row becomes ALL ZEROS, so
# its score against every pattern is exactly 0 - "no correlation" - on either side, and ranks among the real scores
# like any other 0 (ties: lower dictionary index first).  `degenerate="reference"` leaves the reference's NaN.
def degenerate_rows(norm2, constant=None):
    """Rows whose normalisation is undefined (see above): `norm2` (n,) the sum of squares a row is divided by the root
    of - not a positive finite number (all zeros, NaN / inf in the data); `constant` (n,) bool, `ncc` only: all kept
    pixels of the row are EQUAL (minimum == maximum, an exact test: no contrast floor - one pixel off by one count on a
    60 000-count background is an ordinary pattern, `_normalized_cross_correlation.py:228-233` correlates it)."""
    with np.errstate(invalid="ignore", over="ignore"):
        norm2 = np.asarray(norm2, dtype=np.float64)
        bad = ~((norm2 > 0) & (norm2 < np.inf))
    return bad if constant is None else bad | np.asarray(constant, dtype=bool)


def zero_mean_normalize(patterns, degenerate="zero"):
    """indexing/similarity_metrics/_normalized_cross_correlation.py:228-233
    (`_zero_mean_normalize_patterns_numpy`); in place on a private copy."""
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        constant = np.min(patterns, axis=1) == np.max(patterns, axis=1)
        patterns_mean = np.mean(patterns, axis=1, keepdims=True)
        patterns -= patterns_mean
        norm2 = np.sum(np.square(patterns), axis=1, keepdims=True)
        patterns_norm = np.sqrt(norm2)
        patterns /= patterns_norm
    if degenerate == "zero":
        patterns[degenerate_rows(norm2[:, 0], constant)] = 0
    return patterns


def normalize(patterns, degenerate="zero"):
    """indexing/similarity_metrics/_normalized_dot_product.py:181-194
    (`_normalize_patterns`): L2 only, no mean subtraction."""
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        norm2 = np.sum(np.square(patterns), axis=1)
        out = patterns / np.sqrt(norm2)[..., np.newaxis]
    if degenerate == "zero":
        out[degenerate_rows(norm2)] = 0
    return out


def reference_topk_with_nan(similarities, k):
    """What the reference's `argtopk` / `topk` return when scores are NaN (dask/array/chunk.py:167-258 + NumPy's sort
    order: NaN is the largest value): the NaN entries FIRST.  Documentation of the difference to the engine's rule,
    used by tests/test_gpu_degenerate.py; (indices, scores) of the k 'largest' per row."""
    order = np.argsort(similarities, axis=1, kind="stable")[:, ::-1][:, :k]  # ascending with NaN last, reversed
    return order, np.take_along_axis(similarities, order, axis=1)


def check_dtype(dtype):
    """indexing/similarity_metrics/_similarity_metric.py:244-253."""
    dtype = np.dtype(dtype)
    if dtype not in ALLOWED_DTYPES:
        raise ValueError(
            f"Data type {dtype} not among supported data types "
            f"{[np.float32, np.float64]}"
        )
    return dtype


def prepare_experimental(
    patterns, metric="ncc", n_experimental=None, navigation_mask=None,
    signal_mask=None, dtype=np.float32,
):
    """_normalized_cross_correlation.py:88-128 / _normalized_dot_product.py:80-120.

    cast -> reshape (M_all, -1) -> drop rows where nav mask is True -> drop
    columns where signal mask is True -> normalise.  (Rechunking is a Dask
    scheduling detail with no numerical effect.)"""
    dtype = check_dtype(dtype)
    patterns = np.asarray(patterns).astype(dtype)
    if n_experimental is None:
        n_experimental = max(int(np.prod(patterns.shape[:-2])), 1)
    patterns = patterns.reshape((n_experimental, -1))
    if navigation_mask is not None:
        patterns = patterns[~np.asarray(navigation_mask).ravel()]
    if signal_mask is not None:
        patterns = patterns[:, ~np.asarray(signal_mask).ravel()]
    if metric == "ncc":
        return zero_mean_normalize(patterns)
    return normalize(patterns)


def prepare_dictionary(patterns, metric="ncc", signal_mask=None, dtype=np.float32):
    """_normalized_cross_correlation.py:130-159 / _normalized_dot_product.py:122-150.
    `astype` copies, so the caller's dictionary is never mutated
    (tests/test_indexing/test_dictionary_indexing.py:41-43)."""
    dtype = check_dtype(dtype)
    patterns = np.asarray(patterns)
    patterns = patterns.reshape((patterns.shape[0], -1)).astype(dtype)
    if signal_mask is not None:
        patterns = patterns[:, ~np.asarray(signal_mask).ravel()]
    if metric == "ncc":
        return zero_mean_normalize(patterns)
    return normalize(patterns)


# --------------------------------------------------------------------------
# a7/a8: match + top-k
# --------------------------------------------------------------------------
def match(experimental, dictionary):
    """_normalized_cross_correlation.py:161-183: einsum("ik,mk->im") == X @ Y.T."""
    return experimental @ dictionary.T


def topk_desc(similarities, k):
    """indexing/_dictionary_indexing.py:197-198 (`argtopk`/`topk`, axis=-1).

    Returns (indices int64, scores) of the k largest per row, descending.  Ties:
    lower dictionary index first (the engine's documented rule; the reference's
    own order among equal scores is unspecified)."""
    n = similarities.shape[1]
    k = min(k, n)
    # stable argsort of -score keeps lower index first among equal scores
    order = np.argsort(-similarities, axis=1, kind="stable")[:, :k]
    scores = np.take_along_axis(similarities, order, axis=1)
    return order.astype(np.int64), scores


def merge_topk(scores, indices, scores_i, indices_i, keep_n):
    """indexing/_dictionary_indexing.py:120-128: hstack running best with the
    chunk's best, argsort descending, keep the first `keep_n`.  Tie rule as in
    `topk_desc` (sort key = (-score, index))."""
    all_scores = np.hstack((scores, scores_i))
    all_idx = np.hstack((indices, indices_i))
    order = np.lexsort((all_idx, -all_scores), axis=1)[:, :keep_n]
    return (
        np.take_along_axis(all_scores, order, axis=1),
        np.take_along_axis(all_idx, order, axis=1),
    )


# --------------------------------------------------------------------------
# a9: the chunk loop
# --------------------------------------------------------------------------
def dictionary_indexing(
    experimental, dictionary, metric="ncc", keep_n=20, n_per_iteration=None,
    navigation_mask=None, signal_mask=None, dtype=np.float32,
):
    """indexing/_dictionary_indexing.py:36-128 on plain arrays.

    experimental: (..., sy, sx); dictionary: (N, sy, sx).  Returns
    (scores (M, k) dtype, simulation_indices (M, k) int64) for the M patterns
    where the navigation mask is False."""
    dictionary = np.asarray(dictionary)
    dictionary_size = dictionary.shape[0]
    sig = experimental.shape[-2:]
    n_exp_all = max(int(np.prod(experimental.shape[:-2])), 1)
    if n_per_iteration is None:
        n_per_iteration = dictionary_size  # signals/ebsd.py:1925-1929
    keep_n = min(keep_n, dictionary_size)  # :67
    n_iterations = int(np.ceil(dictionary_size / n_per_iteration))  # :68
    exp = prepare_experimental(
        experimental, metric, n_exp_all, navigation_mask, signal_mask, dtype
    )  # :70
    dictionary = dictionary.reshape((dictionary_size, -1))  # :71
    if sig[0] * sig[1] != dictionary.shape[1]:
        raise ValueError("signal shapes differ")
    n_exp = exp.shape[0]

    if dictionary_size == n_per_iteration:  # :88-93
        sim = match(exp, prepare_dictionary(dictionary, metric, signal_mask, dtype))
        idx, scores = topk_desc(sim, keep_n)
        return scores, idx

    sign = 1  # both metrics: greater is better
    indices = np.zeros((n_exp, keep_n), dtype=np.int64)  # :97 (int32 there; hstack -> int64)
    scores = np.full((n_exp, keep_n), -sign, dtype=dtype)  # :98  -1.0 sentinel
    starts = np.cumsum([0] + [n_per_iteration] * (n_iterations - 1))  # :102
    ends = np.cumsum([n_per_iteration] * n_iterations)  # :103
    ends[-1] = max(ends[-1], dictionary_size)  # :104
    for start, end in zip(starts, ends):
        chunk = dictionary[start:end]
        end = min(end, dictionary_size)
        sim = match(exp, prepare_dictionary(chunk, metric, signal_mask, dtype))
        idx_i, scores_i = topk_desc(sim, min(keep_n, end - start))  # :110-115
        idx_i = idx_i + start  # :118
        scores, indices = merge_topk(scores, indices, scores_i, idx_i, keep_n)
    return scores, indices


def plugin_prepare_metric(metric, navigation_size, navigation_mask, signal_mask, dtype, n_dictionary_patterns):
    """What `EBSD._prepare_metric` does to a metric OBJECT it was handed (signals/ebsd.py:3072-3086; the
    `isinstance` gate of :3065-3070 needs the reference's own ABC and is exercised by oracle/seam_check.py)."""
    metric.n_experimental_patterns = max(navigation_size, 1)
    metric.n_dictionary_patterns = max(n_dictionary_patterns, 1)
    if navigation_mask is not None:
        metric.navigation_mask = navigation_mask
    if signal_mask is not None:
        metric.signal_mask = signal_mask
    if dtype is not None:
        metric.dtype = dtype
    metric.raise_error_if_invalid()
    return metric


def plugin_loop(metric, experimental, experimental_nav_shape, dictionary, keep_n, n_per_iteration, phase_name="ni"):
    """The reference's loop around a metric PLUGIN - `_dictionary_indexing` (indexing/_dictionary_indexing.py:36-169)
    and `_match_chunk` (:172-203) restated call for call: which methods of the metric object are called, in which
    order, with what; the lazy-dictionary `.compute()` branch (:106-108), the host merge (:118-128), the scatter under
    a navigation mask (:142-158, masked-out rows zero here, `np.empty` there).  Pinned against the reference's own
    functions by oracle/seam_check.py (identical outputs for the same metric objects).  `dictionary`: NumPy, or
    anything with `.reshape`, slicing and `.compute()` (the reference tests `isinstance(dictionary, da.Array)`).
    Returns (scores, simulation_indices, information text of :77-85)."""
    dictionary_size = metric.n_dictionary_patterns  # :66
    keep_n = min(keep_n, dictionary_size)  # :67
    n_iterations = int(np.ceil(dictionary_size / n_per_iteration))  # :68
    experimental = metric.prepare_experimental(experimental)  # :70
    dictionary = dictionary.reshape((dictionary_size, -1))  # :71
    n_experimental_all = int(np.prod(experimental_nav_shape))  # :73
    n_experimental = experimental.shape[0]  # :74
    info = f"Dictionary indexing information:\n  Phase name: {phase_name}\n"  # :206-237
    if n_experimental != n_experimental_all:
        info += f"  Matching {n_experimental}/{n_experimental_all} experimental pattern(s)"
    else:
        info += f"  Matching {n_experimental_all} experimental pattern(s)"
    info += f" to {dictionary_size} dictionary pattern(s)\n  {metric}\n"

    def match_chunk(chunk, k):  # :172-203
        simulated = metric.prepare_dictionary(chunk)
        similarities = metric.match(experimental, simulated)
        idx = similarities.argtopk(k, axis=-1)
        sc = similarities.topk(k, axis=-1)
        return idx.reshape((-1, k)), sc.reshape((-1, k))

    if dictionary_size == n_per_iteration:  # :88-93 (da.compute passes NumPy arrays through)
        simulation_indices, scores = match_chunk(dictionary, keep_n)
    else:
        negative_sign = -metric.sign
        simulation_indices = np.zeros((n_experimental, keep_n), dtype=np.int32)  # :97
        scores = np.full((n_experimental, keep_n), negative_sign, dtype=metric.dtype)  # :98
        lazy = hasattr(dictionary, "compute")  # :100
        starts = np.cumsum([0] + [n_per_iteration] * (n_iterations - 1))
        ends = np.cumsum([n_per_iteration] * n_iterations)
        ends[-1] = max(ends[-1], dictionary_size)
        for start, end in zip(starts, ends):
            chunk = dictionary[start:end]
            if lazy:
                chunk = chunk.compute()  # :106-108
            idx_i, scores_i = match_chunk(chunk, min(keep_n, end - start))
            idx_i = idx_i + start  # :118
            all_scores = np.hstack((scores, scores_i))
            all_idx = np.hstack((simulation_indices, idx_i))
            best = np.argsort(negative_sign * all_scores, axis=1)[:, :keep_n]  # :123
            scores = np.take_along_axis(all_scores, best, axis=1)
            simulation_indices = np.take_along_axis(all_idx, best, axis=1)
    if metric.navigation_mask is not None:  # :142-158
        scores, simulation_indices, _ = scatter_navigation_mask(scores, simulation_indices, metric.navigation_mask, keep_n)
    return scores, simulation_indices, info


class LazyArray:
    """The little of `dask.array.Array` that `_dictionary_indexing` touches on a lazy dictionary (`reshape`, slicing
    along the first axis, `compute`, `chunksize`): Dask itself is not installed beside the system Python of the GPU box."""

    def __init__(self, array, chunk):
        self._a, self._chunk = array, int(chunk)
        self.computed = []  # (start, stop) of every chunk that was materialised

    shape = property(lambda self: self._a.shape)
    ndim = property(lambda self: self._a.ndim)
    dtype = property(lambda self: self._a.dtype)
    chunksize = property(lambda self: (min(self._chunk, self._a.shape[0]),) + self._a.shape[1:])

    def reshape(self, shape):
        out = LazyArray(self._a.reshape(shape), self._chunk)
        out.computed = self.computed
        return out

    def __getitem__(self, key):
        if not isinstance(key, slice):
            raise IndexError("LazyArray: slices along the first axis only")
        start, stop, _ = key.indices(self._a.shape[0])
        out = LazyArray(self._a[key], self._chunk)
        out.computed = self.computed
        out._span = (start, stop)
        return out

    def compute(self):
        self.computed.append(getattr(self, "_span", (0, self._a.shape[0])))
        return np.array(self._a)


def scatter_navigation_mask(scores, indices, navigation_mask, keep_n):
    """indexing/_dictionary_indexing.py:142-158: with a navigation mask the
    results are scattered into (M_all, k) arrays (uninitialised elsewhere in
    the reference; zero-filled here) and `keep_n == 1` squeezes to 1-D."""
    in_data = ~np.asarray(navigation_mask).ravel()
    n_all = in_data.size
    scores_all = np.zeros((n_all, keep_n), dtype=scores.dtype)
    idx_all = np.zeros((n_all, keep_n), dtype=indices.dtype)
    scores_all[in_data] = scores
    idx_all[in_data] = indices
    if keep_n == 1:
        scores_all = scores_all.squeeze()
        idx_all = idx_all.squeeze()
    return scores_all, idx_all, in_data


# --------------------------------------------------------------------------
# a-mask / windows
# --------------------------------------------------------------------------
def circular_window(shape):
    """filters/window.py:163-187, :249-269 (`Window("circular", shape)`):
    ones, zero where the distance to origin (shape//2) exceeds max(origin)."""
    sy, sx = shape
    oy, ox = sy // 2, sx // 2
    yy, xx = np.ogrid[:sy, :sx]
    dist = np.sqrt((yy - oy) ** 2 + (xx - ox) ** 2)
    w = np.ones(shape, dtype=np.float64)
    w[dist > max(oy, ox)] = 0
    return w


def gaussian_window_1d(n, std):
    """scipy.signal.windows.gaussian(n, std, sym=True), as reached through
    filters/window.py:167-174 with `fftbins=False`."""
    x = np.arange(n, dtype=np.float64) - (n - 1.0) / 2.0
    return np.exp(-0.5 * (x / std) ** 2)


def dynamic_background_window(std, truncate):
    """pattern/_pattern.py:604-613: n = int(truncate*std) per axis,
    outer(g, g) / (2 pi std^2), then / sum."""
    n = int(truncate * std)
    g = gaussian_window_1d(n, std)
    w = np.outer(g, g) / (2 * np.pi * std**2)
    w /= np.sum(w)
    return w


# --------------------------------------------------------------------------
# a-pre1: rescale + static background
# --------------------------------------------------------------------------
def rescale_with_min_max(pattern, imin, imax, omin, omax):
    """pattern/_pattern.py:96-111, evaluated as the `.py_func` does under
    NumPy: float32 throughout for a float32 pattern."""
    rescaled = (pattern - imin) / float(imax - imin)
    return rescaled * (omax - omin) + omin


def remove_background_subtract(pattern, background, omin, omax):
    """pattern/_pattern.py:484-495."""
    pattern = pattern - background
    return rescale_with_min_max(pattern, np.min(pattern), np.max(pattern), omin, omax)


def remove_background_divide(pattern, background, omin, omax):
    """pattern/_pattern.py:498-509."""
    pattern = pattern / background
    return rescale_with_min_max(pattern, np.min(pattern), np.max(pattern), omin, omax)


def remove_static_background(
    patterns, static_bg, operation="subtract", scale_bg=False, dtype_out=None
):
    """signals/ebsd.py:518-573 + pattern/_pattern.py:392-435, per pattern.

    patterns (..., sy, sx) of an integer or float dtype; static_bg (sy, sx) of
    the SAME dtype (ebsd.py:535-539).  Output dtype = input dtype, values
    truncated toward zero by `.astype`."""
    patterns = np.asarray(patterns)
    dtype_out = np.dtype(dtype_out or patterns.dtype)
    if np.dtype(static_bg.dtype) != patterns.dtype:
        raise ValueError(
            f"Static background dtype_out {static_bg.dtype} is not the same as "
            f"pattern dtype_out {patterns.dtype}"
        )
    if static_bg.shape != patterns.shape[-2:]:
        raise ValueError(
            f"Signal {patterns.shape[-2:]} and static background {static_bg.shape} "
            "shapes are not the same"
        )
    omin, omax = DTYPE_RANGE[dtype_out]
    bg0 = static_bg.astype(np.float32)
    flat = patterns.reshape((-1,) + patterns.shape[-2:])
    out = np.empty(flat.shape, dtype=dtype_out)
    op = remove_background_subtract if operation == "subtract" else remove_background_divide
    for i, p in enumerate(flat):
        p = p.astype(np.float32)
        bg = bg0
        if scale_bg:
            bg = rescale_with_min_max(
                bg0, np.min(bg0), np.max(bg0), np.min(p), np.max(p)
            )
        out[i] = op(p, bg, omin, omax).astype(dtype_out)
    return out.reshape(patterns.shape)


# --------------------------------------------------------------------------
# a-pre2: dynamic background
# --------------------------------------------------------------------------
def fft_filter_setup(image_shape, window):
    """filters/fft_barnes.py:29-51, :99-117."""
    from scipy.fft import next_fast_len, rfft2

    wy, wx = window.shape
    fft_shape = (
        next_fast_len(image_shape[0] + wy - 1, real=True),
        next_fast_len(image_shape[1] + wx - 1, real=True),
    )
    window_pad = np.zeros(fft_shape, dtype=np.float32)
    window_pad[:wy, :wx] = np.flipud(np.fliplr(window))
    transfer_function = rfft2(window_pad)
    offset_before = (wy - ((wy - 1) // 2) - 1, wx - ((wx - 1) // 2) - 1)
    offset_after = ((wy - 1) // 2, (wx - 1) // 2)
    return fft_shape, transfer_function, offset_before, offset_after


def pad_image(image, fft_shape, window_shape, offset_before_fft):
    """filters/fft_barnes.py:119-152: edge-replicating pad laid out for a
    circular convolution."""
    iy, ix = image.shape
    wy, wx = window_shape
    fy, fx = fft_shape
    oy, ox = offset_before_fft
    p = np.zeros(fft_shape, dtype=np.float32)
    p[0:iy, 0:ix] = image
    p[iy : iy + (wy - 1) // 2, :ix] = image[-1, :]
    p[:iy, ix : ix + (wx - 1) // 2] = np.expand_dims(image[:, -1], axis=1)
    p[fy - oy :, :ix] = image[0, :]
    p[:iy, fx - ox :] = np.expand_dims(image[:, 0], axis=1)
    p[iy : iy + (wy - 1) // 2, ix : ix + (wx - 1) // 2] = image[-1, -1]
    p[fy - oy :, ix : ix + (wx - 1) // 2] = image[0, -1]
    p[iy : iy + (wy - 1) // 2, fx - ox :] = image[-1, 0]
    p[fy - oy :, fx - ox :] = image[0, 0]
    return p


def fft_filter(image, window):
    """filters/fft_barnes.py:155-177 (`_fft_filter`) incl. its set-up."""
    from scipy.fft import irfft2, rfft2

    fft_shape, tf, ob, oa = fft_filter_setup(image.shape, window)
    p = pad_image(image, fft_shape, window.shape, ob)
    res = irfft2(rfft2(p) * tf, fft_shape)
    iy, ix = image.shape
    return np.real(res[oa[0] : oa[0] + iy, oa[1] : oa[1] + ix])


def correlate_nearest(image, window):
    """What `fft_filter` computes, written as a direct sum (the form the HIP
    kernel evaluates): out[i,j] = sum_uv W[u,v] * I[clamp(i+u-cy), clamp(j+v-cx)]
    with c = w - 1 - (w-1)//2.  Float64 accumulate; used by the tests to bound
    the FFT-vs-direct difference (tests/test_filters/test_fft_barnes.py:135-173
    pins `_fft_filter` == correlation with edge replication)."""
    iy, ix = image.shape
    wy, wx = window.shape
    cy, cx = wy - 1 - (wy - 1) // 2, wx - 1 - (wx - 1) // 2
    out = np.zeros((iy, ix), dtype=np.float64)
    img = image.astype(np.float64)
    for u in range(wy):
        yy = np.clip(np.arange(iy) + u - cy, 0, iy - 1)
        for v in range(wx):
            xx = np.clip(np.arange(ix) + v - cx, 0, ix - 1)
            out += window[u, v] * img[np.ix_(yy, xx)]
    return out


def remove_dynamic_background(
    patterns, operation="subtract", filter_domain="frequency", std=None,
    truncate=4.0, dtype_out=None,
):
    """signals/ebsd.py:645-696 + pattern/_pattern.py:438-481, per pattern."""
    patterns = np.asarray(patterns)
    dtype_out = np.dtype(dtype_out or patterns.dtype)
    sy, sx = patterns.shape[-2:]
    if std is None:
        std = sx / 8  # ebsd.py:648-649: axes_manager.signal_shape[0] = n columns
    omin, omax = DTYPE_RANGE[dtype_out]
    if filter_domain == "frequency":
        window = dynamic_background_window(std, truncate)

        def filt(p):
            return fft_filter(p, window)

    elif filter_domain == "spatial":
        from scipy.ndimage import gaussian_filter

        def filt(p):
            return gaussian_filter(p, sigma=std, truncate=truncate)

    else:
        raise ValueError(
            f"{filter_domain} must be either of ['frequency', 'spatial']"
        )
    flat = patterns.reshape((-1, sy, sx))
    out = np.empty(flat.shape, dtype=dtype_out)
    op = remove_background_subtract if operation == "subtract" else remove_background_divide
    for i, p in enumerate(flat):
        p = p.astype(np.float32)
        out[i] = op(p, filt(p), omin, omax).astype(dtype_out)
    return out.reshape(patterns.shape)


# --------------------------------------------------------------------------
# dictionary generation: projection of a master pattern onto the detector
# (all f64 like the reference)
# --------------------------------------------------------------------------
SQRT_PI = np.sqrt(np.pi)
SQRT_PI_HALF = np.sqrt(np.pi / 2)
SQRT_PI_OVER_2 = SQRT_PI / 2
TWO_OVER_SQRT_PI = 2 / SQRT_PI


def sample_to_detector_matrix(sample_tilt=70.0, tilt=0.0, azimuthal=0.0, twist=0.0):
    """detectors/_ebsd_detector.py:100-150 (+ :836-845): rows are the detector
    axes (X_d, Y_d, Z_d) in sample coordinates, starting from (Y_s, Z_s, X_s)
    and turned, in this order, about the current X_d by -sample_tilt, about X_d
    by +tilt, about Y_d by -azimuthal, about Z_d by -twist (Rodrigues formula,
    all three rows turned by each step).  Angles in degrees."""
    basis = np.array([[0, 1, 0], [0, 0, 1], [1, 0, 0]], dtype=np.float64)
    angles = np.deg2rad(np.array([-sample_tilt, tilt, -azimuthal, -twist], dtype=np.float64))
    for axis_row, angle in zip((0, 0, 1, 2), angles):
        u = basis[axis_row] / np.sqrt(np.sum(basis[axis_row] ** 2))
        c, s = np.cos(angle), np.sin(angle)
        for j in range(3):
            v = basis[j].copy()
            basis[j] = v * c + np.cross(u, v) * s + u * np.dot(u, v) * (1.0 - c)
    return basis


def gnomonic_bounds(shape, pc):
    """detectors/_ebsd_detector.py:731-818: (x_min, x_max, y_min, y_max) of the
    detector in gnomonic coordinates for a Bruker-convention PC (pcx, pcy, pcz)."""
    nrows, ncols = shape
    pcx, pcy, pcz = (np.float64(v) for v in pc)
    aspect = ncols / nrows
    return np.array(
        [-aspect * (pcx / pcz), aspect * (1 - pcx) / pcz, -(1 - pcy) / pcz, pcy / pcz],
        dtype=np.float64,
    )


def direction_cosines_fixed_pc(bounds, pcz, nrows, ncols, om_detector_to_sample, signal_mask=None):
    """signals/util/_master_pattern.py:133-204: unit vectors from the source
    point through the centre of every (kept) detector pixel, in the sample
    frame.  `signal_mask` follows that function's convention: True = keep."""
    x_scale = (bounds[1] - bounds[0]) / ncols
    y_scale = (bounds[3] - bounds[2]) / nrows
    det_gn_x = bounds[0] + np.arange(ncols) * x_scale      # np.arange(start, stop, step)
    det_gn_y = bounds[3] + np.arange(nrows) * (-y_scale)
    idx = np.arange(nrows * ncols)
    if signal_mask is not None:
        idx = idx[np.asarray(signal_mask, dtype=bool).ravel()]
    rows, cols = idx // ncols, idx % ncols
    r_g = np.empty((idx.size, 3), dtype=np.float64)
    r_g[:, 0] = (det_gn_x[cols] + x_scale / 2) * pcz
    r_g[:, 1] = (det_gn_y[rows] - y_scale / 2) * pcz
    r_g[:, 2] = pcz
    r_g = r_g @ np.asarray(om_detector_to_sample, dtype=np.float64).T
    return r_g / np.sqrt(np.sum(r_g**2, axis=-1))[:, None]


def detector_direction_cosines(shape, pc, sample_tilt=70.0, tilt=0.0, azimuthal=0.0, twist=0.0,
                               signal_mask=None):
    """signals/util/_master_pattern.py:83-124 for one PC: detector -> sample
    matrix is the inverse (transpose) of `sample_to_detector_matrix`."""
    m = sample_to_detector_matrix(sample_tilt, tilt, azimuthal, twist)
    return direction_cosines_fixed_pc(gnomonic_bounds(shape, pc), np.float64(pc[2]), shape[0], shape[1],
                                      m.T, signal_mask)


def project_patterns_varying_pc(rotations, pcs, shape, om_detector_to_sample, master_upper, master_lower,
                                rescale=False, out_min=-1, out_max=1, dtype_out=np.float32):
    """signals/util/_master_pattern.py:216-295 + :374-445: pattern i is projected
    with its own projection centre pcs[i] (Bruker convention): direction cosines
    per PC, then the single-pattern projection."""
    rotations = np.asarray(rotations, dtype=np.float64).reshape(-1, 4)
    out = np.empty((rotations.shape[0], shape[0] * shape[1]), dtype=dtype_out)
    for i, (rot, pc) in enumerate(zip(rotations, np.asarray(pcs, dtype=np.float64).reshape(-1, 3))):
        dc = direction_cosines_fixed_pc(gnomonic_bounds(shape, pc), pc[2], shape[0], shape[1], om_detector_to_sample)
        out[i] = project_patterns(rot, dc, master_upper, master_lower, rescale, out_min, out_max, dtype_out)[0]
    return out


def rotate_vector(rotation, vector):
    """_utils/numba.py:59-81: passive rotation of (n, 3) vectors by the unit
    quaternion (a, b, c, d)."""
    a, b, c, d = (np.float64(v) for v in rotation)
    x, y, z = vector[:, 0], vector[:, 1], vector[:, 2]
    aa, bb, cc, dd = a * a, b * b, c * c, d * d
    ac, ab, ad, bc, bd, cd = a * c, a * b, a * d, b * c, b * d, c * d
    out = np.empty(vector.shape, dtype=np.float64)
    out[:, 0] = (aa + bb - cc - dd) * x + 2 * ((ac + bd) * z + (bc - ad) * y)
    out[:, 1] = (aa - bb + cc - dd) * y + 2 * ((ad + bc) * x + (cd - ab) * z)
    out[:, 2] = (aa - bb - cc + dd) * z + 2 * ((ab + cd) * y + (bd - ac) * x)
    return out


def vector2lambert(v):
    """signals/util/_master_pattern.py:530-568: square Lambert (X, Y) of (n, 3)
    vectors (normalised first); (0, 0) at the poles."""
    w = v / np.sqrt(np.sum(v**2, axis=1))[:, None]
    x, y, z = w[:, 0], w[:, 1], w[:, 2]
    abs_z = np.abs(z)
    sqrt_z = np.sqrt(2 * (1 - abs_z))
    out = np.zeros((v.shape[0], 2), dtype=np.float64)
    pole = abs_z == 1
    xdom = (np.abs(y) <= np.abs(x)) & ~pole
    ydom = ~xdom & ~pole
    with np.errstate(divide="ignore", invalid="ignore"):
        sx, sy = np.sign(x), np.sign(y)
        out[xdom, 0] = (sx * sqrt_z * SQRT_PI_OVER_2)[xdom]
        out[xdom, 1] = (sx * sqrt_z * TWO_OVER_SQRT_PI * np.arctan(y / x))[xdom]
        out[ydom, 0] = (sy * sqrt_z * TWO_OVER_SQRT_PI * np.arctan(x / y))[ydom]
        out[ydom, 1] = (sy * sqrt_z * SQRT_PI_OVER_2)[ydom]
    return out


def lambert_interpolation_parameters(v, npx, npy, scale):
    """signals/util/_master_pattern.py:580-678: the four neighbouring master-
    pattern pixels of every vector and their bilinear weights.  The row index
    comes from Lambert Y, the column index from Lambert X; `int32()` truncates
    toward zero."""
    xy = scale * vector2lambert(v) / SQRT_PI_HALF
    i, j = xy[:, 1], xy[:, 0]
    nii = np.trunc(i + scale).astype(np.int32)
    nij = np.trunc(j + scale).astype(np.int32)
    niip, nijp = nii + 1, nij + 1
    niip = np.where(niip >= npx, nii, niip)
    nijp = np.where(nijp >= npy, nij, nijp)
    nii = np.where(nii < 0, niip, nii)
    nij = np.where(nij < 0, nijp, nij)
    di = i - nii + scale
    dj = j - nij + scale
    return nii, nij, niip, nijp, di, dj, 1 - di, 1 - dj


def project_patterns(rotations, direction_cosines, master_upper, master_lower, rescale=False,
                     out_min=-1, out_max=1, dtype_out=np.float32):
    """signals/util/_master_pattern.py:299-370, :449-527: one simulated pattern
    per unit quaternion: rotate the direction cosines, bilinear interpolation in
    the square-Lambert master pattern of the hemisphere the rotated vector
    points into (z >= 0: upper), optional min-max rescale of each pattern
    (pattern/_pattern.py:97-111), cast (integers truncate)."""
    rotations = np.asarray(rotations, dtype=np.float64).reshape(-1, 4)
    npy, npx = master_upper.shape  # axes_manager.signal_shape = (npx, npy), square in practice
    scale = float((npx - 1) / 2)
    out = np.empty((rotations.shape[0], direction_cosines.shape[0]), dtype=dtype_out)
    for n, rot in enumerate(rotations):
        v = rotate_vector(rot, direction_cosines)
        nii, nij, niip, nijp, di, dj, dim, djm = lambert_interpolation_parameters(v, npx, npy, scale)
        up = v[:, 2] >= 0
        pattern = np.empty(v.shape[0], dtype=np.float64)
        for sel, mp in ((up, master_upper), (~up, master_lower)):
            pattern[sel] = (
                mp[nii[sel], nij[sel]] * dim[sel] * djm[sel]
                + mp[niip[sel], nij[sel]] * di[sel] * djm[sel]
                + mp[nii[sel], nijp[sel]] * dim[sel] * dj[sel]
                + mp[niip[sel], nijp[sel]] * di[sel] * dj[sel]
            )
        if rescale:
            imin, imax = np.min(pattern), np.max(pattern)
            pattern = (pattern - imin) / float(imax - imin) * (out_max - out_min) + out_min
        out[n] = pattern.astype(dtype_out)
    return out


# --------------------------------------------------------------------------
# orientation / projection-centre refinement: objective functions and the
# SciPy Nelder-Mead solver chunk
# --------------------------------------------------------------------------
def rotation_from_euler(phi1, Phi, phi2):
    """_utils/numba.py:43-57: Bunge Euler angles (radians) -> unit quaternion
    with a non-negative scalar part."""
    sigma = 0.5 * (phi1 + phi2)
    delta = 0.5 * (phi1 - phi2)
    c, s = np.cos(0.5 * Phi), np.sin(0.5 * Phi)
    rot = np.array([c * np.cos(sigma), -s * np.cos(delta), -s * np.sin(delta), -c * np.sin(sigma)],
                   dtype=np.float64)
    return -rot if rot[0] < 0 else rot


def refinement_master_pattern(master_upper, master_lower):
    """indexing/_refinement/_refinement.py:1288-1320: hemispheres as float32;
    anything else is rescaled to [-1, 1] over its own min/max
    (pattern/_pattern.py:66-93, `rescale_intensity(mp, dtype_out=float32)`)."""
    out = []
    for mp in (master_upper, master_lower):
        if mp.dtype != np.float32:
            imin, imax = np.nanmin(mp), np.nanmax(mp)
            mp = ((mp - imin) / float(imax - imin) * 2 + -1).astype(np.float32)
        out.append(mp)
    return out


def prepare_refinement_pattern(pattern, rescale):
    """indexing/_refinement/_solvers.py:51-74: float32 cast, optional rescale
    to [-1, 1] (float32 arithmetic, pattern/_pattern.py:125-133), centring, and
    the squared norm of the centred pattern (pattern/_pattern.py:136-139)."""
    pattern = pattern.astype(np.float32)
    if rescale:
        imin, imax = np.min(pattern), np.max(pattern)
        pattern = (pattern - imin) / float(imax - imin) * 2 + -1
    pattern = pattern - np.mean(pattern)
    return pattern, np.square(pattern).sum()


def ncc_exp_centered(exp, sim, exp_squared_norm):
    """similarity_metrics/_normalized_cross_correlation.py:200-225."""
    sim = sim - np.mean(sim)
    return np.divide(np.sum(exp * sim), np.sqrt(exp_squared_norm * np.sum(np.square(sim))))


def refinement_objective(x, mode, pattern, squared_norm, master_upper, master_lower, *, direction_cosines=None,
                         rotation=None, signal_mask_keep=None, nrows=None, ncols=None,
                         om_detector_to_sample=None):
    """indexing/_refinement/_objective_functions.py:36-190: 1 - NCC between the
    centred experimental pattern and the pattern projected for the control
    variables `x`:  mode "ori": x = Euler angles, fixed `direction_cosines`;
    "pc": x = (PCx, PCy, PCz), fixed quaternion `rotation`;  "ori_pc": x =
    Euler angles + PC.  `signal_mask_keep`: 1D, True = pixel is used."""
    if mode == "ori":
        rot, dc = rotation_from_euler(*x[:3]), direction_cosines
    else:
        pc = x if mode == "pc" else x[3:]
        rot = rotation if mode == "pc" else rotation_from_euler(*x[:3])
        dc = direction_cosines_fixed_pc(gnomonic_bounds((nrows, ncols), pc), pc[2], nrows, ncols,
                                        om_detector_to_sample, signal_mask_keep)
    sim = project_patterns(rot, dc, master_upper, master_lower, False, 0, 1, np.float32)[0]
    return 1 - ncc_exp_centered(pattern, sim, squared_norm)


def refine_solver(pattern, mode, x0, master_upper, master_lower, rescale, bounds=None, method_kwargs=None,
                  **fixed):
    """indexing/_refinement/_solvers.py:79-250 (orientation), :253-330 (PC),
    :333-460 (both) with `method=scipy.optimize.minimize`: one SciPy
    Nelder-Mead run per row of `x0` (row 0 = the indexed orientation, further
    rows = its pseudo-symmetry equivalents); the best run wins (first maximum).
    SciPy is the reference's own third-party optimiser (pyproject: scipy >= 1.7)
    and is called here as there.  Returns (ncc, num_evals, *x[, best_index])."""
    import scipy.optimize

    pattern, squared_norm = prepare_refinement_pattern(pattern, rescale)
    kwargs = {"method": "Nelder-Mead"}
    kwargs.update(method_kwargs or {})
    x0 = np.atleast_2d(np.asarray(x0, dtype=np.float64))
    rotations = fixed.pop("rotation", None)
    results = []
    for i, start in enumerate(x0):
        kw = dict(kwargs)
        if bounds is not None:
            kw["bounds"] = np.atleast_3d(bounds)[i] if np.ndim(bounds) == 3 else bounds
        extra = dict(fixed)
        if rotations is not None:
            extra["rotation"] = np.atleast_2d(rotations)[i]
        res = scipy.optimize.minimize(
            fun=lambda x: refinement_objective(x, mode, pattern, squared_norm, master_upper, master_lower, **extra),
            x0=start, **kw)
        results.append(res)
    ncc_all = [1 - r.fun for r in results]
    best = int(np.argmax(ncc_all))
    out = (ncc_all[best], results[best].nfev) + tuple(results[best].x)
    return out + (best,) if len(results) > 1 else out


# --------------------------------------------------------------------------
# consumers of the indexing result
# --------------------------------------------------------------------------
def orientation_similarity_map(simulation_indices, shape, n_best=None, normalize=False, from_n_best=None,
                               footprint=None, center_index=2):
    """indexing/_orientation_similarity_map.py:30-152.  Per map point: the mean,
    over its neighbours in `footprint` (scipy.ndimage.generic_filter: footprint
    centred at shape // 2, values outside the map = -1, neighbours = footprint
    points that are inside the map and are not the centre point), of the number
    of dictionary indices shared by the n best matches of the point and of the
    neighbour (`len(np.intersect1d(a, b))`: unique values).  One layer per n from
    `n_best` down to `from_n_best`; float32; squeezed."""
    simulation_indices = np.asarray(simulation_indices)
    nav_size, keep_n = simulation_indices.shape
    if n_best is None:
        n_best = keep_n
    elif n_best > keep_n:
        raise ValueError(f"n_best {n_best} cannot be greater than keep_n {keep_n}")
    if from_n_best is None:
        from_n_best = n_best
    ny, nx = shape
    if footprint is None:
        footprint = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    footprint = np.asarray(footprint)
    offsets = [(i - footprint.shape[0] // 2, j - footprint.shape[1] // 2)
               for i in range(footprint.shape[0]) for j in range(footprint.shape[1]) if footprint[i, j]]
    osm = np.zeros((ny, nx, n_best - from_n_best + 1), dtype=np.float32)
    for layer, n in enumerate(range(n_best, from_n_best - 1, -1)):
        sets = [set(row[:n].tolist()) for row in simulation_indices]
        for y in range(ny):
            for x in range(nx):
                v = [(y + dy) * nx + (x + dx) if 0 <= y + dy < ny and 0 <= x + dx < nx else -1 for dy, dx in offsets]
                centre = v[center_index]
                counts = [len(sets[centre] & sets[p]) for p in v if p != -1 and p != centre]
                with np.errstate(invalid="ignore"):
                    value = np.float64(np.sum(counts)) / len(counts) if counts else np.float64(np.nan)
                if normalize:
                    value /= n
                osm[y, x, layer] = value
    return osm.squeeze()


# --------------------------------------------------------------------------
# comparison helper shared by the parity tests
# --------------------------------------------------------------------------
def assert_topk_parity(scores, indices, ref_scores, ref_indices, atol=1e-5, tie=2e-5):
    """The parity contract of SURVEY.md section 8(a):

    * |score - ref_score| <= atol element-wise,
    * scores non-increasing along k,
    * indices equal wherever the reference's neighbouring scores are more than
      `tie` apart; inside a group of near-tied reference scores the index SETS
      must agree (order inside a tie is unspecified in the reference).  A
      group that touches the k-th place may differ in membership (the
      (k+1)-th candidate is not visible), so there only the scores are
      compared."""
    scores = np.asarray(scores, dtype=np.float64)
    ref_scores = np.asarray(ref_scores, dtype=np.float64)
    assert scores.shape == ref_scores.shape, (scores.shape, ref_scores.shape)
    assert indices.shape == ref_indices.shape
    err = np.abs(scores - ref_scores)
    assert np.all(err <= atol), f"max score error {err.max():.3e} > {atol}"
    assert np.all(np.diff(scores, axis=1) <= 0), "scores not sorted descending"
    k = scores.shape[1]
    for m in np.nonzero(np.any(indices != ref_indices, axis=1))[0]:
        rs = ref_scores[m]
        # split into groups of near-tied neighbours
        breaks = np.nonzero(-np.diff(rs) > tie)[0] + 1
        groups = np.split(np.arange(k), breaks)
        for g in groups:
            if g[-1] == k - 1:
                continue  # open-ended group: membership not decidable
            assert set(indices[m, g]) == set(ref_indices[m, g]), (
                f"row {m}: indices {indices[m, g]} vs reference {ref_indices[m, g]}"
            )
