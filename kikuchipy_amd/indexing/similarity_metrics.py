"""Similarity metrics of the dictionary-indexing path, backed by libkpdi.

Mirror of the reference's plugin interface (paths under
/root/reference/src/kikuchipy/indexing/similarity_metrics/):

* `SimilarityMetric`                      <- _similarity_metric.py:23-253
* `NormalizedCrossCorrelationMetric`      <- _normalized_cross_correlation.py:26-226
* `NormalizedDotProductMetric`            <- _normalized_dot_product.py:25-194

Same attribute names, same `__repr__`, same error text, so the objects can be
handed to an unmodified `kikuchipy.indexing._dictionary_indexing._dictionary_indexing`
(which only needs `prepare_experimental`, `prepare_dictionary`, `match`, and on
the result of `match`: `.argtopk(k, axis=-1)`, `.topk(k, axis=-1)`, then
`.reshape`; indexing/_dictionary_indexing.py:193-201).  What differs is WHERE
the work happens: `prepare_experimental` uploads the patterns to the GPU once,
`match` returns a lazy proxy, and the first `argtopk`/`topk` call on it runs
ONE fused launch sequence (normalise chunk -> f32 MFMA GEMM with in-register
top-k -> merge) and memoises both results.  The (M x N) similarity matrix is
never formed.

Arithmetic is float32 on the MFMA pipe; with `dtype=float64` (as in the reference:
float64 arithmetic) the float32 path only screens candidates, which are rescored
in float64 from the raw patterns and certified (`compute="f64"`, csrc/rescore.hip).
"""

import abc
import atexit
import os
import queue
import threading
import warnings
import weakref

import numpy as np

from kikuchipy_amd import _lib


class SimilarityMetric(abc.ABC):
    """Abstract similarity metric (interface of the reference's
    `kikuchipy.indexing.SimilarityMetric`).

    Masks follow the reference's convention: `True` = excluded, for both the
    navigation mask (patterns) and the signal mask (detector pixels).
    """

    _allowed_dtypes = []
    _sign = None

    def __init__(self, n_experimental_patterns=None, n_dictionary_patterns=None, navigation_mask=None,
                 signal_mask=None, dtype="float32", rechunk=False):
        self.n_experimental_patterns = n_experimental_patterns
        self.n_dictionary_patterns = n_dictionary_patterns
        self.navigation_mask = navigation_mask
        self.signal_mask = signal_mask
        self.dtype = dtype
        self.rechunk = rechunk

    def __repr__(self):
        better = {1: "greater is better", -1: "lower is better"}[self.sign]
        return (
            f"{type(self).__name__}: {np.dtype(self.dtype).name}, {better}, "
            f"rechunk: {self.rechunk}, "
            f"navigation mask: {self.navigation_mask is not None}, "
            f"signal mask: {self.signal_mask is not None}"
        )

    @property
    def dtype(self):
        return self._dtype

    @dtype.setter
    def dtype(self, value):
        self._dtype = np.dtype(value)

    @property
    def allowed_dtypes(self):
        return self._allowed_dtypes

    @property
    def sign(self):
        return self._sign

    @abc.abstractmethod
    def prepare_experimental(self, *args, **kwargs):
        return NotImplemented  # pragma: no cover

    @abc.abstractmethod
    def prepare_dictionary(self, *args, **kwargs):
        return NotImplemented  # pragma: no cover

    @abc.abstractmethod
    def match(self, *args, **kwargs):
        return NotImplemented  # pragma: no cover

    def raise_error_if_invalid(self):
        allowed = self.allowed_dtypes
        if len(allowed) != 0 and self.dtype not in allowed:
            raise ValueError(
                f"Data type {self.dtype} not among supported data types {allowed}"
            )


class PreparedExperimental:
    """Handle of the experimental patterns resident on the GPU.  Exposes
    `.shape` because `_dictionary_indexing` reads `experimental.shape[0]`
    (indexing/_dictionary_indexing.py:74)."""

    def __init__(self, metric, n_patterns, n_pixels):
        self.metric = metric
        self.shape = (n_patterns, n_pixels)


class DictionaryChunk:
    """A dictionary chunk waiting to be matched (still on the host)."""

    def __init__(self, patterns):
        self.patterns = patterns
        self.shape = patterns.shape


class Similarities:
    """Stands in for the (M x n_chunk) similarity matrix.  `argtopk`/`topk`
    trigger one fused GPU sweep over the chunk and share its result."""

    def __init__(self, metric, chunk):
        self._metric = metric
        self._chunk = chunk
        self._cache = {}
        self.shape = (metric._engine_m, chunk.shape[0])

    def _run(self, k):
        if k not in self._cache:
            self._cache[k] = self._metric._match_chunk(self._chunk.patterns, int(k))
        return self._cache[k]

    @staticmethod
    def _check_axis(axis):
        if axis not in (-1, 1):
            raise ValueError("the fused engine ranks along the dictionary axis (axis=-1) only")

    def argtopk(self, k, axis=-1):
        self._check_axis(axis)
        return self._run(k)[1]

    def topk(self, k, axis=-1):
        self._check_axis(axis)
        return self._run(k)[0]


_LIVE_LOOKAHEADS = weakref.WeakSet()


@atexit.register
def _cancel_lookaheads_at_exit():
    # (a loop abandoned half-way leaves a worker inside the engine: it must have left before the interpreter tears the
    # HIP runtime down under it)
    for la in list(_LIVE_LOOKAHEADS):
        try:
            la.cancel()
        except Exception:  # noqa: BLE001
            pass


class _LookAhead:
    """The sweeps of the dictionary chunks the reference's loop is ABOUT to ask for, run ahead of it on a thread of
    their own (the drop-in seam).

    The loop of indexing/_dictionary_indexing.py:105-128 hands the metric one chunk at a time - `dictionary[start:end]`,
    a VIEW of the caller's array for a NumPy dictionary - and merges the chunk's best-k on the host before it slices the
    next one, so a metric that only works when asked leaves the GPU idle during every host merge and the host idle
    during every upload and sweep.  After the first chunk of a call the next ones are predictable: same number of rows,
    adjacent in the same buffer, up to `n_dictionary_patterns` rows in all.  The worker uploads and sweeps them in that
    order - the upload of chunk j + 1 overlapping the sweep of chunk j (`finalize_async` / `finalize_wait`) - and
    `take()` hands a result over when the loop asks for exactly that chunk (same address, rows and k).  A request
    for anything else cancels the look-ahead (its results are dropped) and is served the ordinary way.  At most two
    finished results wait (+ two chunks in flight): that is all a wrong guess can waste.  Same calls into the
    engine per chunk as without it, so the results are the same bit for bit ($KPDI_SEAM_LOOKAHEAD=0 switches it off).

    HAZARD, and what guards it.  The worker reads rows of the caller's buffer that `match()` has not been handed yet.
    The reference's loop never writes to the dictionary, but a CUSTOM loop around the metric may fill or update a
    pre-allocated buffer chunk by chunk - the look-ahead would then have swept stale rows.  Every chunk is therefore
    fingerprinted when the worker reads it (`_fingerprint`: 4096 evenly spaced 8-byte words, ~20 us), and `take()` only
    hands a result over if the chunk `match()` was given still has that fingerprint; otherwise the look-ahead is
    cancelled and the chunk is swept the ordinary way.  A sampled fingerprint sees a refilled or rescaled chunk, not a
    single modified pixel between samples: a loop that edits its dictionary in place at that granularity must set
    $KPDI_SEAM_LOOKAHEAD=0 (or pass a read-only array - `flags.writeable = False` - and keep it so).
    """

    def __init__(self, ctx, owner, first_row, rows, total_rows, sig_shape, k, pipelined=True):
        self._ctx = ctx
        self._pipelined = pipelined  # False (float64 arithmetic: no finalize_async): one chunk at a time, still beside the host merge
        self._flat = owner.reshape(-1)  # (a view: the owner is C-contiguous; keeps the caller's buffer alive)
        self._row_elems = int(np.prod(sig_shape))
        self._sig_shape = tuple(sig_shape)
        self._rows, self._total, self._k = int(rows), int(total_rows), int(k)
        self._next = int(first_row)        # next row the CONSUMER will ask for
        self._results = queue.Queue(maxsize=2)
        self._stop = False
        self._thread = threading.Thread(target=self._run, args=(int(first_row),), daemon=True, name="kpdi-lookahead")
        _LIVE_LOOKAHEADS.add(self)
        self._thread.start()

    def _chunk(self, row):
        n = min(self._rows, self._total - row)
        a = self._flat[row * self._row_elems:(row + n) * self._row_elems]
        return a.reshape((n,) + self._sig_shape)

    @staticmethod
    def _fingerprint(chunk):
        """4096 evenly spaced 8-byte words of a C-contiguous chunk (all of it when it is smaller), as bytes."""
        raw = chunk.reshape(-1).view(np.uint8)
        words = raw[:raw.size - raw.size % 8].view(np.uint64)
        if words.size <= 4096:
            return raw.tobytes()
        return words[::words.size // 4096][:4096].tobytes() + raw[-8:].tobytes()

    def _run(self, row):
        ctx, pending = self._ctx, None
        try:
            while row < self._total and not self._stop:
                chunk = self._chunk(row)
                fp = self._fingerprint(chunk)  # (what the worker saw: compared with what match() is handed, take())
                k_run = min(self._k, len(chunk))
                ctx.set_keep_n(k_run)
                ctx.set_dictionary_size(0)
                ctx.push_dictionary_chunk(chunk, 0)
                if not self._pipelined:
                    self._put((row, ctx.finalize(k_run), fp))
                    row += len(chunk)
                    continue
                ticket = ctx.finalize_async(k_run)
                if pending is not None:
                    self._put((pending[1], ctx.finalize_wait(pending[0]), pending[2]))
                pending = (ticket, row, fp)
                row += len(chunk)
            if pending is not None:
                res = ctx.finalize_wait(pending[0])  # (always collected: the slot must be free for whoever comes next)
                self._put((pending[1], res, pending[2]))
        except BaseException as e:  # noqa: BLE001 - handed to the consumer, which raises it in the caller's thread
            if pending is not None:  # the chunk before the failing one is still good (and its result slot must not stay taken)
                try:
                    self._put((pending[1], ctx.finalize_wait(pending[0]), pending[2]))
                except Exception:  # noqa: BLE001
                    pass
            self._put((None, e, None))

    def _put(self, item):
        while not self._stop:  # (a cancelled look-ahead drops what it has: nobody will ask)
            try:
                return self._results.put(item, timeout=0.1)
            except queue.Full:
                pass

    @property
    def exhausted(self):
        return self._next >= self._total

    def expects(self, patterns, k):
        """Whether `patterns` is exactly the chunk whose result comes next."""
        if self.exhausted or int(k) != self._k:
            return False
        want = self._chunk(self._next)
        return (patterns.shape == want.shape and patterns.dtype == want.dtype and patterns.flags.c_contiguous
                and patterns.ctypes.data == want.ctypes.data)

    def take(self, patterns):
        """The result of the chunk that comes next - `patterns`, as `expects()` has established - or None when its rows
        are no longer what the worker read (the caller wrote to its buffer in between): the look-ahead is then over."""
        row, res, fp = self._results.get()
        if row is None:
            raise res
        assert row == self._next
        if fp != self._fingerprint(patterns):
            self.cancel()
            return None
        self._next += min(self._rows, self._total - row)
        if self.exhausted:
            self._thread.join()
        return res

    def cancel(self):
        """Stop running ahead; wait until the worker has left the engine (its results are dropped)."""
        self._stop = True
        while self._thread.is_alive():
            try:
                self._results.get(timeout=0.05)
            except queue.Empty:
                pass
        self._thread.join()


class _HipMetric(SimilarityMetric):
    _allowed_dtypes = [np.float32, np.float64]
    _sign = 1
    _metric_code = None

    COMPUTE_MODES = {"f32": _lib.COMPUTE_F32, "f16x2": _lib.COMPUTE_F16X2, "f16": _lib.COMPUTE_F16,
                     "f64": _lib.COMPUTE_F64}

    def __init__(self, *args, device=0, devices=None, context=None, compute=None, **kwargs):
        """device, devices
            The GPU the engine context is created on, or - `devices="all"` / a list of ids - the GPUs
            of a `kikuchipy_amd._lib.Group`: the dictionary is then sharded over them inside this one
            process and the per-device best-k lists are merged by an in-process RCCL all-gather (or peer
            copies); results are identical to one device's.  `context`: an existing engine instead.
        compute
            Arithmetic of the match kernel (not part of the reference's interface).  None (default):
            follows `dtype` like the reference does - "f64" for `dtype=float64`, else "f32".
            "f64" = float64 arithmetic: the float32 path screens keep_n + 12 candidates per pattern and
            dictionary chunk, those are rescored in float64 from the raw patterns and the result is
            certified (csrc/rescore.hip); scores agree with a float64 evaluation to ~1e-15.
            "f32" = exact float32 products on the f32 matrix cores; "f16x2" = every
            prepared value split into two float16 (22 significant bits), three float16
            matrix-core products per term, float32 accumulation: ~2.5x the throughput, scores
            within ~1e-6 of the float32 path; "f16" = every prepared value rounded to ONE float16
            (reduced precision: scores within ~1e-3, near-ties may rank differently), one float16
            matrix-core product per term."""
        super().__init__(*args, **kwargs)
        if compute is not None and compute not in self.COMPUTE_MODES:
            raise ValueError(f"compute must be one of {sorted(self.COMPUTE_MODES)} or None, not {compute!r}")
        self.compute = compute
        self._device = 0 if device is None else device
        self._devices = devices
        self._ctx = context
        self._engine_m = 0
        self._problem = None
        self._lookahead = None
        self._lookahead_off = False  # set by a wrong guess, until the next prepare_experimental
        self._rows_seen = 0   # dictionary rows the loop has asked for since prepare_experimental
        self.lookahead_hits = 0  # chunks that were served from the look-ahead (diagnostics, tests)

    # ------------------------------------------------------------------ engine
    @property
    def context(self):
        """The libkpdi context (created on first use: needs a GPU).  Asking for it ends a look-ahead that is still
        running: the engine is then the caller's alone."""
        self._cancel_lookahead()
        return self._engine()

    def _engine(self):
        if self._ctx is None:
            # (a metric made by `dictionary_indexing` for one call takes the idle engine an earlier call left: _lib.acquire_engine)
            make = _lib.acquire_engine if getattr(self, "_pooled_engine", False) else _lib.make_engine
            self._ctx = make(self._device, self._devices)
        return self._ctx

    @property
    def effective_compute(self):
        """The arithmetic in use: `compute`, or what `dtype` asks for when that is None."""
        if self.compute is not None:
            return self.compute
        return "f64" if np.dtype(self.dtype) == np.float64 else "f32"

    def _set_problem(self, sig_shape, keep_n):
        sm = None if self.signal_mask is None else np.asarray(self.signal_mask)
        if sm is not None and sm.shape != tuple(sig_shape):
            raise ValueError(
                f"The signal mask shape {sm.shape} and the detector shape {tuple(sig_shape)} must be identical"
            )
        self._engine().set_problem(sig_shape[0], sig_shape[1], sm, self._metric_code, keep_n,
                                 self.COMPUTE_MODES[self.effective_compute])
        self._problem = tuple(sig_shape)

    def _match_chunk(self, patterns, k):
        """The best `k` of one chunk.  The reference's loop asks for `min(keep_n, end - start)` entries where `end` is
        the chunk's NOMINAL end (indexing/_dictionary_indexing.py:104, :112): for a last chunk shorter than `keep_n`
        that is more than the chunk holds.  Dask's `topk` then yields only as many columns as there are patterns (and
        announces `k`, so the loop's `.reshape((-1, k))` is a no-op on it); a NumPy result cannot change shape behind
        the loop's back, so the missing columns are returned as entries that can never be selected - score -inf -
        which the host merge of :120-128 drops (`keep_n <= dictionary size`, :67, guarantees enough real ones)."""
        ctx = self._engine()
        n = patterns.shape[0]
        k_run = min(k, n)
        la, self._lookahead = self._lookahead, None
        served = None
        if la is not None and la.expects(patterns, k):
            served = la.take(patterns)  # swept while the caller was merging the previous chunk (_LookAhead)
            if served is None:
                self._lookahead_off = True  # the caller writes to its dictionary between calls: no running ahead (until the next call)
                la = None
        if served is not None:
            scores, indices = served
            self.lookahead_hits += 1
            if not la.exhausted:
                self._lookahead = la
        else:
            if la is not None:
                la.cancel()
                self._lookahead_off = True  # a wrong guess: this is not the loop the look-ahead was made for (until the next call)
            ctx.set_keep_n(k_run)
            # (inside the reference's loop every chunk is a sweep of its own, collected at once: a group cuts it over its
            # members only when the pieces are worth a launch each - the rule for an unannounced dictionary size)
            ctx.set_dictionary_size(0)
            ctx.push_dictionary_chunk(patterns, 0)
            scores, indices = ctx.finalize(k_run)
            self._lookahead = self._start_lookahead(patterns, k)
        self._rows_seen += n
        scores = scores.astype(self.dtype, copy=False)
        if k_run < k:
            pad = ((0, 0), (0, k - k_run))
            scores = np.pad(scores, pad, constant_values=-np.inf)
            indices = np.pad(indices, pad, constant_values=0)
        return scores, indices

    def _start_lookahead(self, patterns, k):
        """A `_LookAhead` over the chunks behind `patterns` - the chunk just served the ordinary way - or None when they
        cannot be predicted: not a view into a larger C-contiguous array of the same dtype, nothing left of the
        `n_dictionary_patterns` rows, a group of devices.  (float64 arithmetic has no pipelined hand-over: its look-ahead
        sweeps one chunk at a time - still beside the caller's host merge.)"""
        ctx = self._engine()
        if self._lookahead_off or os.environ.get("KPDI_SEAM_LOOKAHEAD", "1") == "0" or hasattr(ctx, "members") \
                or not hasattr(ctx, "finalize_async"):
            return None
        total, n = self.n_dictionary_patterns, patterns.shape[0]
        if total is None or not patterns.flags.c_contiguous or n < 1:
            return None
        first, seen = self._rows_seen, self._rows_seen + n  # rows of the dictionary: this chunk = [first, seen)
        if seen >= total:
            return None
        owner = patterns
        while isinstance(owner.base, np.ndarray):
            owner = owner.base
        if owner is patterns or owner.dtype != patterns.dtype or not owner.flags.c_contiguous:
            return None
        row_bytes = patterns[0].nbytes
        off = patterns.ctypes.data - owner.ctypes.data
        if row_bytes == 0 or off < 0 or off % row_bytes:
            return None
        row0 = off // row_bytes - first  # the dictionary's row 0 within the owner
        if row0 < 0 or (row0 + total) * row_bytes > owner.nbytes:
            return None  # (the rest of the dictionary would lie outside this buffer: not the layout the loop slices)
        return _LookAhead(ctx, owner, row0 + seen, n, row0 + total, patterns.shape[1:], k,
                          pipelined=self.effective_compute != "f64")

    def _cancel_lookahead(self):
        la, self._lookahead = self._lookahead, None
        if la is not None:
            la.cancel()

    def close(self):
        """Stop a look-ahead that is still running (a loop that was abandoned half-way) and close the engine."""
        self._cancel_lookahead()
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None

    def __del__(self):
        try:
            self._cancel_lookahead()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    # ------------------------------------------------------------------ plugin API
    def __call__(self, experimental, dictionary):
        """Full similarity matrix is not available from the fused engine; use
        `dictionary_indexing` or `match(...).topk(k)`."""
        experimental = self.prepare_experimental(experimental)
        dictionary = np.asarray(dictionary).reshape((self.n_dictionary_patterns,) + self._problem)
        return self.match(experimental, self.prepare_dictionary(dictionary))

    def prepare_experimental(self, patterns):
        """Upload and keep the experimental patterns resident; casting, masking
        and normalisation (_normalized_cross_correlation.py:88-128) run on the GPU
        when the first dictionary chunk arrives."""
        self.raise_error_if_invalid()
        self._cancel_lookahead()  # (a new call of the loop: nothing of the last one is wanted any more)
        self._lookahead_off = False
        self._rows_seen = 0
        if np.dtype(self.dtype) == np.float64 and self.effective_compute != "f64":
            warnings.warn(
                f"dtype=float64 with compute={self.effective_compute!r}: the metric is evaluated in that arithmetic "
                "and the scores are returned as float64 values of it - within the parity contract of the "
                "reference's float64 evaluation, not a float64 computation (compute=None or 'f64' is one)",
                UserWarning, stacklevel=2)
        if hasattr(patterns, "compute"):
            patterns = patterns.compute()
        patterns = np.asarray(patterns)
        if patterns.ndim < 2:
            raise ValueError("experimental patterns need at least the two detector axes")
        sig_shape = patterns.shape[-2:]
        n = self.n_experimental_patterns
        if n is None:
            n = max(int(np.prod(patterns.shape[:-2])), 1)
        patterns = patterns.reshape((n,) + sig_shape)
        self._set_problem(sig_shape, 1)
        self._engine().set_experimental(patterns, self.navigation_mask)
        self._engine_m = self._engine().n_experimental
        n_pix = int(np.prod(sig_shape))
        if self.signal_mask is not None:
            n_pix = int((~np.asarray(self.signal_mask, dtype=bool)).sum())
        return PreparedExperimental(self, self._engine_m, n_pix)

    def prepare_dictionary(self, patterns):
        """Nothing happens on the host (_normalized_cross_correlation.py:130-159
        runs on the GPU inside the sweep); the caller's array is never modified."""
        if hasattr(patterns, "compute"):
            patterns = patterns.compute()
        patterns = np.asarray(patterns)
        if self._problem is not None:
            patterns = patterns.reshape((patterns.shape[0],) + self._problem)
        return DictionaryChunk(patterns)

    def match(self, experimental, dictionary):
        if not isinstance(experimental, PreparedExperimental) or experimental.metric is not self:
            raise ValueError("`experimental` must come from this metric's prepare_experimental()")
        if not isinstance(dictionary, DictionaryChunk):
            dictionary = self.prepare_dictionary(dictionary)
        return Similarities(self, dictionary)


class NormalizedCrossCorrelationMetric(_HipMetric):
    r"""Normalized cross-correlation (Pearson correlation coefficient)

    .. math:: r = \frac{\sum_i (x_i - \bar{x})(y_i - \bar{y})}
                       {\sqrt{\sum_i (x_i - \bar{x})^2}\sqrt{\sum_i (y_i - \bar{y})^2}}

    evaluated as zero-mean/unit-norm rows and one f32 MFMA GEMM on the GPU
    (reference: _normalized_cross_correlation.py:26-226)."""

    _metric_code = _lib.METRIC_NCC


class NormalizedDotProductMetric(_HipMetric):
    r"""Normalized dot product

    .. math:: \rho = \frac{\langle \mathbf{X}, \mathbf{Y} \rangle}{||\mathbf{X}|| \cdot ||\mathbf{Y}||}

    (reference: _normalized_dot_product.py:25-194; only the L2 norm is removed,
    not the mean)."""

    _metric_code = _lib.METRIC_NDP


METRICS = {"ncc": NormalizedCrossCorrelationMetric, "ndp": NormalizedDotProductMetric}
