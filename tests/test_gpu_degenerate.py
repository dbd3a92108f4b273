"""Degenerate patterns: defined (include/kpdi.h "Degenerate patterns"), implemented (csrc/prep_device.h:
degenerate_pattern, csrc/rescore.hip), tested against the oracle, which states the same rule
(oracle/kpdi_oracle.py: degenerate_rows) - and, for patterns that are NEARLY constant, against the reference's own
arithmetic (`degenerate="reference"`): the rule is an exact test for "all kept pixels equal", not a contrast floor.

Constant / all-zero / saturated patterns (dead detector frames are ordinary in real maps) and NaN / inf pixels, on the
experimental AND on the dictionary side, inside 256-row tiles of ordinary patterns, every arithmetic, one and several
chunks, both f32 kernels.  The reference divides 0 by 0 (similarity_metrics/_normalized_cross_correlation.py:228-233,
_normalized_dot_product.py:181-194) and its `topk` ranks the NaN first (dask/array/chunk.py:167-258); the engine's
rule is "all-zero row, score exactly 0".  What must hold: the oracle's result; no NaN anywhere; and the ordinary
patterns' results are BIT-IDENTICAL to a run without any degenerate experimental pattern (nothing leaks through the
shared rejection bound, the tile, the lists)."""

import numpy as np
import pytest

from oracle import kpdi_oracle as ko

pytestmark = pytest.mark.gpu
SY, SX = 24, 20


def problem(dtype_exp, seed=3, m=300, n=3000):
    """m experimental patterns (two 256-row tiles) with degenerate ones spread among them; a dictionary with
    degenerate rows inside its tiles.  Returns (exp, dic, degenerate exp rows, degenerate dictionary rows)."""
    rng = np.random.default_rng(seed)
    if np.issubdtype(dtype_exp, np.integer):
        exp = rng.integers(0, 256, (m, SY, SX)).astype(dtype_exp)
        bad_e = {5: 0, 100: 200, 255: 255, 256: 7, 299: 0}  # row -> constant value (0 = dead, 255 = saturated)
        bad_e = {r: v for r, v in bad_e.items() if r < m}
        for r, v in bad_e.items():
            exp[r] = v
    else:
        exp = rng.random((m, SY, SX)).astype(dtype_exp)
        bad_e = {5: 0.0, 100: 0.1, 255: 1.0, 299: 0.0, 256: "nan pixel", 17: "inf pixel", 200: "all nan"}
        bad_e = {r: v for r, v in bad_e.items() if r < m}
        for r, v in bad_e.items():
            if v == "nan pixel":
                exp[r, 3, 4] = np.nan
            elif v == "inf pixel":
                exp[r, 4, 0] = np.inf
            elif v == "all nan":
                exp[r] = np.nan
            else:
                exp[r] = v
    dic = rng.random((n, SY, SX)).astype(np.float32)
    plant = {0: 0.0, 77: 0.25, 128: "nan pixel", 1000: "inf pixel", 1023: 0.0, 2999: "-inf pixel"}
    bad_d = [r for r in plant if r < n]
    for r in bad_d:        # 0.0: all zeros, degenerate for ncc AND ndp; 0.25: constant, degenerate for ncc only
        v = plant[r]
        if isinstance(v, str):
            dic[r, 5, 5] = {"nan pixel": np.nan, "inf pixel": np.inf, "-inf pixel": -np.inf}[v]
        else:
            dic[r] = v
    return exp, dic, sorted(bad_e), bad_d


def degenerate_for(metric, exp, rows):
    """Which of the planted rows are degenerate under `metric` (a non-zero constant is an ordinary pattern for ndp)."""
    flat = exp.reshape(len(exp), -1).astype(np.float64)
    out = []
    for r in rows:
        x = flat[r]
        if not np.isfinite(x).all() or (metric == "ncc" and x.std() == 0) or (metric == "ndp" and not x.any()):
            out.append(r)
    return out


def run(exp, dic, metric, keep_n, compute, chunk=None, signal_mask=None):
    import kikuchipy_amd as ka

    kw = {"dtype": np.float64} if compute == "f64" else {"compute": compute}
    r = ka.dictionary_indexing(exp, dic, metric, keep_n, n_per_iteration=chunk, signal_mask=signal_mask, device=0,
                               verbose=False, **kw)
    return r.scores, r.simulation_indices


@pytest.mark.parametrize("compute", ["f32", "f16x2", "f16", "f64"])
@pytest.mark.parametrize("metric,dtype_exp,chunk,masked", [
    ("ncc", np.uint8, None, False), ("ncc", np.float32, 700, True), ("ndp", np.uint8, 1100, False),
    ("ndp", np.float32, None, True),
])
def test_degenerate_patterns_follow_the_rule(metric, dtype_exp, chunk, masked, compute):
    exp, dic, bad_e, bad_d = problem(dtype_exp)
    signal_mask = None
    if masked:
        signal_mask = np.zeros((SY, SX), dtype=bool)
        signal_mask[:3] = True       # (the NaN / inf pixels planted above stay inside the kept area)
        signal_mask[10, 8:12] = True
    keep_n = 20
    scores, idx = run(exp, dic, metric, keep_n, compute, chunk, signal_mask)
    assert np.isfinite(scores).all() and idx.min() >= 0 and idx.max() < len(dic)
    # the oracle states the same rule
    odt = np.float64 if compute == "f64" else np.float32
    rs, ri = ko.dictionary_indexing(exp, dic, metric=metric, keep_n=keep_n, n_per_iteration=chunk, signal_mask=signal_mask,
                                    dtype=odt)
    tol = {"f32": 1e-5, "f16x2": 1e-5, "f16": 2e-3, "f64": 1e-12}[compute]
    ko.assert_topk_parity(scores, idx, rs, ri, atol=tol, tie=max(2 * tol, 2e-5) if compute != "f64" else 1e-11)
    # a degenerate experimental pattern: score exactly 0 against everything -> the lowest dictionary indices
    deg_e = degenerate_for(metric, exp, bad_e)
    assert deg_e, "the problem must contain degenerate experimental patterns for this metric"
    for r in deg_e:
        assert np.array_equal(scores[r], np.zeros(keep_n)) and not np.signbit(scores[r]).any(), (r, scores[r])
        assert np.array_equal(idx[r], np.arange(keep_n)), (r, idx[r])
    # a degenerate dictionary pattern is selected only where fewer than keep_n real scores are positive
    deg_d = degenerate_for(metric, dic, bad_d)
    ordinary = np.setdiff1d(np.arange(len(exp)), deg_e)
    picked = np.isin(idx[ordinary], deg_d)
    assert np.array_equal(scores[ordinary][picked], np.zeros(picked.sum()))
    if metric == "ndp":  # non-negative patterns: every real score is positive, a 0 never makes the best 20 of 3000
        assert not picked.any()
    # ... and nothing leaks: the ordinary patterns' results are bit-identical to a run in which the degenerate
    # experimental patterns are ordinary ones
    rng = np.random.default_rng(99)
    exp2 = exp.copy()
    for r in deg_e:
        exp2[r] = (rng.integers(0, 256, (SY, SX)) if np.issubdtype(dtype_exp, np.integer) else rng.random((SY, SX))).astype(dtype_exp)
    s2, i2 = run(exp2, dic, metric, keep_n, compute, chunk, signal_mask)
    assert np.array_equal(scores[ordinary], s2[ordinary]) and np.array_equal(idx[ordinary], i2[ordinary])


@pytest.mark.parametrize("wide", ["0", "1"])
def test_both_f32_kernels_and_a_keep_n_that_forces_degenerate_entries_in(monkeypatch, wide):
    """keep_n = the whole (small) dictionary: the degenerate dictionary patterns MUST appear - with score 0, between the
    positive and the negative real scores, lower index first among themselves."""
    monkeypatch.setenv("KPDI_F32_WIDE", wide)
    exp, dic, bad_e, bad_d = problem(np.uint8, m=260, n=40)
    dic[3] = 0.5
    dic[20] = 0.0
    dic[21, 1, 1] = np.nan
    deg_d = degenerate_for("ncc", dic, [0, 3, 20, 21])
    assert deg_d == [0, 3, 20, 21]
    scores, idx = run(exp, dic, "ncc", 40, "f32")
    rs, ri = ko.dictionary_indexing(exp, dic, metric="ncc", keep_n=40)
    ko.assert_topk_parity(scores, idx, rs, ri, atol=1e-5)
    for r in np.setdiff1d(np.arange(len(exp)), degenerate_for("ncc", exp, bad_e)):
        zeros = np.flatnonzero(scores[r] == 0)
        assert list(idx[r][zeros]) == deg_d, (r, idx[r][zeros])      # all four, ascending index
        assert (scores[r][:zeros[0]] > 0).all() and (scores[r][zeros[-1] + 1:] < 0).all()
        assert sorted(idx[r]) == list(range(40))


def test_what_the_reference_does_instead():
    """For the record (no GPU work): with the reference's arithmetic a degenerate DICTIONARY pattern is NaN against
    everybody and its `topk` puts it FIRST for every experimental pattern (NaN sorts as the largest value)."""
    exp, dic, _, _ = problem(np.uint8, m=8, n=50)
    e = ko.prepare_experimental(exp[:8], "ncc")
    with np.errstate(invalid="ignore", divide="ignore"):
        d = ko.zero_mean_normalize(dic.reshape(50, -1).astype(np.float32), degenerate="reference")
        sim = e @ d.T
    order, s = ko.reference_topk_with_nan(sim, 5)
    ordinary = [r for r in range(8) if r != 5]  # (row 5 is itself degenerate: scores 0 in the oracle's rule)
    assert np.isnan(s[ordinary, 0]).all() and set(order[ordinary, 0]) == {0}
    # the engine's rule on the same data: no NaN, the degenerate pattern is not anybody's best match
    rs, ri = ko.dictionary_indexing(exp[:8], dic, keep_n=5)
    assert np.isfinite(rs).all() and not np.isin(ri[ordinary, 0], [0]).any()


def faint_patterns(dtype, base, n=40, seed=8):
    """Patterns on a `base`-count background with ONE to three pixels off by one count: RMS contrast ~1e-7 of the mean,
    far below round 5's floor (2^-20), yet the reference correlates them (a centred delta)."""
    rng = np.random.default_rng(seed)
    p = np.full((n, SY, SX), base, dtype=dtype)
    for r in range(n):
        for _ in range(1 + r % 3):
            y, x = rng.integers(SY), rng.integers(SX)
            p[r, y, x] = p[r, y, x] + 1 if r % 2 else p[r, y, x] - 1
    return p


@pytest.mark.parametrize("compute", ["f32", "f64"])
@pytest.mark.parametrize("dtype,base", [(np.uint16, 60000), (np.uint8, 200), (np.float32, 60000.0)])
def test_no_contrast_floor(dtype, base, compute):
    """uint16 at ~60 000 counts with a single pixel off by one must correlate LIKE THE REFERENCE
    (similarity_metrics/_normalized_cross_correlation.py:228-233 evaluated by the oracle with degenerate="reference",
    i.e. no rule of this library applied), on the experimental and on the dictionary side; an exactly constant frame
    among them stays degenerate (score 0)."""
    exp = faint_patterns(dtype, base)
    exp[7] = base  # exactly constant: degenerate
    rng = np.random.default_rng(4)
    dic = rng.random((600, SY, SX)).astype(np.float32)
    faint_d = faint_patterns(np.float32, 60000.0, n=20, seed=9)
    dic[100:120] = faint_d
    keep_n = 10
    scores, idx = run(exp, dic, "ncc", keep_n, compute)
    assert np.isfinite(scores).all()
    # the reference's arithmetic, nothing else: float32 (or float64) zero-mean-normalise, matrix product, top-k
    odt = np.float64 if compute == "f64" else np.float32
    with np.errstate(invalid="ignore", divide="ignore"):
        e = ko.zero_mean_normalize(exp.reshape(len(exp), -1).astype(odt), degenerate="reference")
        d = ko.zero_mean_normalize(dic.reshape(len(dic), -1).astype(odt), degenerate="reference")
    ordinary = np.setdiff1d(np.arange(len(exp)), [7])
    assert np.isfinite(e[ordinary]).all() and np.isfinite(d).all(), "the reference itself is well defined on these"
    sim = (e[ordinary].astype(np.float64) @ d.astype(np.float64).T).astype(np.float32)
    ri, rs = ko.topk_desc(sim, keep_n)
    tol = 1e-5 if compute == "f32" else 1e-6
    ko.assert_topk_parity(scores[ordinary], idx[ordinary], rs, ri, atol=tol)
    # the faint dictionary patterns are real candidates: a delta correlates +-1 / 0.5 / ... with another delta
    assert np.abs(scores[ordinary]).max() > 0.3
    assert np.array_equal(scores[7], np.zeros(keep_n)) and np.array_equal(idx[7], np.arange(keep_n))
