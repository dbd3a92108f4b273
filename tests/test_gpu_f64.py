"""`dtype=float64` (reference: _similarity_metric.py:244-253, _normalized_cross_correlation.py:88-183 in
float64): the f32 path screens candidates, csrc/rescore.hip rescores them in double from the raw patterns,
keeps the best-k in double and certifies it.  Checked against the oracle evaluated in float64
(oracle/kpdi_oracle.py with dtype=float64 = the reference's arithmetic): scores to 1e-12, indices exactly."""
import numpy as np
import pytest

from oracle import kpdi_oracle as ko

pytestmark = pytest.mark.gpu

TOL = 1e-12


def oracle64(exp, dic, metric, keep_n, signal_mask=None, navigation_mask=None, n_per_iteration=None):
    return ko.dictionary_indexing(exp, dic, metric=metric, keep_n=keep_n, signal_mask=signal_mask,
                                  navigation_mask=navigation_mask, n_per_iteration=n_per_iteration,
                                  dtype=np.float64)[:2]


def engine64(exp, dic, metric, keep_n, signal_mask=None, navigation_mask=None, chunk=None, ctx=None):
    from kikuchipy_amd import _lib

    own = ctx is None
    c = _lib.Context(0) if own else ctx
    try:
        code = _lib.METRIC_NCC if metric == "ncc" else _lib.METRIC_NDP
        c.set_problem(exp.shape[-2], exp.shape[-1], signal_mask, code, keep_n, _lib.COMPUTE_F64)
        c.set_experimental(exp.reshape((-1,) + exp.shape[-2:]), navigation_mask)
        n = len(dic)
        chunk = chunk or n
        for s in range(0, n, chunk):
            c.push_dictionary_chunk(dic[s:s + chunk], s)
        scores, idx = c.finalize(keep_n)
        cnt = c.counters()
    finally:
        if own:
            c.close()
    return scores, idx, cnt


def assert_exact(scores, idx, ref_s, ref_i):
    assert scores.dtype == np.float64
    assert np.abs(scores - ref_s).max() <= TOL
    # indices: identical wherever the reference's neighbouring scores are further apart than the tolerance
    same = idx == ref_i
    if not same.all():
        gap = np.minimum(np.abs(np.diff(ref_s, axis=1, prepend=np.inf)), np.abs(np.diff(ref_s, axis=1, append=-np.inf)))
        assert (gap[~same] <= 4 * TOL).all()


@pytest.mark.parametrize("m,n,sy,sx,k,chunk,metric,masked,exp_dtype,dic_dtype", [
    (1, 1, 8, 8, 1, None, "ncc", False, np.uint8, np.float32),
    (37, 500, 20, 20, 5, 170, "ncc", True, np.uint8, np.float32),
    (64, 700, 24, 18, 20, None, "ndp", False, np.uint8, np.float32),
    (300, 2100, 60, 60, 20, 1000, "ncc", False, np.uint8, np.float32),   # several launches worth of tiles
    (33, 400, 31, 33, 40, 150, "ncc", True, np.uint16, np.float64),      # keep_n + 12 > 32: two screening passes
    (20, 90, 16, 16, 50, 40, "ndp", False, np.float32, np.uint8),        # chunks smaller than keep_n + 12
    (9, 260, 120, 120, 7, None, "ncc", False, np.uint8, np.float32),
])
def test_float64_scores_and_indices(m, n, sy, sx, k, chunk, metric, masked, exp_dtype, dic_dtype):
    rng = np.random.default_rng(1000 * m + n)
    exp = rng.integers(0, 256, (m, sy, sx)).astype(exp_dtype)
    dic = (rng.random((n, sy, sx)) * 200).astype(dic_dtype) if dic_dtype == np.uint8 else rng.random((n, sy, sx)).astype(dic_dtype)
    mask = None
    if masked:
        yy, xx = np.ogrid[:sy, :sx]
        mask = np.hypot(yy - sy // 2, xx - sx // 2) > min(sy, sx) // 2
    scores, idx, cnt = engine64(exp, dic, metric, k, mask, chunk=chunk)
    ref_s, ref_i = oracle64(exp, dic, metric, min(k, n), mask, n_per_iteration=chunk)
    assert_exact(scores, idx, ref_s, ref_i)
    assert cnt["uncertified_patterns"] == 0


def test_navigation_mask_and_python_api():
    import kikuchipy_amd as ka

    rng = np.random.default_rng(5)
    exp = rng.integers(0, 256, (6, 7, 30, 30), dtype=np.uint8)
    dic = rng.random((900, 30, 30), dtype=np.float32)
    nav = np.zeros((6, 7), dtype=bool)
    nav[::2, 1::3] = True
    res = ka.dictionary_indexing(exp, dic, metric="ncc", keep_n=10, n_per_iteration=400, navigation_mask=nav,
                                 dtype=np.float64, verbose=False)
    ref_s, ref_i = oracle64(exp, dic, "ncc", 10, navigation_mask=nav, n_per_iteration=400)
    assert res.scores.dtype == np.float64
    in_data = ~nav.ravel()
    assert_exact(res.scores[in_data], res.simulation_indices[in_data], ref_s, ref_i)
    # explicit f32 arithmetic with dtype=float64 still exists, and says so
    with pytest.warns(UserWarning, match="float64"):
        res32 = ka.dictionary_indexing(exp, dic, metric="ncc", keep_n=10, dtype=np.float64, compute="f32", verbose=False)
    assert res32.scores.dtype == np.float64
    assert np.abs(res32.scores[in_data] - ref_s).max() <= 1e-5
    assert np.abs(res32.scores[in_data] - ref_s).max() > TOL


def test_equal_scores_need_more_screening_passes():
    """60 copies of one dictionary pattern among the best: the first 32 screened candidates cannot certify the
    best 20 (the 33rd may tie), so further passes run; ties keep dictionary order."""
    rng = np.random.default_rng(11)
    m, n, s = 40, 600, 20
    dic = rng.random((n, s, s), dtype=np.float32)
    exp = rng.integers(0, 256, (m, s, s), dtype=np.uint8)
    twins = rng.permutation(n)[:60]
    base = rng.random((s, s), dtype=np.float32)  # the twins; every experimental pattern is a noisy image of them
    dic[twins] = base
    exp[:] = np.clip(base * 255 + rng.normal(0, 3, (m, s, s)), 0, 255).astype(np.uint8)
    scores, idx, cnt = engine64(exp, dic, "ncc", 20)
    ref_s, ref_i = oracle64(exp, dic, "ncc", 20)
    assert np.abs(scores - ref_s).max() <= TOL
    assert np.array_equal(idx, np.broadcast_to(np.sort(twins)[:20], idx.shape))
    assert cnt["rescore_extra_passes"] >= 1
    assert cnt["uncertified_patterns"] == 0


def test_extra_passes_of_an_earlier_host_chunk():
    """Host chunks stream through two staging buffers and the look at a chunk's certification is left to the next call
    (api.hip: resolve_exact64): the 60 equal-scoring twins sit in the FIRST of four chunks, so its extra screening
    passes are queued while the second chunk's upload is already in flight - same result as one chunk."""
    rng = np.random.default_rng(21)
    m, n, s = 30, 800, 20
    dic = rng.random((n, s, s), dtype=np.float32)
    exp = rng.integers(0, 256, (m, s, s), dtype=np.uint8)
    twins = rng.permutation(200)[:60]
    base = rng.random((s, s), dtype=np.float32)
    dic[twins] = base
    exp[:] = np.clip(base * 255 + rng.normal(0, 3, (m, s, s)), 0, 255).astype(np.uint8)
    ref_s, ref_i = oracle64(exp, dic, "ncc", 20)
    for chunk in (200, 800):
        scores, idx, cnt = engine64(exp, dic, "ncc", 20, chunk=chunk)
        assert np.abs(scores - ref_s).max() <= TOL
        assert np.array_equal(idx, np.broadcast_to(np.sort(twins)[:20], idx.shape))
        assert cnt["rescore_extra_passes"] >= 1 and cnt["uncertified_patterns"] == 0


@pytest.mark.parametrize("n_near", [40, 300])
def test_near_ties_below_the_f32_resolution(monkeypatch, n_near):
    """Adversarial for the certification: dictionary patterns that differ from one base pattern by 1e-8 .. 1e-5
    relative - their float32 scores cannot rank them, their float64 scores can.  Whatever the engine returns with
    `uncertified_patterns == 0` must BE the float64 best-20 (scores, indices, order); where the near-copies
    outnumber what the screening passes can rescore (300 of them against 32 + 3 x 32) it must SAY so.  With the
    worst-case bound (the default: `f64_certificate == 2`, a proof for any data) and with the statistical one
    (KPDI_F64_EPS=statistical: 8 x the largest |f32 - f64| seen)."""
    rng = np.random.default_rng(12)
    m, n, s = 24, 900, 24
    dic = rng.random((n, s, s), dtype=np.float32)
    base = rng.random((s, s)).astype(np.float32)
    near = rng.permutation(n)[:n_near]
    for j, d in enumerate(near):  # near-copies: each differs in a few pixels by a few ulps
        t = base.copy()
        px = rng.integers(0, s * s, 3)
        t.ravel()[px] *= np.float32(1 + (j + 1) * 3e-8)
        dic[d] = t
    exp = np.clip(base * 255 + rng.normal(0, 2, (m, s, s)), 0, 255).astype(np.uint8)
    ref_s, ref_i = oracle64(exp, dic, "ncc", 20)
    for mode in (None, "statistical"):
        if mode:
            monkeypatch.setenv("KPDI_F64_EPS", mode)
        scores, idx, cnt = engine64(exp, dic, "ncc", 20)
        # the bound in force is read from the environment by every kpdi_set_problem and reported (a process-wide
        # static used to make the second leg re-run the statistical bound silently)
        assert cnt["f64_certificate"] == (1 if mode else 2), (mode, cnt["f64_certificate"])
        if n_near == 40:
            assert cnt["uncertified_patterns"] == 0 and cnt["rescore_extra_passes"] >= 1, (mode, cnt)
        else:
            assert cnt["uncertified_patterns"] > 0, (mode, cnt)  # honest: more near-ties than it can rescore
        if cnt["uncertified_patterns"] == 0:
            assert_exact(scores, idx, ref_s, ref_i)


def test_config2_shape_sample():
    """4096 x 20 000 x 60 x 60 (a fifth of configs[1]'s dictionary): 256 rows against the float64 C oracle."""
    from oracle import c_oracle

    rng = np.random.default_rng(2024)
    exp = rng.integers(0, 256, (4096, 60, 60), dtype=np.uint8)
    dic = rng.random((20000, 60, 60), dtype=np.float32)
    scores, idx, cnt = engine64(exp, dic, "ncc", 20)
    rows = np.arange(0, 4096, 16)
    ref_s, ref_i = c_oracle.rows_topk_f64(exp, dic, rows, "ncc", 20)
    assert np.abs(scores[rows].astype(np.float32) - ref_s).max() <= 1e-7  # (the C oracle hands out float32)
    assert np.array_equal(idx[rows], ref_i)
    assert cnt["uncertified_patterns"] == 0 and cnt["rescore_extra_passes"] == 0


def test_single_rank_communicator_and_resident_refusal():
    import kikuchipy_amd as ka
    from kikuchipy_amd import _lib

    rng = np.random.default_rng(3)
    exp = rng.integers(0, 256, (50, 20, 20), dtype=np.uint8)
    dic = rng.random((300, 20, 20), dtype=np.float32)
    ref_s, ref_i = oracle64(exp, dic, "ncc", 8)
    with _lib.Context(0) as c:
        c.comm_init(0, 1, _lib.Context.comm_unique_id())
        scores, idx, _ = engine64(exp, dic, "ncc", 8, ctx=c)
        assert_exact(scores, idx, ref_s, ref_i)
        # a held chunk keeps only the prepared form: rescoring cannot read it
        c.hold_dictionary_chunk(dic, 0)
        with pytest.raises(_lib.KpdiError, match="RAW"):
            c.sweep_held()
    with pytest.raises(ValueError, match="f64"):
        ka.ResidentDictionary(dic, "ncc", compute="f64")


def test_recorded_preprocessing_then_float64():
    """Recorded background removal runs (fused with the f32 preparation) before the screen; the rescoring kernel reads
    the PROCESSED patterns, like the reference's float64 metric after `remove_*_background`."""
    from kikuchipy_amd import _lib

    rng = np.random.default_rng(8)
    exp = rng.integers(0, 256, (70, 40, 40), dtype=np.uint8)
    dic = rng.random((800, 40, 40), dtype=np.float32)
    with _lib.Context(0) as c:
        c.set_problem(40, 40, None, _lib.METRIC_NCC, 10, _lib.COMPUTE_F64)
        c.set_experimental(exp)
        c.remove_static_background(rng.integers(1, 256, (40, 40)).astype(np.float32), _lib.OP_SUBTRACT, False)
        c.remove_dynamic_background(_lib.OP_SUBTRACT, _lib.DOMAIN_FREQUENCY, 0.0, 4.0)
        c.push_dictionary_chunk(dic[:500], 0)
        c.push_dictionary_chunk(dic[500:], 500)
        scores, idx = c.finalize(10)
        processed = c.get_experimental()
        assert c.counters()["uncertified_patterns"] == 0
    assert not np.array_equal(processed, exp)
    ref_s, ref_i = oracle64(processed, dic, "ncc", 10, n_per_iteration=500)
    assert_exact(scores, idx, ref_s, ref_i)
