"""The opt-in split-f16 arithmetic of the match kernel (KPDI_COMPUTE_F16X2)
against the same oracle, golden vectors and tolerance as the f32 path: scores
within 1e-5 absolute (BASELINE.json north_star), indices per
`assert_topk_parity`.  What changes is how a product is formed (three f16
matrix-core products per term instead of one f32 product); everything around it
- masks, normalisation, top-k, merge, multi-pass, chunking - is shared."""

import numpy as np
import pytest

from conftest import load_golden
from oracle import kpdi_oracle as ko

pytestmark = pytest.mark.gpu
ATOL = 1e-5


@pytest.fixture(scope="module")
def ctx():
    from kikuchipy_amd import _lib

    c = _lib.Context(0)
    yield c
    c.close()


def run_engine(ctx, exp, dic, metric="ncc", keep_n=20, chunk=None, signal_mask=None, navigation_mask=None):
    from kikuchipy_amd import _lib

    sy, sx = exp.shape[-2:]
    n = dic.shape[0]
    keep_n = min(keep_n, n)
    ctx.set_problem(sy, sx, signal_mask, {"ncc": _lib.METRIC_NCC, "ndp": _lib.METRIC_NDP}[metric], keep_n,
                    _lib.COMPUTE_F16X2)
    ctx.set_experimental(exp.reshape(-1, sy, sx), navigation_mask)
    chunk = chunk or n
    for start in range(0, n, chunk):
        ctx.push_dictionary_chunk(dic[start:start + chunk], start)
    return ctx.finalize(keep_n)


SYNTH_CASES = {
    "ncc_k20": dict(metric="ncc", keep_n=20),
    "ncc_k1": dict(metric="ncc", keep_n=1),
    "ncc_k5_it700": dict(metric="ncc", keep_n=5, chunk=700),
    "ndp_k20": dict(metric="ndp", keep_n=20),
    "ncc_k20_circ_it999": dict(metric="ncc", keep_n=20, signal_mask="circ", chunk=999),
    "ndp_k50": dict(metric="ndp", keep_n=50),
}


@pytest.mark.parametrize("name", sorted(SYNTH_CASES))
def test_golden_synth(ctx, name, synth_inputs):
    """The reference's own results (tests/golden/di_synth.npz)."""
    exp, dic, g = synth_inputs
    kw = dict(SYNTH_CASES[name])
    if kw.get("signal_mask") == "circ":
        kw["signal_mask"] = g["circular_mask"]
    scores, idx = run_engine(ctx, exp, dic, **kw)
    ko.assert_topk_parity(scores, idx, g[f"{name}__scores"], g[f"{name}__indices"], atol=ATOL)
    assert scores.dtype == np.float32 and idx.dtype == np.int64


def test_golden_config1_and_navmask(ctx, config1_inputs, synth_inputs):
    exp, dic, g = config1_inputs
    s, i = run_engine(ctx, exp, dic, metric="ncc", keep_n=5)
    ko.assert_topk_parity(s, i, g["ncc_k5__scores"], g["ncc_k5__indices"], atol=ATOL)
    assert list(i[:, 0]) == list(range(0, 999, 111)) and np.allclose(s[:, 0], 1, atol=ATOL)
    exp, dic, g = synth_inputs
    nav = g["nav_mask"]
    s, i = run_engine(ctx, exp.reshape(6, 8, 60, 60), dic, metric="ncc", keep_n=7, chunk=1500, navigation_mask=nav)
    ko.assert_topk_parity(s, i, g["ncc_k7_nav__scores"][~nav.ravel()], g["ncc_k7_nav__indices"][~nav.ravel()], atol=ATOL)


@pytest.mark.parametrize("m,n,sy,sx,k,chunk,metric", [
    (1, 1, 8, 8, 1, None, "ncc"),
    (3, 130, 16, 12, 20, None, "ncc"),
    (130, 257, 20, 20, 8, 100, "ndp"),
    (260, 1000, 31, 33, 20, 333, "ncc"),
    (17, 640, 60, 60, 33, 250, "ncc"),   # keep_n > 32: multi-pass path
    (5, 70, 10, 10, 70, None, "ndp"),
])
def test_vs_oracle_shapes(ctx, m, n, sy, sx, k, chunk, metric):
    rng = np.random.default_rng(m * 1000 + n)
    exp = rng.integers(0, 256, (m, sy, sx)).astype(np.uint8)
    dic = rng.random((n, sy, sx)).astype(np.float32)
    s, i = run_engine(ctx, exp, dic, metric=metric, keep_n=k, chunk=chunk)
    rs, ri = ko.dictionary_indexing(exp, dic, metric=metric, keep_n=k, n_per_iteration=chunk)
    ko.assert_topk_parity(s, i, rs, ri, atol=ATOL)


def test_wide_dynamic_range_patterns(ctx):
    """Values spanning six decades inside one pattern: the two-term float16 representation is
    scaled by 2^12 so that the low halves of all but negligible values stay normal numbers."""
    rng = np.random.default_rng(8)
    exp = (10.0 ** rng.uniform(-3, 3, (40, 24, 24))).astype(np.float32)
    dic = (10.0 ** rng.uniform(-3, 3, (500, 24, 24))).astype(np.float32)
    dic[123] = exp[5] * 7
    s, i = run_engine(ctx, exp, dic, keep_n=6)
    rs, ri = ko.dictionary_indexing(exp, dic, keep_n=6)
    ko.assert_topk_parity(s, i, rs, ri, atol=ATOL)
    assert i[5, 0] == 123 and abs(s[5, 0] - 1) < ATOL


def test_chunking_invariance_and_ties(ctx, synth_inputs):
    """Same total order as the f32 path: results do not depend on chunking, ties go to the lower index."""
    from kikuchipy_amd import _lib

    exp, dic, g = synth_inputs
    ref = run_engine(ctx, exp, dic, keep_n=20)
    ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 20, _lib.COMPUTE_F16X2)
    ctx.set_experimental(exp)
    bounds = [0, 17, 900, 901, 2049, 3000]
    for a, b in reversed(list(zip(bounds[:-1], bounds[1:]))):
        ctx.push_dictionary_chunk(dic[a:b], a)
    s, i = ctx.finalize(20)
    assert np.array_equal(i, ref[1]) and np.array_equal(s, ref[0])
    rng = np.random.default_rng(3)
    base = rng.random((50, 16, 16)).astype(np.float32)
    dup = np.concatenate([base, base, base])
    s, i = run_engine(ctx, (base[:10] * 255).astype(np.uint8), dup, keep_n=6, chunk=64)
    assert np.array_equal(i[:, :3] % 50, np.repeat(np.arange(10)[:, None], 3, axis=1))
    assert np.all(np.diff(i[:, :3], axis=1) > 0)


def test_close_to_the_f32_path(ctx, synth_inputs):
    from kikuchipy_amd import _lib

    exp, dic, g = synth_inputs
    s16, i16 = run_engine(ctx, exp, dic, keep_n=20)
    ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 20, _lib.COMPUTE_F32)
    ctx.set_experimental(exp)
    ctx.push_dictionary_chunk(dic, 0)
    s32, i32 = ctx.finalize(20)
    assert np.abs(s16 - s32).max() < 2e-6
    assert np.mean(i16 != i32) < 0.01


def test_python_interface(synth_inputs):
    import kikuchipy_amd as ka

    exp, dic, g = synth_inputs
    res = ka.dictionary_indexing(exp, dic, keep_n=20, compute="f16x2", verbose=False)
    ko.assert_topk_parity(res.scores, res.simulation_indices, g["ncc_k20__scores"], g["ncc_k20__indices"], atol=ATOL)
    m = ka.NormalizedDotProductMetric(compute="f16x2")
    res = ka.dictionary_indexing(exp, dic, metric=m, keep_n=20, verbose=False)
    ko.assert_topk_parity(res.scores, res.simulation_indices, g["ndp_k20__scores"], g["ndp_k20__indices"], atol=ATOL)
    with pytest.raises(ValueError, match="compute must be one of"):
        ka.NormalizedCrossCorrelationMetric(compute="bf16")


def test_config2_planted_matches():
    """configs[1] sizes: planted exact copies come out first with score 1, order and range hold."""
    from kikuchipy_amd import _lib

    rng = np.random.default_rng(2024)
    exp = rng.integers(0, 256, (4096, 60, 60), dtype=np.uint8)
    dic = rng.random((100000, 60, 60), dtype=np.float32)
    rows = rng.choice(4096, 32, replace=False)
    at = rng.choice(100000, 32, replace=False)
    dic[at] = exp[rows].astype(np.float32) / 255.0
    with _lib.Context(0) as c:
        c.set_problem(60, 60, None, _lib.METRIC_NCC, 20, _lib.COMPUTE_F16X2)
        c.set_experimental(exp)
        c.push_dictionary_chunk(dic, 0)
        s, i = c.finalize(20)
    assert np.array_equal(i[rows, 0], at) and np.allclose(s[rows, 0], 1, atol=ATOL)
    assert np.all(np.diff(s, axis=1) <= 0) and s.max() <= 1 + ATOL
    spot = np.array([0, 1777, 4095, rows[0]])
    rs, ri = ko.dictionary_indexing(exp[spot], dic, keep_n=20, n_per_iteration=25000)
    ko.assert_topk_parity(s[spot], i[spot], rs, ri, atol=ATOL)


def test_config5_geometry(ctx):
    """configs[4] (120x120 detector, K = 14 400: the fp16-input / fp32-accumulate configuration
    of BASELINE.md) in the split-f16 form, held to the fp32 parity bar."""
    rng = np.random.default_rng(5)
    exp = rng.integers(0, 256, (300, 120, 120), dtype=np.uint8)
    dic = rng.random((5000, 120, 120), dtype=np.float32)
    dic[4321] = exp[7].astype(np.float32) + 3.0
    s, i = run_engine(ctx, exp, dic, keep_n=20, chunk=2600)
    assert i[7, 0] == 4321 and abs(s[7, 0] - 1) < ATOL
    rows = np.array([0, 7, 150, 299])
    rs, ri = ko.dictionary_indexing(exp[rows], dic, keep_n=20)
    ko.assert_topk_parity(s[rows], i[rows], rs, ri, atol=ATOL)
