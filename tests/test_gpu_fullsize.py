"""BASELINE.json's full sizes on the GPU, checked through size-independent
properties (the oracle cannot sweep these sizes in seconds): planted exact
matches, invariance to chunking, and an oracle spot check of a few rows."""

import numpy as np
import pytest

from oracle import kpdi_oracle as ko

pytestmark = pytest.mark.gpu
ATOL = 1e-5


def sweep(ctx, exp, dic, metric, keep_n, chunks=1, signal_mask=None):
    from kikuchipy_amd import _lib

    sy, sx = exp.shape[-2:]
    ctx.set_problem(sy, sx, signal_mask, {"ncc": _lib.METRIC_NCC, "ndp": _lib.METRIC_NDP}[metric], keep_n)
    ctx.set_experimental(exp)
    bounds = np.linspace(0, len(dic), chunks + 1).astype(int)
    for a, b in zip(bounds[:-1], bounds[1:]):
        ctx.push_dictionary_chunk(dic[a:b], int(a))
    return ctx.finalize(keep_n)


def spot_check(exp, dic, rows, metric, keep_n, scores, idx, signal_mask=None):
    rs, ri = ko.dictionary_indexing(exp[rows], dic, metric=metric, keep_n=keep_n, n_per_iteration=25000,
                                    signal_mask=signal_mask)
    ko.assert_topk_parity(scores[rows], idx[rows], rs, ri, atol=ATOL)


@pytest.fixture(scope="module")
def config2():
    """configs[1]: 4096 x 100k x 60x60 (SURVEY.md 8(d) generator), with 32 experimental
    patterns planted into the dictionary as exact (rescaled) copies."""
    rng = np.random.default_rng(2024)
    exp = rng.integers(0, 256, (4096, 60, 60), dtype=np.uint8)
    dic = rng.random((100000, 60, 60), dtype=np.float32)
    planted_rows = rng.choice(4096, 32, replace=False)
    planted_at = rng.choice(100000, 32, replace=False)
    dic[planted_at] = exp[planted_rows].astype(np.float32) / 255.0
    return exp, dic, planted_rows, planted_at


def test_config2_properties(config2):
    from kikuchipy_amd import _lib

    exp, dic, planted_rows, planted_at = config2
    with _lib.Context(0) as ctx:
        s1, i1 = sweep(ctx, exp, dic, "ncc", 20)
        s4, i4 = sweep(ctx, exp, dic, "ncc", 20, chunks=7)
    # chunking invariance: bit-identical
    assert np.array_equal(i1, i4) and np.array_equal(s1, s4)
    # planted copies come out first with score 1 (affine copies under NCC)
    assert np.array_equal(i1[planted_rows, 0], planted_at)
    assert np.allclose(s1[planted_rows, 0], 1, atol=ATOL)
    # order and range
    assert np.all(np.diff(s1, axis=1) <= 0) and s1.max() <= 1 + ATOL and s1.min() >= -1 - ATOL
    assert i1.min() >= 0 and i1.max() < len(dic)
    assert all(len(set(r)) == 20 for r in i1[::97])
    rows = np.concatenate([planted_rows[:2], [0, 1777, 4095]])
    spot_check(exp, dic, rows, "ncc", 20, s1, i1)


def test_config3_mask_properties(config2):
    """configs[2] match stage: circular signal mask (K = 2819)."""
    from kikuchipy_amd import _lib

    exp, dic, planted_rows, planted_at = config2
    mask = ~ko.circular_window((60, 60)).astype(bool)
    with _lib.Context(0) as ctx:
        s, i = sweep(ctx, exp[:1024], dic, "ncc", 20, chunks=3, signal_mask=mask)
    sel = planted_rows < 1024
    assert np.array_equal(i[planted_rows[sel], 0], planted_at[sel])
    spot_check(exp[:1024], dic, np.array([3, 500, 1023]), "ncc", 20, s, i, signal_mask=mask)


def test_config4_shard_ndp():
    """configs[3], one rank's share: 200x200 map (40 000 patterns) against a
    37 500-pattern shard (300k / 8), ndp, keep_n=20, indices offset like rank 3's."""
    from kikuchipy_amd import _lib

    rng = np.random.default_rng(4)
    exp = rng.integers(0, 256, (40000, 60, 60), dtype=np.uint8)
    dic = rng.random((37500, 60, 60), dtype=np.float32)
    start = 3 * 37500
    planted_rows = rng.choice(40000, 16, replace=False)
    planted_at = rng.choice(37500, 16, replace=False)
    dic[planted_at] = exp[planted_rows].astype(np.float32) * 0.5
    with _lib.Context(0) as ctx:
        ctx.set_problem(60, 60, None, _lib.METRIC_NDP, 20)
        ctx.set_experimental(exp)
        ctx.push_dictionary_chunk(dic, start)
        s, i = ctx.finalize(20)
    assert np.array_equal(i[planted_rows, 0], planted_at + start)
    assert np.allclose(s[planted_rows, 0], 1, atol=ATOL)
    rows = np.array([0, 12345, 39999, planted_rows[0]])
    rs, ri = ko.dictionary_indexing(exp[rows], dic, metric="ndp", keep_n=20, n_per_iteration=12500)
    ko.assert_topk_parity(s[rows], i[rows] - start, rs, ri, atol=ATOL)


def test_config5_large_detector():
    """configs[4] geometry: 120x120 patterns (K = 14 400), f32 MFMA path."""
    from kikuchipy_amd import _lib

    rng = np.random.default_rng(5)
    exp = rng.integers(0, 256, (300, 120, 120), dtype=np.uint8)
    dic = rng.random((5000, 120, 120), dtype=np.float32)
    dic[4321] = exp[7].astype(np.float32) + 3.0
    with _lib.Context(0) as ctx:
        s, i = sweep(ctx, exp, dic, "ncc", 20, chunks=2)
    assert i[7, 0] == 4321 and abs(s[7, 0] - 1) < ATOL
    spot_check(exp, dic, np.array([0, 7, 150, 299]), "ncc", 20, s, i)
