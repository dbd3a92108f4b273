"""GPU parity of the orientation similarity map through the C
ABI against the reference's outputs (tests/golden/consumers.npz) and the
oracle: integer work, bit-exact."""

import numpy as np
import pytest

from conftest import load_golden
from oracle import kpdi_oracle as ko

pytestmark = pytest.mark.gpu

OSM_CASES = {
    "osm_default": dict(),
    "osm_normalized": dict(normalize=True),
    "osm_nbest7": dict(n_best=7),
    "osm_from5_to8": dict(n_best=8, from_n_best=5, normalize=True),
    "osm_square_fp": dict(footprint=np.ones((3, 3), dtype=int), center_index=4),
    "osm_row_fp": dict(n_best=10, footprint=np.array([[1, 1, 1, 1, 1]]), center_index=2),
}


@pytest.fixture(scope="module")
def g():
    return load_golden("consumers.npz")


@pytest.mark.parametrize("name", sorted(OSM_CASES))
def test_osm_golden(g, name):
    import kikuchipy_amd as ka

    got = ka.orientation_similarity_map(g["idx_9x13_k20"], shape=(9, 13), **OSM_CASES[name])
    assert got.dtype == np.float32 and np.array_equal(got, g[name])


def test_osm_duplicates_and_reference_cases(g):
    import kikuchipy_amd as ka

    assert np.array_equal(ka.orientation_similarity_map(g["idx_dup"], n_best=12, shape=(9, 13)), g["osm_dup"])
    idx = np.tile(np.arange(5), (100, 1))
    assert np.array_equal(ka.orientation_similarity_map(idx, shape=(10, 10)), np.full((10, 10), 5, np.float32))
    assert np.array_equal(ka.orientation_similarity_map(idx, shape=(10, 10), normalize=True), np.ones((10, 10), np.float32))
    osm = ka.orientation_similarity_map(np.ones((100, 5)), shape=(10, 10), from_n_best=2)
    assert osm.shape == (10, 10, 4)


def test_osm_large_map_and_long_lists_vs_oracle():
    """keep_n > 64 (several list elements per lane), awkward footprint, 1-wide map."""
    import kikuchipy_amd as ka

    rng = np.random.default_rng(5)
    idx = rng.integers(0, 300, (23 * 17, 70))
    fp = np.array([[1, 0, 1], [0, 1, 0], [1, 1, 0]])
    got = ka.orientation_similarity_map(idx, shape=(23, 17), n_best=70, from_n_best=68, footprint=fp, center_index=2)
    want = ko.orientation_similarity_map(idx, (23, 17), n_best=70, from_n_best=68, footprint=fp, center_index=2)
    assert np.array_equal(got, want)
    col = rng.integers(0, 40, (31, 9))
    assert np.array_equal(ka.orientation_similarity_map(col, shape=(31, 1)), ko.orientation_similarity_map(col, (31, 1)))
    one = ka.orientation_similarity_map(col[:1], shape=(1, 1))
    assert one.shape == () and np.isnan(one)  # no neighbour: nanmean of nothing


def test_osm_from_resident_result(synth_inputs):
    """A map straight from the best-k lists left in HBM by the sweep."""
    import kikuchipy_amd as ka
    from kikuchipy_amd import _lib

    exp, dic, g = synth_inputs
    with _lib.Context(0) as ctx:
        ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 20)
        ctx.set_experimental(exp)
        ctx.push_dictionary_chunk(dic, 0)
        scores, idx = ctx.finalize(20)
        got = ka.orientation_similarity_map(idx, shape=(6, 8), n_best=10, context=ctx)
        assert np.array_equal(got, ko.orientation_similarity_map(idx, (6, 8), n_best=10))
        assert ctx.holds_result(idx)
        # anything that is NOT what the last finalize() returned is uploaded, never read from the
        # resident lists (another run's result, a merged or refined map of the same shape ...)
        other = idx[:, ::-1].copy()
        assert not ctx.holds_result(other)
        # ... including the returned array itself once the caller has edited it IN PLACE (the context keeps no
        # reference to it: it compares with what kpdi_finalize left in its own staging buffer)
        keep = idx.copy()
        idx[3, 0] = idx[3, 1]
        assert not ctx.holds_result(idx)
        got = ka.orientation_similarity_map(idx, shape=(6, 8), n_best=10, context=ctx)
        assert np.array_equal(got, ko.orientation_similarity_map(idx, (6, 8), n_best=10))
        idx[:] = keep
        assert ctx.holds_result(idx)
        got = ka.orientation_similarity_map(other, shape=(6, 8), context=ctx)
        assert np.array_equal(got, ko.orientation_similarity_map(other, (6, 8)))
        got = ka.orientation_similarity_map(idx[:40], shape=(5, 8), context=ctx)
        assert np.array_equal(got, ko.orientation_similarity_map(idx[:40], (5, 8)))
        # the raw ABI still refuses a resident map of the wrong shape / after a reset
        with pytest.raises(_lib.KpdiError, match="resident result is 48 x 20"):
            ctx.orientation_similarity_map(None, (5, 8), 20, 20, 20, [[0, 0]], 0, False)
        ctx.reset_topk()
        with pytest.raises(_lib.KpdiError, match="no resident result"):
            ctx.orientation_similarity_map(None, (6, 8), 20, 20, 20, [[0, 0]], 0, False)
        got = ka.orientation_similarity_map(idx, shape=(6, 8), n_best=10, context=ctx)
        assert np.array_equal(got, ko.orientation_similarity_map(idx, (6, 8), n_best=10))


def test_osm_on_indexing_result(synth_inputs):
    import kikuchipy_amd as ka

    exp, dic, g = synth_inputs
    res = ka.dictionary_indexing(exp.reshape(6, 8, 60, 60), dic, keep_n=8, verbose=False)
    got = ka.orientation_similarity_map(res, normalize=True)
    assert got.shape == (6, 8)
    assert np.array_equal(got, ko.orientation_similarity_map(res.simulation_indices, (6, 8), normalize=True))
