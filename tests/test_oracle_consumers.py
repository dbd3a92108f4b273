"""Pin the oracle's restatement of the indexing-result consumers (SURVEY.md
8(f3)) to the reference: tests/golden/consumers.npz holds what the reference's
`orientation_similarity_map` returned (oracle/gen_golden.py `gen_consumers`)."""

import numpy as np
import pytest

from conftest import load_golden
from oracle import kpdi_oracle as ko

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
    got = ko.orientation_similarity_map(g["idx_9x13_k20"], (9, 13), **OSM_CASES[name])
    assert got.dtype == np.float32 and got.shape == g[name].shape
    assert np.array_equal(got, g[name])
    assert g[name].std() > 0  # the fixture is not trivial


def test_osm_duplicates_use_set_semantics(g):
    got = ko.orientation_similarity_map(g["idx_dup"], (9, 13), n_best=12)
    assert np.array_equal(got, g["osm_dup"])


def test_osm_reference_test_cases(g):
    """tests/test_indexing/test_orientation_similarity_map.py:27-64 of the reference."""
    idx = np.tile(np.arange(5), (100, 1))
    assert np.allclose(ko.orientation_similarity_map(idx, (10, 10)), np.full((10, 10), 5))
    assert np.array_equal(ko.orientation_similarity_map(idx, (10, 10)), g["osm_reftest_tile"])
    assert np.allclose(ko.orientation_similarity_map(idx, (10, 10), normalize=True), np.ones((10, 10)))
    with pytest.raises(ValueError, match="n_best 6 cannot be greater than"):
        ko.orientation_similarity_map(np.ones((100, 5)), (10, 10), n_best=6)
    osm = ko.orientation_similarity_map(np.ones((100, 5)), (10, 10), from_n_best=2)
    assert osm.shape == (10, 10, 4) == tuple(g["osm_reftest_from2_shape"])
