"""`merge_crystal_maps` (host NumPy mirror) against what the reference's
function returned for the same inputs (tests/golden/consumers.npz; the
reference ran with stand-in CrystalMaps, oracle/ref_shim.py), and the argument
checks of `orientation_similarity_map`."""

import numpy as np
import pytest

from conftest import load_golden

import kikuchipy_amd as ka


class Xmap:
    def __init__(self, g, tag, j, shape, name):
        self.prop = {"scores": g[f"{tag}__in{j}_scores"], "simulation_indices": g[f"{tag}__in{j}_simulation_indices"]}
        self.rotations = g[f"{tag}__in{j}_rotations"]
        self.phase_id = g[f"{tag}__in{j}_phase_id"]
        self.shape = shape
        self.phase_name = name
        key = f"{tag}__in{j}_mask"
        self.is_in_data = ~g[key].ravel() if key in g else np.ones(int(np.prod(shape)), dtype=bool)


@pytest.fixture(scope="module")
def g():
    return load_golden("consumers.npz")


def check(res, g, tag, with_indices=True):
    assert np.array_equal(res.phase_id, g[f"{tag}__phase_id"])
    assert list(res.phase_names) == [n for n in g[f"{tag}__phase_names"] if n != "not_indexed"]
    assert np.array_equal(res.scores, g[f"{tag}__scores"]) and res.scores.dtype == g[f"{tag}__scores"].dtype
    assert np.array_equal(res.merged_scores, g[f"{tag}__merged_scores"], equal_nan=True)
    assert np.array_equal(res.rotations, g[f"{tag}__rotations"])
    if with_indices:
        assert np.array_equal(res.simulation_indices, g[f"{tag}__simulation_indices"])
        assert res.simulation_indices.dtype == np.int32
        assert np.array_equal(res.merged_simulation_indices, g[f"{tag}__merged_simulation_indices"], equal_nan=True)
        assert set(res.prop) == {"scores", "merged_scores", "simulation_indices", "merged_simulation_indices"}
    else:
        assert res.simulation_indices is None and set(res.prop) == {"scores", "merged_scores"}


@pytest.mark.parametrize("tag,kw", [
    ("merge2", dict()),
    ("merge2_mean3", dict(mean_n_best=3)),
    ("merge2_lower", dict(greater_is_better=False)),
    ("merge2_negmean", dict(mean_n_best=-2)),
])
def test_merge_two_maps(g, tag, kw):
    maps = [Xmap(g, tag, 0, (4, 3), "a"), Xmap(g, tag, 1, (4, 3), "b")]
    res = ka.merge_crystal_maps(maps, simulation_indices_prop="simulation_indices", **kw)
    check(res, g, tag)
    assert res.shape == (4, 3) and res.size == 12 and res.rotations_per_point == 5
    # every phase won somewhere in these fixtures
    assert set(res.phase_id) == {0, 1}


def test_merge_without_indices(g):
    maps = [Xmap(g, "merge2_no_indices", 0, (4, 3), "a"), Xmap(g, "merge2_no_indices", 1, (4, 3), "b")]
    check(ka.merge_crystal_maps(maps), g, "merge2_no_indices", with_indices=False)


def test_merge_with_navigation_masks_and_not_indexed(g):
    tag = "merge3_masks"
    maps = [Xmap(g, tag, j, (4, 3), n) for j, n in enumerate("abc")]
    masks = [g[f"{tag}__in0_mask"], g[f"{tag}__in1_mask"], None]
    res = ka.merge_crystal_maps(maps, simulation_indices_prop="simulation_indices", navigation_masks=masks)
    check(res, g, tag)
    # point (0, 0) is masked out of two maps and marked not indexed in the third: it is NOT -1,
    # because a point has to be not indexed in every map (and see the note in merge_crystal_maps)
    assert res.phase_id[0] == 2 and -1 not in res.phase_id
    assert np.isnan(res.merged_scores).any()
    # masks are derived from `is_in_data` when not given
    res2 = ka.merge_crystal_maps(maps, simulation_indices_prop="simulation_indices")
    check(res2, g, tag)


def test_merge_not_indexed_points(g):
    tag = "merge2_not_indexed"
    maps = [Xmap(g, tag, 0, (4, 3), "a"), Xmap(g, tag, 1, (4, 3), "b")]
    res = ka.merge_crystal_maps(maps, simulation_indices_prop="simulation_indices")
    check(res, g, tag)
    assert res.phase_id[3] == -1 and np.count_nonzero(res.phase_id == -1) == 1
    assert "not_indexed" in list(g[f"{tag}__phase_names"])


def test_merge_maps_of_the_same_phase(g):
    tag = "merge3_same_name"
    maps = [Xmap(g, tag, j, (4, 3), n) for j, n in enumerate("aab")]
    res = ka.merge_crystal_maps(maps, simulation_indices_prop="simulation_indices")
    check(res, g, tag)
    assert list(res.phase_names) == ["a", "b"]


def test_merge_full_size_results_with_masked_rows(g):
    """This package's own results carry full-size arrays (zero rows where masked)."""
    tag = "merge3_masks"
    maps = [Xmap(g, tag, j, (4, 3), n) for j, n in enumerate("abc")]
    for m in maps:
        isin = m.is_in_data
        for key in ("scores", "simulation_indices"):
            full = np.zeros((12,) + m.prop[key].shape[1:], dtype=m.prop[key].dtype)
            full[isin] = m.prop[key]
            m.prop[key] = full
        full = np.zeros((12,) + m.rotations.shape[1:])
        full[isin] = m.rotations
        m.rotations = full
    # phase_id of map 0 refers to its in-data points
    res = ka.merge_crystal_maps(maps, simulation_indices_prop="simulation_indices")
    check(res, g, tag)


def test_merge_errors(g):
    a, b = Xmap(g, "merge2", 0, (4, 3), "a"), Xmap(g, "merge2", 1, (4, 3), "b")
    with pytest.raises(ValueError, match="Number of crystal maps and navigation masks must be equal"):
        ka.merge_crystal_maps([a, b], navigation_masks=[None])
    with pytest.raises(ValueError, match="does not have as many 'False'"):
        ka.merge_crystal_maps([a, b], navigation_masks=[np.ones((4, 3), bool), None])
    with pytest.raises(ValueError, match="must be a NumPy array or 'None'"):
        ka.merge_crystal_maps([a, b], navigation_masks=[[0], None])
    c = Xmap(g, "merge2", 1, (3, 4), "b")
    with pytest.raises(ValueError, match="must have the same navigation shape"):
        ka.merge_crystal_maps([a, c])
    d = Xmap(g, "merge3_same_name", 0, (4, 3), "b")
    with pytest.raises(ValueError, match="same number of rotations and scores per point"):
        ka.merge_crystal_maps([a, d])
    e = Xmap(g, "merge2", 0, (4, 3), "a")
    e.prop = {"scores": e.prop["scores"][:, :3], "simulation_indices": e.prop["simulation_indices"]}
    f = Xmap(g, "merge2", 1, (4, 3), "b")
    f.prop = {"scores": f.prop["scores"][:, :3], "simulation_indices": f.prop["simulation_indices"]}
    with pytest.raises(ValueError, match="more simulation indices than scores"):
        ka.merge_crystal_maps([e, f], simulation_indices_prop="simulation_indices")


def test_osm_argument_checks():
    idx = np.ones((100, 5), dtype=np.int64)
    with pytest.raises(ValueError, match="n_best 6 cannot be greater than keep_n 5"):
        ka.orientation_similarity_map(idx, n_best=6, shape=(10, 10))
    with pytest.raises(ValueError, match="`shape` is needed"):
        ka.orientation_similarity_map(idx)
    with pytest.raises(ValueError, match="does not hold 100 points"):
        ka.orientation_similarity_map(idx, shape=(9, 10))
    from kikuchipy_amd.indexing._orientation_similarity_map import _footprint_offsets

    assert _footprint_offsets(np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]])).tolist() == [[-1, 0], [0, -1], [0, 0], [0, 1], [1, 0]]
    assert _footprint_offsets(np.ones((1, 5))).tolist() == [[0, -2], [0, -1], [0, 0], [0, 1], [0, 2]]
