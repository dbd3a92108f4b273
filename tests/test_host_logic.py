"""Host-side logic that must work without a GPU: the metric interface, argument
validation and messages of dictionary_indexing (mirrors the reference's
tests/test_indexing/test_dictionary_indexing.py:90-180 and
tests/test_indexing/test_similarity_metrics.py:28-39), chunk/shard arithmetic."""

import numpy as np
import pytest

import kikuchipy_amd as kpa
from conftest import load_golden
from kikuchipy_amd.indexing._dictionary_indexing import chunk_bounds, info_message
from kikuchipy_amd.parallel import Communicator, shard_range


@pytest.fixture
def dummy():
    g = load_golden("di_dummy.npz")
    return g["dummy"], g["dummy_bg"], g


class FakeDask:
    """Quacks like a dask array for the `isinstance(..., np.ndarray)` checks."""

    def __init__(self, a):
        self._a = np.asarray(a)
        self.shape = self._a.shape

    def all(self):
        return self._a.all()

    def compute(self):
        return self._a


def test_metric_repr_and_dtype():
    """Exact `repr` string and dtype error (test_similarity_metrics.py:28-39)."""
    g = load_golden("di_dummy.npz")
    m = kpa.NormalizedCrossCorrelationMetric()
    assert repr(m) == (
        "NormalizedCrossCorrelationMetric: float32, greater is better, "
        "rechunk: False, navigation mask: False, signal mask: False"
    )
    assert repr(m) == str(g["ncc_all__repr"])
    m2 = kpa.NormalizedDotProductMetric(rechunk=True, signal_mask=np.zeros((3, 3), bool), dtype=np.float64)
    assert repr(m2) == (
        "NormalizedDotProductMetric: float64, greater is better, "
        "rechunk: True, navigation mask: False, signal mask: True"
    )
    assert m.sign == 1 and m.allowed_dtypes == [np.float32, np.float64]
    with pytest.raises(ValueError, match="Data type float16 not among"):
        kpa.NormalizedCrossCorrelationMetric(dtype=np.float16).raise_error_if_invalid()
    assert issubclass(kpa.NormalizedCrossCorrelationMetric, kpa.SimilarityMetric)
    with pytest.raises(TypeError):
        kpa.SimilarityMetric()  # abstract


def test_info_message_matches_reference(dummy):
    data, _, g = dummy
    m = kpa.NormalizedCrossCorrelationMetric()
    msg = info_message(m, 9, 9, "ni", 9)
    assert msg in str(g["ncc_all__msg"])
    m.navigation_mask = g["nav_mask"]
    m2 = kpa.NormalizedCrossCorrelationMetric(navigation_mask=g["nav_mask"])
    assert "Matching 8/9 experimental pattern(s) to 9 dictionary pattern(s)" in info_message(m2, 9, 9, "ni", 8)
    assert info_message(m2, 9, 9, "ni", 8) in str(g["ncc_navmask_k1__msg"])


def test_invalid_metric(dummy):
    data, _, _ = dummy
    with pytest.raises(ValueError, match="'invalid' must be either of "):
        kpa.dictionary_indexing(data, data.reshape(-1, 3, 3), metric="invalid")


def test_invalid_signal_shapes(dummy):
    data, _, _ = dummy
    with pytest.raises(ValueError, match=r"Experimental \(3, 3\) and dictionary \(2, 2\) signal shapes"):
        kpa.dictionary_indexing(data, data[:, :, :2, :2].reshape(-1, 2, 2))


def test_invalid_dictionary(dummy):
    data, _, _ = dummy
    s = kpa.EBSD(data)
    s_dict = kpa.EBSD(data)  # two navigation axes, no xmap
    with pytest.raises(ValueError, match="Dictionary signal must have a non-empty"):
        s.dictionary_indexing(s_dict)
    s_dict.xmap = kpa.DictionaryXmap.empty((3, 3))
    with pytest.raises(ValueError, match="Dictionary signal must have a non-empty"):
        s.dictionary_indexing(s_dict)
    with pytest.raises(ValueError, match="Dictionary signal must have a non-empty"):
        kpa.dictionary_indexing(data, data.reshape(-1, 3, 3), dictionary_rotations=np.zeros((8, 4)))


def test_navigation_mask_raises(dummy):
    data, _, _ = dummy
    dic = data.reshape(-1, 3, 3)
    with pytest.raises(ValueError, match=r"The navigation mask shape \(8,\) and "):
        kpa.dictionary_indexing(data, dic, navigation_mask=np.ones(8, dtype=bool))
    with pytest.raises(ValueError, match=r"The navigation mask must allow for "):
        kpa.dictionary_indexing(data, dic, navigation_mask=np.ones((3, 3), dtype=bool))
    nav = np.ones((3, 3), dtype=bool)
    nav[0, 0] = False
    with pytest.raises(ValueError, match=r"The navigation mask must be a NumPy "):
        kpa.dictionary_indexing(data, dic, navigation_mask=FakeDask(nav))


def test_signal_mask_raises(dummy):
    data, _, g = dummy
    with pytest.raises(ValueError, match="The signal mask must be a NumPy array"):
        kpa.dictionary_indexing(data, data.reshape(-1, 3, 3), signal_mask=FakeDask(g["sig_mask"]))


def test_dtype_rejected_before_any_gpu_work(dummy):
    data, _, _ = dummy
    with pytest.raises(ValueError, match="Data type float16 not among"):
        kpa.dictionary_indexing(data, data.reshape(-1, 3, 3), dtype=np.float16)


def test_static_background_validation(dummy):
    """tests/test_signals/test_ebsd.py:445-466 of the reference."""
    data, bg, _ = dummy
    with pytest.raises(ValueError, match="Static background dtype_out"):
        kpa.remove_static_background(data, np.ones((3, 3), dtype=np.int8))
    with pytest.raises(ValueError, match="Signal"):
        kpa.remove_static_background(data, np.ones((3, 2), dtype=np.uint8))
    with pytest.raises(ValueError, match="`EBSD.static_background` is not a valid array"):
        kpa.EBSD(data).remove_static_background()
    with pytest.raises(ValueError, match="must be either of"):
        kpa.remove_dynamic_background(data, filter_domain="wrong")


def test_chunk_bounds():
    """Same starts/ends as indexing/_dictionary_indexing.py:100-104."""
    assert chunk_bounds(9, 2) == [(0, 2), (2, 4), (4, 6), (6, 8), (8, 9)]
    assert chunk_bounds(9, 9) == [(0, 9)]
    assert chunk_bounds(10, 4) == [(0, 4), (4, 8), (8, 10)]
    assert chunk_bounds(3000, 999) == [(0, 999), (999, 1998), (1998, 2997), (2997, 3000)]
    assert chunk_bounds(5, 100) == [(0, 5)]


@pytest.mark.parametrize("n,world", [(100000, 8), (300000, 8), (10, 3), (7, 8), (1, 1), (0, 2)])
def test_shard_range_partitions(n, world):
    blocks = [shard_range(n, r, world) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(blocks[:-1], blocks[1:]))
    sizes = [b - a for a, b in blocks]
    assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(n, world, world)


def test_communicator_single_rank_needs_no_torch():
    c = Communicator(0, 1)
    assert c.exchange_unique_id(lambda: b"x" * 128) == b"x" * 128
    c.barrier()


def test_filters_window():
    """`~Window("circular", shape).astype(bool)`: the signal mask of the canonical pipeline
    (filters/window.py:163-187, :249-269), against the mask the reference made."""
    from conftest import load_golden

    import kikuchipy_amd as ka
    from oracle import kpdi_oracle as ko

    g = load_golden("di_synth.npz")
    w = ka.filters.Window("circular", (60, 60))
    assert np.array_equal(~w.astype(bool), g["circular_mask"]) and int(w.sum()) == 2819
    assert w.circular and w.name == "circular" and w.origin == (30, 30) and tuple(w.n_neighbours) == (29, 29)
    for shape in [(61, 47), (3, 3), (120, 120), (5, 8)]:
        assert np.array_equal(np.asarray(ka.filters.Window("circular", shape)), ko.circular_window(shape))
    gw = ka.filters.Window("gaussian", (30, 30), std=7.5)
    assert np.allclose(gw, np.outer(ko.gaussian_window_1d(30, 7.5), ko.gaussian_window_1d(30, 7.5)))
    with pytest.raises(NotImplementedError, match="FFT-filter window"):
        ka.filters.Window("modified_hann", (5, 5))


def test_filters_window_like_the_reference_tests():
    """The known answers and error texts of the reference's tests/test_filters/test_window.py:36-222 (restated as data)."""
    from scipy.signal.windows import gaussian, general_gaussian

    from kikuchipy_amd.filters import Window

    circular33 = np.array([0, 1, 0, 1, 1, 1, 0, 1, 0]).reshape(3, 3)
    circular54 = np.array([0, 0, 1, 0, 0, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 0, 0, 1, 0]).reshape(5, 4)
    gauss33_circular = np.array([0, 0.60653066, 0, 0.60653066, 1, 0.60653066, 0, 0.60653066, 0]).reshape(3, 3)
    custom = np.arange(25).reshape(5, 5)
    w = Window()  # the defaults: "circular", (3, 3)
    assert w.is_valid and w.name == "circular" and w.circular is True and np.array_equal(w, circular33)
    assert w.__array_finalize__(None) is None
    w = Window(window=custom, shape=(10, 20))
    assert w.name == "custom" and w.shape == (5, 5) and w.circular is False and np.array_equal(w, custom)
    w = Window(window="gaussian", shape=(5, 5), kwargs=2)  # (how the reference's test passes std: by position in **kwargs)
    assert w.name == "gaussian" and np.allclose(w, np.outer(gaussian(5, 2), gaussian(5, 2)))
    w = Window(window="general_gaussian", shape=(5, 5), p=0.5, std=2)
    assert w.is_valid and w.name == "general_gaussian"
    assert np.allclose(w, np.outer(general_gaussian(5, 0.5, 2), general_gaussian(5, 0.5, 2)))
    for nx in (3, 5, 7, 8):
        assert Window(Nx=nx).shape == (nx,)
    for window, shape, err, match in [
        ([[0, 1, 0], [1, 1, 1], [0, 1, 0]], (5, 5), ValueError, "Window <class 'list'> must be of type numpy.ndarray,"),
        ("boxcar", (5, -5), ValueError, "All window axes .* must be > 0"),
        ("boxcar", (5, 5.1), TypeError, "Window shape .* must be a sequence of ints."),
    ]:
        with pytest.raises(err, match=match):
            Window(window=window, shape=shape)
    a = np.arange(5)
    w = Window(a)
    assert isinstance(w, Window) and w.name == "custom" and w.circular is False and w.sum() == a.sum()
    assert isinstance(w[1:], Window) and w[1:].name == "custom" and isinstance(a.view(Window), Window)
    for window, shape, coeff, circ, name in [
        ("rectangular", (3, 3), circular33, True, "circular"), ("boxcar", (3, 3), circular33, True, "circular"),
        ("rectangular", (3,), np.ones(3), False, "rectangular"), ("gaussian", (3, 3), gauss33_circular, True, "gaussian"),
        ("rectangular", (5, 4), circular54, True, "circular"),
    ]:
        k = Window(window=window, shape=shape, **({"std": 1} if window == "gaussian" else {}))
        k.make_circular()
        assert np.allclose(k, coeff) and k.name == name and k.circular is circ
    for shape, ok in [((3,), True), ((3, 3), True), ((3, 4), False), ((4, 3), False), ((4, 4), False)]:
        assert Window(shape=shape).shape_compatible((3, 3)) == ok  # (the dummy signal's navigation shape)
    # is_valid: a non-string name, a third axis, a non-bool `circular`
    w = Window()
    w._name = 1
    assert not w.is_valid
    assert not np.expand_dims(Window(), 1).is_valid
    w = Window()
    w._circular = "True"
    assert not w.is_valid
    assert repr(Window()).startswith("Window (3, 3) circular\n")


# ---------------------------------------------------------------------------------------------
# a10: the hand-over to orix' CrystalMap (indexing/_dictionary_indexing.py:141-167 of the reference)
class _Recorder:
    """Stands in for orix' CrystalMap / Rotation (orix is not installed): records its arguments,
    like the stub the golden generator runs the reference with (oracle/ref_shim.py:_Recorder)."""

    def __init__(self, *args, **kwargs):
        self.args, self.kw = args, kwargs


@pytest.fixture
def orix_stub(monkeypatch):
    import sys
    import types

    calls = []

    def create_coordinate_arrays(shape, step_sizes=None):
        calls.append((tuple(shape), tuple(step_sizes)))
        return {"x": np.arange(int(np.prod(shape))), "coords_of": tuple(shape)}, int(np.prod(shape))

    orix, cm, qu = types.ModuleType("orix"), types.ModuleType("orix.crystal_map"), types.ModuleType("orix.quaternion")
    cm.CrystalMap, cm.create_coordinate_arrays, qu.Rotation = _Recorder, create_coordinate_arrays, _Recorder
    orix.crystal_map, orix.quaternion = cm, qu
    for name, mod in (("orix", orix), ("orix.crystal_map", cm), ("orix.quaternion", qu)):
        monkeypatch.setitem(sys.modules, name, mod)
    return calls


@pytest.mark.parametrize("keep_n,masked", [(3, False), (3, True), (1, True), (1, False)])
def test_to_crystal_map_hands_over_what_the_reference_does(orix_stub, keep_n, masked):
    from kikuchipy_amd.indexing._dictionary_indexing import DictionaryIndexingResult

    rng = np.random.default_rng(1)
    nav_shape, n_dict = (3, 4), 50
    n_all = 12
    in_data = np.ones(n_all, dtype=bool)
    if masked:
        in_data[[1, 5, 6]] = False
    m = int(in_data.sum())
    idx = rng.integers(0, n_dict, (m, keep_n))
    scores = np.sort(rng.random((m, keep_n)).astype(np.float32), axis=1)[:, ::-1]
    dict_rot = rng.standard_normal((n_dict, 4))
    # what dictionary_indexing() assembles (reference :142-166)
    if masked:
        s_all = np.zeros((n_all, keep_n), np.float32)
        i_all = np.zeros((n_all, keep_n), np.int64)
        rot = np.zeros((n_all, keep_n, 4))
        rot[..., 0] = 1
        s_all[in_data], i_all[in_data], rot[in_data] = scores, idx, dict_rot[idx]
        if keep_n == 1:
            s_all, i_all, rot = s_all.squeeze(), i_all.squeeze(), rot.reshape(-1, 4)
    else:
        s_all, i_all, rot = scores, idx, dict_rot[idx]
    res = DictionaryIndexingResult(s_all, i_all, nav_shape, (1.5, 2.0), in_data, keep_n, rotations=rot,
                                   phase_name="ni", scan_unit="um")
    phases = object()
    xmap = res.to_crystal_map(phase_list=phases)
    assert orix_stub == [((3, 4), (1.5, 2.0))]                    # create_coordinate_arrays(nav_shape, step_sizes)
    kw = xmap.kw
    assert kw["phase_list"] is phases and kw["coords_of"] == (3, 4)  # CrystalMap(phase_list=..., **xmap_kw)
    assert set(kw["prop"]) == {"scores", "simulation_indices"}
    assert kw["prop"]["scores"] is s_all and kw["prop"]["simulation_indices"] is i_all
    r = kw["rotations"].args[0]                                   # Rotation(quaternions)
    if masked:
        assert np.array_equal(kw["is_in_data"], in_data)
        flat = r.reshape(n_all, keep_n, 4)
        assert np.array_equal(flat[in_data], dict_rot[idx])       # rot[nav_mask] = rotations[simulation_indices]
        assert np.all(flat[~in_data] == [1, 0, 0, 0])             # Rotation.identity elsewhere
        assert r.shape == ((n_all, 4) if keep_n == 1 else (n_all, keep_n, 4))
        assert kw["prop"]["scores"].shape == ((n_all,) if keep_n == 1 else (n_all, keep_n))
    else:
        assert "is_in_data" not in kw
        assert np.array_equal(r, dict_rot[idx])
    assert xmap.scan_unit == "um"


def test_to_crystal_map_needs_rotations(orix_stub):
    from kikuchipy_amd.indexing._dictionary_indexing import DictionaryIndexingResult

    res = DictionaryIndexingResult(np.zeros((2, 1)), np.zeros((2, 1), int), (2,), (1,), np.ones(2, bool), 1)
    with pytest.raises(ValueError, match="dictionary_rotations"):
        res.to_crystal_map()


def test_get_map_data_of_the_result_holders():
    """`xmap.get_map_data("scores")` as the reference's tutorial calls it on the maps this path returns (orix's
    CrystalMap.get_map_data: the property laid out on the map, NaN outside the navigation mask)."""
    from kikuchipy_amd.indexing._dictionary_indexing import DictionaryIndexingResult

    in_data = np.array([1, 0, 1, 1, 1, 1], dtype=bool)
    scores = np.arange(10, dtype=np.float32).reshape(5, 2)
    res = DictionaryIndexingResult(scores, np.arange(10).reshape(5, 2), (2, 3), (1, 1), in_data, 2)
    m = res.get_map_data("scores")
    assert m.shape == (2, 3, 2) and np.isnan(m[0, 1]).all() and np.array_equal(m.reshape(6, 2)[in_data], scores)
    i = res.get_map_data("simulation_indices")
    assert i.dtype.kind == "i" and (i[0, 1] == 0).all() and i[1, 2, 1] == 9
    assert np.array_equal(res.get_map_data(np.array([0.26, 1, 2, 3, 4]), decimals=1)[0], [0.3, np.nan, 1.0], equal_nan=True)
    with pytest.raises(ValueError, match="not among the properties"):
        res.get_map_data("nothing")


def test_signal_detector_attribute():
    """signals/ebsd.py:188-223: a signal always has a detector (by default one of its shape), a detector that is set is
    checked against the signal with the reference's texts (signals/util/_detector.py:28-59), copies carry it."""
    s = kpa.EBSD(np.zeros((2, 3, 6, 8), np.uint8))
    assert s.detector.shape == (6, 8) and s.detector.navigation_shape == (1,)
    with pytest.raises(ValueError, match=r"Detector shape \(6, 6\) must be equal to the signal shape \(6, 8\)"):
        s.detector = kpa.EBSDDetector(shape=(6, 6))
    with pytest.raises(ValueError, match="Detector must have exactly one projection center"):
        s.detector = kpa.EBSDDetector(shape=(6, 8), pc=np.ones((4, 3)))
    s.detector = kpa.EBSDDetector(shape=(6, 8), pc=np.full((2, 3, 3), 0.5))
    t = s.deepcopy()
    t.detector.pcx = 0.1
    assert t.detector.navigation_shape == (2, 3) and s.detector.pcx[0, 0] == 0.5
    u = kpa.EBSD(np.zeros((6, 8)), detector=kpa.EBSDDetector(shape=(6, 8), pc=(0.4, 0.2, 0.6)))
    assert u.detector.pcz == 0.6


def test_signal_xmap_and_static_background_setters():
    """signals/ebsd.py:236-266: a crystal map of another shape is refused (signals/util/_crystal_map.py:55-59), a
    background of another data type or shape is taken with a warning."""
    s = kpa.EBSD(np.zeros((2, 3, 6, 8), np.uint8), static_background=np.zeros((6, 8), np.uint8))
    with pytest.warns(UserWarning, match="Background pattern has different data type from patterns"):
        s.static_background = np.zeros((6, 8), np.float32)
    with pytest.warns(UserWarning, match="Background pattern has different shape from patterns"):
        s.static_background = np.zeros((6, 6), np.uint8)
    with pytest.raises(ValueError, match=r"Crystal map shape \(5,\) and signal's navigation shape \(2, 3\) must be the same"):
        s.xmap = kpa.DictionaryXmap(np.zeros((5, 4)))
    s.xmap = kpa.DictionaryXmap.empty((2, 3))
    assert s.xmap.shape == (2, 3)
