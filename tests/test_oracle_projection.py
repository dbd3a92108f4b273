"""Pin the oracle's master-pattern projection to the
reference: tests/golden/projection.npz holds what the reference's own
`signals/util/_master_pattern.py` functions returned (oracle/gen_golden.py
`gen_projection`, build container only) for the Ni master pattern the
reference ships for its tests."""

import numpy as np
import pytest

from conftest import load_golden
from oracle import kpdi_oracle as ko

DETECTORS = {
    "det60": dict(shape=(60, 60), pc=(0.4210, 0.7794, 0.5049), sample_tilt=70.0),
    "det48x60": dict(shape=(48, 60), pc=(0.52, 0.71, 0.63), sample_tilt=69.5, tilt=5.0, azimuthal=3.0,
                     twist=1.5),
}


@pytest.fixture(scope="module")
def g():
    return load_golden("projection.npz")


def test_vector2lambert_known_answers():
    """tests/test_signals/test_ebsd_master_pattern.py:746-775 of the reference."""
    xyz = np.array([[0, 0, 1], [0, 1, 0], [2, 0, 0], [0, 0, -3], [0, 0, -1], [0, -1, 0], [-2, 0, 0],
                    [0, 0, 3]], dtype=np.float64)
    h = np.sqrt(np.pi / 2)
    want = [[0, 0], [0, h], [h, 0], [0, 0], [0, 0], [0, -h], [-h, 0], [0, 0]]
    assert np.allclose(ko.vector2lambert(xyz), want)


def test_lambert_golden(g):
    assert np.allclose(ko.vector2lambert(g["vec"]), g["vec__lambert"], rtol=0, atol=1e-14)
    got = ko.lambert_interpolation_parameters(g["vec"], 401, 401, 200.0)
    for name, val in zip(("nii", "nij", "niip", "nijp"), got[:4]):
        assert np.array_equal(val, g[f"vec__{name}"]), name
        assert val.dtype == np.int32
    for name, val in zip(("di", "dj", "dim", "djm"), got[4:]):
        assert np.allclose(val, g[f"vec__{name}"], rtol=0, atol=1e-11), name


def test_interpolation_parameter_ranges(g):
    """The property the reference asserts (test_ebsd_master_pattern.py:463-481)."""
    dc = g["det60__dc"]
    nii, nij, niip, nijp = ko.lambert_interpolation_parameters(dc, 101, 101, 50.0)[:4]
    for a in (nii, nij, niip, nijp):
        assert a.min() >= 0 and a.max() < 101
    assert np.all(nii <= niip) and np.all(nij <= nijp)


@pytest.mark.parametrize("name", sorted(DETECTORS))
def test_detector_geometry(g, name):
    d = DETECTORS[name]
    m = ko.sample_to_detector_matrix(d["sample_tilt"], d.get("tilt", 0), d.get("azimuthal", 0),
                                     d.get("twist", 0))
    assert np.allclose(m, g[f"{name}__s2d"], rtol=0, atol=1e-15)
    assert np.allclose(m @ m.T, np.eye(3), atol=1e-14)
    dc = ko.detector_direction_cosines(**d)
    assert dc.shape == (d["shape"][0] * d["shape"][1], 3)
    assert np.allclose(dc, g[f"{name}__dc"], rtol=0, atol=1e-14)
    assert np.allclose(np.sum(dc**2, axis=1), 1)


def test_direction_cosines_signal_mask(g):
    keep = ko.circular_window((60, 60)).astype(bool)
    dc = ko.detector_direction_cosines(signal_mask=keep, **DETECTORS["det60"])
    assert np.allclose(dc, g["det60__dc_circ"], rtol=0, atol=1e-14)
    assert np.allclose(dc, g["det60__dc"][keep.ravel()], rtol=0, atol=1e-14)


def test_rotate_vector_is_a_rotation(g):
    dc = g["det60__dc"]
    assert np.array_equal(ko.rotate_vector([1, 0, 0, 0], dc), dc)
    for q in g["rot8"]:
        v = ko.rotate_vector(q, dc)
        assert np.allclose(np.sum(v**2, axis=1), 1)
        a, b, c, d = q
        back = ko.rotate_vector([a, -b, -c, -d], v)
        assert np.allclose(back, dc, atol=1e-14)


FLOAT_CASES = {
    # name: (master pattern dtype, lower hemisphere, rescale, out range)
    "u8mp_f32": (np.uint8, "lower", True, (-1, 1)),
    "f32mp_f32": (np.float32, "lower", False, (1, 2)),
    "hemis_f32": (np.float32, "inverted", False, (1, 2)),
}


def master_arrays(g, dtype, lower):
    up = g["mp_upper"].astype(dtype)
    lo = g["mp_lower"].astype(dtype) if lower == "lower" else (255 - g["mp_upper"]).astype(dtype)
    return up, lo


@pytest.mark.parametrize("name", sorted(FLOAT_CASES))
def test_project_patterns_float(g, name):
    dtype, lower, rescale, (omin, omax) = FLOAT_CASES[name]
    up, lo = master_arrays(g, dtype, lower)
    got = ko.project_patterns(g["rot8"], g["det60__dc"], up, lo, rescale, omin, omax, np.float32)
    want = g[f"{name}__patterns"]
    assert got.dtype == np.float32 and got.shape == want.shape
    # f64 arithmetic, then one rounding to f32: a few ulp of float32 at most
    assert np.allclose(got, want, rtol=3e-7, atol=3e-7 * np.abs(want).max())
    if rescale:
        assert np.all(got.min(axis=1) == omin) and np.allclose(got.max(axis=1), omax, atol=1e-6)


def test_project_patterns_other_detector(g):
    up, lo = master_arrays(g, np.float32, "lower")
    got = ko.project_patterns(g["rot8"][:4], g["det48x60__dc"], up, lo)
    assert np.allclose(got, g["det48x60_f32__patterns"], rtol=3e-7, atol=1e-4)


@pytest.mark.parametrize("name,mp_dtype,rescale", [("f32mp_u8", np.float32, True), ("u8mp_u8", np.uint8, False)])
def test_project_patterns_uint8(g, name, mp_dtype, rescale):
    """Integer output truncates: an interpolated value within rounding of an
    integer may land on either side (the reference's own test allows 254 or
    255 for the maximum, test_ebsd_master_pattern.py:459-461)."""
    up, lo = master_arrays(g, mp_dtype, "lower")
    got = ko.project_patterns(g["rot8"], g["det60__dc"], up, lo, rescale, 0, 255, np.uint8)
    want = g[f"{name}__patterns"]
    assert got.dtype == np.uint8
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1 and np.mean(diff != 0) < 1e-3


def test_project_patterns_one_pc_per_pattern(g):
    om = g["det60__s2d"].T
    pcs = g["varpc__pcs"]
    for i, pc in enumerate(pcs):
        dc = ko.direction_cosines_fixed_pc(ko.gnomonic_bounds((60, 60), pc), pc[2], 60, 60, om)
        assert np.allclose(dc[::97], g["varpc__dc_sample"][i], rtol=0, atol=1e-14)
    up, lo = master_arrays(g, np.float32, "lower")
    got = ko.project_patterns_varying_pc(g["rot8"][:4], pcs, (60, 60), om, up, lo)
    assert np.allclose(got, g["varpc_f32__patterns"], rtol=3e-7, atol=1e-4)
    got = ko.project_patterns_varying_pc(g["rot8"][:4], pcs, (60, 60), om, g["mp_upper"], g["mp_lower"], True, -1, 1)
    assert np.allclose(got, g["varpc_u8mp_f32__patterns"], rtol=3e-7, atol=3e-7)
    # the last PC is the fixed one of the other fixtures
    assert np.allclose(got[3], g["u8mp_f32__patterns"][3], rtol=3e-7, atol=3e-7)


def test_hemisphere_selection_matters(g):
    a, b = g["f32mp_f32__patterns"], g["hemis_f32__patterns"]
    assert np.any(a != b)  # some pixels of the 8 patterns look into the lower hemisphere


def test_dictionary_end_to_end(g):
    """Rotations -> projected dictionary -> DI, against the reference's chain."""
    keep = ko.circular_window((60, 60)).astype(bool)
    dic = ko.project_patterns(g["di_rot"], g["det60__dc"], g["mp_upper"], g["mp_lower"], True, -1, 1,
                              np.float32).reshape(-1, 60, 60)
    assert np.allclose(dic[::100], g["di_dic_sample"], rtol=3e-7, atol=3e-7)
    s, i = ko.dictionary_indexing(g["di_exp"], dic, metric="ncc", keep_n=10, n_per_iteration=500)
    ko.assert_topk_parity(s, i, g["di_ncc_k10__scores"], g["di_ncc_k10__indices"], atol=1e-5)
    assert np.array_equal(i[:, 0], g["di_picks"])
    s, i = ko.dictionary_indexing(g["di_exp"], dic, metric="ndp", keep_n=10, signal_mask=~keep)
    ko.assert_topk_parity(s, i, g["di_ndp_k10_circ__scores"], g["di_ndp_k10_circ__indices"], atol=1e-5)
