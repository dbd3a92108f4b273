"""The h5ebsd reader of libkpdi (host code, no GPU needed)
against a kikuchipy-h5ebsd file written with h5py from the Ni patterns the
reference ships (tests/golden/h5ebsd_ni.h5, oracle/gen_golden.py `gen_h5ebsd`)."""

import os
import warnings

import numpy as np
import pytest

from conftest import GOLDEN, load_golden

import kikuchipy_amd as ka
from kikuchipy_amd import _lib

PATH = os.path.join(GOLDEN, "h5ebsd_ni.h5")


@pytest.fixture(scope="module")
def want():
    try:
        _lib.h5ebsd_info(PATH)
    except _lib.KpdiError as e:
        if "HDF5 C library could not be loaded" in str(e):
            pytest.skip("no libhdf5 on this machine")
        raise
    return load_golden("h5ebsd_expected.npz")


def test_info(want):
    info = _lib.h5ebsd_info(PATH)  # first scan
    assert info.scan == b"Scan 1" and (info.ny, info.nx, info.sy, info.sx) == (3, 3, 60, 60)
    assert info.dtype == _lib.DTYPE_CODES[np.dtype(np.uint8)] and info.has_static_background == 1
    assert info.n_pc == 9 and info.n_stored == 9 * 3600 and info.binning == 8
    assert (info.step_y, info.step_x, info.detector_pixel_size) == (1.5, 1.5, 70.0)
    assert (info.sample_tilt, info.azimuth_angle, info.elevation_angle) == (70.0, 0.0, 1.5)
    info = _lib.h5ebsd_info(PATH, "Scan 3")
    assert (info.ny, info.nx) == (1, 9) and info.dtype == _lib.DTYPE_CODES[np.dtype(np.float32)]
    assert info.has_static_background == 0 and info.n_pc == 0 and info.n_stored == 7 * 3600


def test_load_scan1_equals_the_reference_data(want):
    s = ka.load(PATH)
    assert s.data.dtype == np.uint8 and np.array_equal(s.data, want["scan1"])
    assert np.array_equal(s.static_background, want["static_background"])
    assert s.step_sizes == (1.5, 1.5) and s.xmap.shape == (3, 3)
    det = s.detector
    assert det.shape == (60, 60) and det.navigation_shape == (3, 3) and det.binning == 8
    assert np.array_equal(det.pc.reshape(-1, 3), want["pc1"])
    assert (det.tilt, det.azimuthal, det.sample_tilt, det.px_size) == (1.5, 0.0, 70.0, 70.0)
    # the same patterns as the golden vectors of the indexing tests were made from
    assert s.data.shape == (3, 3, 60, 60)


def test_load_compressed_scan_and_several_scans(want):
    s1, s2 = ka.load(PATH, scan_group_names=["Scan 1", "Scan 2"])
    assert np.array_equal(s2.data, want["scan2"]) and np.array_equal(s1.data, want["scan1"])
    assert s2.step_sizes == (0.25, 0.5)
    assert s2.detector.navigation_shape == (1,) and np.allclose(s2.detector.pc, [[0.42, 0.21, 0.5]])


def test_short_file_is_zero_padded(want):
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        s = ka.load(PATH, scan_group_names="Scan 3")
    assert any("zero padding incomplete patterns" in str(x.message) for x in w)
    assert s.data.shape == (9, 60, 60) and s.data.dtype == np.float32  # (1, 9) navigation shape squeezed
    assert np.array_equal(s.data, want["scan3"]) and s.static_background is None
    assert np.allclose(s.detector.pc, [[0.5, 0.5, 0.5]])


def test_errors(want, tmp_path):
    with pytest.raises(_lib.KpdiError, match="Scan 'Scan 9' is not among the scans"):
        ka.load(PATH, scan_group_names="Scan 9")
    with pytest.raises(_lib.KpdiError, match="cannot open"):
        ka.load(str(tmp_path / "nothing.h5"))
    # like the reference's reader (tests/test_io/test_kikuchipy_h5ebsd.py:201-209): an OSError; with several names one
    # that is missing is a warning, and an error only when it is the only one
    with pytest.raises(OSError, match="Scan 'Scan 9' is not among the"):
        ka.load(PATH, scan_group_names=["Scan 9"])
    with pytest.warns(UserWarning, match="Scan 'Scan 9' is not among "):
        s1, s2 = ka.load(PATH, scan_group_names=["Scan 1", "Scan 2", "Scan 9"])
    assert np.array_equal(s1.data, want["scan1"]) and np.array_equal(s2.data, want["scan2"])
    with pytest.raises(OSError, match="cannot open"):
        ka.load(str(tmp_path / "nothing.h5"))
    bad = tmp_path / "text.h5"
    bad.write_text("not hdf5")
    with pytest.raises(_lib.KpdiError, match="cannot open"):
        ka.load(str(bad))
    buf = np.empty(10, np.uint8)
    rc = _lib.load().kpdi_h5ebsd_read_patterns(PATH.encode(), b"Scan 1", buf.ctypes.data, buf.nbytes)
    assert rc != 0 and "output buffer holds 10 bytes" in _lib.last_error()


@pytest.mark.gpu
def test_file_to_hbm_and_index(want):
    """File -> pinned host buffer -> HBM -> pre-processing -> indexing, equal to the array route."""
    exp = want["scan1"].reshape(9, 60, 60)
    rng = np.random.default_rng(0)
    dic = rng.random((500, 60, 60)).astype(np.float32)
    with _lib.Context(0) as ctx:
        ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 5)
        info = ctx.set_experimental_h5ebsd(PATH)
        assert info.scan == b"Scan 1" and ctx.n_experimental == 9
        assert np.array_equal(ctx.get_experimental(), exp)
        ctx.push_dictionary_chunk(dic, 0)
        a = ctx.finalize(5)
        ctx.set_experimental(exp)
        ctx.push_dictionary_chunk(dic, 0)
        b = ctx.finalize(5)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        nav = np.zeros(9, bool)
        nav[[1, 4]] = True
        ctx.set_experimental_h5ebsd(PATH, "Scan 2", nav)
        assert ctx.n_experimental == 7
