"""Host-side mirror of the dictionary-generation interface (no GPU): the
`EBSDDetector` geometry against the reference's outputs and its tests' known
answers, `EBSDMasterPattern` validation / rescale rule, `ProjectedDictionary`
array protocol."""

import numpy as np
import pytest

from conftest import load_golden

import kikuchipy_amd as ka
from kikuchipy_amd.detectors import sample_to_detector_matrix

PC1 = (0.4210, 0.7794, 0.5049)  # the reference tests' `pc1`


@pytest.fixture(scope="module")
def g():
    return load_golden("projection.npz")


@pytest.mark.parametrize("shape,x_range,y_range", [
    ((60, 60), [-0.833828, 1.146762], [-0.436918, 1.543672]),
    ((510, 510), [-0.833828, 1.146762], [-0.436918, 1.543672]),
    ((1, 1), [-0.833828, 1.146762], [-0.436918, 1.543672]),
    ((480, 640), [-1.111771, 1.529016], [-0.436918, 1.543672]),
])
def test_gnomonic_range(shape, x_range, y_range):
    """tests/test_detectors/test_ebsd_detector.py:228-241 of the reference."""
    det = ka.EBSDDetector(shape=shape, pc=PC1)
    assert np.allclose([det.x_min, det.x_max], x_range, atol=1e-6)
    assert np.allclose([det.y_min, det.y_max], y_range, atol=1e-6)
    assert np.allclose(det.gnomonic_bounds, x_range + y_range, atol=1e-6)


@pytest.mark.parametrize("shape,pc,px_size,binning,version,desired", [
    ((60, 60), [-3.4848, 114.2016, 15767.7], 59.2, 8, 5, [0.50726, 0.26208, 0.55489]),
    ((61, 61), [-10.6320, 145.5187, 19918.9], 59.2, 8, 5, [0.52178688525, 0.20180594262, 0.68948341272]),
    ((80, 60), [-0.55, -13.00, 16075.2], 50, 6, 5, [0.50153, 0.52708, 0.66980]),
    ((80, 60), [0.55, -13.00, 16075.2], 50, 6, 4, [0.50153, 0.52708, 0.66980]),
    ((480, 640), [0, 0, 15000], 50, 1, 5, [0.5, 0.5, 0.625]),
])
def test_pc_from_emsoft(shape, pc, px_size, binning, version, desired):
    """tests/test_detectors/test_ebsd_detector.py:553-617."""
    det = ka.EBSDDetector(shape=shape, pc=pc, px_size=px_size, binning=binning, convention=f"emsoft{version}")
    assert np.allclose(det.pc, desired, atol=1e-5)


def test_pc_from_emsoft_no_version():
    det = ka.EBSDDetector(shape=(60, 60), pc=[3.4848, 114.2016, 15767.7], px_size=59.2, binning=8,
                          convention="emsoft")
    assert np.allclose(det.pc, [0.49274, 0.26208, 0.55489], atol=1e-5)


@pytest.mark.parametrize("shape,pc,convention,desired", [
    ((60, 60), [0.35, 1, 0.65], "tsl", [0.35, 0, 0.65]),
    ((60, 80), [0.35, 1, 0.65], "tsl", [0.35, 0, 0.65]),
    ((60, 60), [0.1, 0.2, 0.3], "amatek", [0.1, 0.8, 0.3]),
    ((60, 60), [0.6, 0.6, 0.6], "edax", [0.6, 0.4, 0.6]),
    ((60, 60), [0.25, 0, 0.75], "oxford", [0.25, 1, 0.75]),
    ((60, 80), [0.25, 0, 0.75], "oxford", [0.25, 1, 1]),
    ((1, 1), [0.1, 0.2, 0.3], "Bruker", [0.1, 0.2, 0.3]),
])
def test_pc_conventions(shape, pc, convention, desired):
    """tests/test_detectors/test_ebsd_detector.py:646-698."""
    assert np.allclose(ka.EBSDDetector(shape=shape, pc=pc, convention=convention).pc, desired, atol=1e-2)


def test_pc_convention_raises():
    with pytest.raises(ValueError, match="Invalid projection/pattern center "):
        ka.EBSDDetector(pc=PC1, convention="nordif")
    with pytest.raises(ValueError, match="must be \\(PCx, PCy, PCz\\)"):
        ka.EBSDDetector(shape=(60, 60), pc=np.ones((4, 2)) * 0.5)


def test_detector_with_one_pc_per_point():
    pcs = np.array([[[0.4, 0.5, 0.4], [0.6, 0.5, 0.4]], [[0.4, 0.5, 0.6], [0.6, 0.5, 0.6]]])
    det = ka.EBSDDetector(shape=(60, 60), pc=pcs)
    assert det.navigation_shape == (2, 2) and det.navigation_size == 4
    assert det.pc_flattened.shape == (4, 3) and det.gnomonic_bounds.shape == (2, 2, 4)
    assert np.allclose(det.pc_average, [0.5, 0.5, 0.5])
    one = ka.EBSDDetector(shape=(60, 60), pc=pcs[1, 0])
    assert np.allclose(det.gnomonic_bounds[1, 0], one.gnomonic_bounds)
    mp = ka.EBSDMasterPattern(np.zeros((11, 11)))
    with pytest.raises(ValueError, match="must be equal to `rotations.shape`"):
        mp.get_patterns(np.array([[1.0, 0, 0, 0]]), det)
    with pytest.raises(NotImplementedError, match="pass compute=True"):  # (a LAZY result is one-dimensional)
        mp.get_patterns(np.tile([1.0, 0, 0, 0], (2, 2, 1)), det)
    flat = ka.EBSDDetector(shape=(60, 60), pc=pcs.reshape(4, 3))
    lazy = mp.get_patterns(np.tile([1.0, 0, 0, 0], (4, 1)), flat).data  # one PC per rotation, lazily (no GPU touched yet)
    assert lazy.shape == (4, 60, 60) and lazy.pcs.shape == (4, 3) and lazy[1:3].pcs.shape == (2, 3)
    assert np.array_equal(lazy[1:3].pcs, pcs.reshape(4, 3)[1:3])
    tsl = ka.EBSDDetector(shape=(60, 80), pc=[[0.35, 1, 0.65], [0.1, 0.2, 0.3]], convention="tsl")
    assert np.allclose(tsl.pc, [[0.35, 0, 0.65], [0.1, 0.8, 0.3]])


def test_sample_to_detector_golden(g):
    assert np.allclose(sample_to_detector_matrix(70.0, 0, 0, 0), g["det60__s2d"], rtol=0, atol=1e-15)
    det = ka.EBSDDetector(shape=(48, 60), pc=(0.52, 0.71, 0.63), sample_tilt=69.5, tilt=5.0, azimuthal=3.0,
                          twist=1.5)
    assert np.allclose(det.sample_to_detector, g["det48x60__s2d"], rtol=0, atol=1e-15)
    assert np.allclose(det.detector_to_sample @ det.sample_to_detector, np.eye(3), atol=1e-14)
    assert det.navigation_shape == (1,) and det.size == 48 * 60 and det.aspect_ratio == 1.25
    assert "shape (Ny, Nx):     (48, 60)" in repr(det)


# ---------------------------------------------------------------- master pattern
def test_master_pattern_shapes_and_energy_selection():
    data = np.arange(2 * 3 * 5 * 5, dtype=np.float32).reshape(2, 3, 5, 5)
    mp = ka.EBSDMasterPattern(data, hemisphere="both", energies=[10, 15, 20])
    up, lo = mp._get_master_pattern_arrays_from_energy()
    assert np.array_equal(up, data[0, 2]) and np.array_equal(lo, data[1, 2])  # highest energy by default
    up, lo = mp._get_master_pattern_arrays_from_energy(15)
    assert np.array_equal(up, data[0, 1]) and np.array_equal(lo, data[1, 1])
    mp = ka.EBSDMasterPattern(data[0], energies=[10, 15, 20])
    up, lo = mp._get_master_pattern_arrays_from_energy(10)
    assert up is lo and np.array_equal(up, data[0, 0]) and mp.hemisphere == "upper"
    mp = ka.EBSDMasterPattern(data[:, 0])
    assert mp.hemisphere == "both"
    mp = ka.EBSDMasterPattern(data[0, 0])
    up, lo = mp._get_master_pattern_arrays_from_energy()
    assert up is lo
    with pytest.raises(ValueError, match="does not match hemisphere"):
        ka.EBSDMasterPattern(data, hemisphere="upper", energies=[10, 15, 20])
    with pytest.raises(ValueError, match="one value per master pattern"):
        ka.EBSDMasterPattern(data, hemisphere="both", energies=[10, 15])


def test_master_pattern_suitability():
    """signals/ebsd_master_pattern.py:331-377; tests/test_signals/test_ebsd_master_pattern.py:282-296."""
    det = ka.EBSDDetector(shape=(6, 6))
    rot = np.array([[1.0, 0, 0, 0]])
    mp = ka.EBSDMasterPattern(np.zeros((11, 11)), projection="stereographic")
    with pytest.raises(NotImplementedError, match="square Lambert projection"):
        mp.get_patterns(rot, det)
    mp = ka.EBSDMasterPattern(np.zeros((2, 11, 11)), has_inversion_symmetry=None)
    with pytest.raises(AttributeError, match="Master pattern `phase` attribute"):
        mp.get_patterns(rot, det)
    mp = ka.EBSDMasterPattern(np.zeros((10, 11, 11)), energies=np.arange(10), has_inversion_symmetry=False)
    with pytest.raises(AttributeError, match="For point groups without inversion"):
        mp.get_patterns(rot, det)
    mp = ka.EBSDMasterPattern(np.zeros((11, 11)))
    with pytest.raises(ValueError, match="can only have one or two dimensions"):
        mp.get_patterns(np.zeros((2, 2, 2, 4)), det)
    # tests/test_signals/test_ebsd_master_pattern.py: the texts of _utils/exceptions.py
    with pytest.raises(ValueError, match="Unknown projection 'gnomonic'"):
        ka.EBSDMasterPattern(np.zeros((11, 11)), projection="gnomonic")
    with pytest.raises(ValueError, match="Unknown hemisphere 'west'"):
        ka.EBSDMasterPattern(np.zeros((11, 11)), hemisphere="west")
    assert ka.EBSDMasterPattern(np.zeros((2, 11, 11)), projection="Lambert", hemisphere="BOTH").hemisphere == "both"


def test_get_patterns_lazy_protocol_and_rescale_rule(g):
    det = ka.EBSDDetector(shape=(60, 60), pc=PC1)
    rot = g["di_rot"]
    mp = ka.EBSDMasterPattern(np.stack([g["mp_upper"], g["mp_lower"]]), phase_name="ni")
    sim = mp.get_patterns(rot, det, chunk_shape=500)  # uint8 master pattern -> float32: rescaled to [-1, 1]
    d = sim.data
    assert isinstance(d, ka.ProjectedDictionary) and d.rescale and (d.out_min, d.out_max) == (-1.0, 1.0)
    assert d.shape == (1200, 60, 60) and d.ndim == 3 and d.dtype == np.float32 and len(d) == 1200
    assert d.chunksize == (500, 60, 60)
    assert sim.xmap.shape == (1200,) and sim.xmap.phase_name == "ni"
    assert sim._navigation_shape_rc == (1200,) and sim._signal_shape_rc == (60, 60)
    part = d[100:350]
    assert part.shape == (250, 60, 60) and np.array_equal(part.rotations, rot[100:350])
    assert d[:, :, :].shape == d.shape
    with pytest.raises(IndexError, match="first axis"):
        d[:, 3]
    # same dtype as the master pattern: no rescale (signals/ebsd_master_pattern.py:224-233)
    d8 = mp.get_patterns(rot, det, dtype_out=np.uint8).data
    assert not d8.rescale and d8.dtype == np.uint8
    mpf = ka.EBSDMasterPattern(g["mp_upper"].astype(np.float32))
    assert not mpf.get_patterns(rot, det).data.rescale
    assert mpf.get_patterns(rot, det, dtype_out=np.uint16).data.out_max == 65535.0
    # default chunk: 8 GiB of float32 patterns (it only exists in device memory)
    assert mpf.get_patterns(rot, det).data.chunksize == (1200, 60, 60)
    big = ka.ProjectedDictionary(None, None, np.zeros((800000, 4)), det, False, 1, 2)
    assert big.chunksize[0] == (8 << 30) // (4 * 3600)


def test_detector_like_the_reference_tests():
    """Known answers and error texts of the reference's tests/test_detectors/test_ebsd_detector.py:41-176 (restated as
    data): validated attributes, derived sizes, the text representation, PC components that can be set."""
    import kikuchipy_amd as ka

    pc1 = [0.4210, 0.7794, 0.5049]
    det = ka.EBSDDetector(shape=(1, 2), px_size=3, binning=4, tilt=5, pc=pc1)
    assert det.shape == (1, 2) and det.aspect_ratio == 2 and np.issubdtype(det.pc.dtype, np.floating)
    assert all(type(v) is float for v in (det.sample_tilt, det.tilt, det.azimuthal, det.px_size))
    d = ka.EBSDDetector()
    for attr, value, text in [("shape", 2, "Invalid shape 2. Must be an iterable of"),
                              ("shape", (2,), r"Invalid shape \(2,\). Must be an "),
                              ("shape", (2, "a"), r"Invalid shape \(2, 'a'\). Must be an "),
                              ("binning", (3, 4), r"Invalid binning \(3, 4\). Must be an "),
                              ("sample_tilt", "a", "Invalid sample tilt "), ("tilt", "a", "Invalid detector tilt "),
                              ("azimuthal", "a", "Invalid azimuthal "), ("twist", "a", "Invalid twist "),
                              ("px_size", "a", "Invalid pixel size ")]:
        with pytest.raises(ValueError, match=text):
            setattr(d, attr, value)
    for nav_shape, want_shape, want_dim in [((), (1,), 1), ((1,), (1,), 1), ((10, 1), (10,), 1), ((10, 10, 1), (10, 10), 2)]:
        det = ka.EBSDDetector(pc=np.tile(pc1, nav_shape))
        assert det.navigation_shape == want_shape and det.navigation_dimension == want_dim
    for shape, px_size, binning, pc, ssd, width, height, size, unbinned, px_binned in [
        ((60, 60), 70, 8, [1, 1, 0.5], 16800, 33600, 33600, 3600, (480, 480), 560),
        ((60, 60), 70, 8, [1, 1, 0.7], 23520, 33600, 33600, 3600, (480, 480), 560),
        ((480, 460), 70, 0.5, [1, 1, 0.7], 11760, 16100, 16800, 220800, (240, 230), 35),
        ((340, 680), 40, 2, [1, 1, 0.7], 19040, 54400, 27200, 231200, (680, 1360), 80),
    ]:
        det = ka.EBSDDetector(shape=shape, px_size=px_size, binning=binning, pc=pc)
        assert np.isclose(det.specimen_scintillator_distance, ssd) and (det.width, det.height, det.size) == (width, height, size)
        assert det.unbinned_shape == unbinned and det.px_size_binned == px_binned
    det = ka.EBSDDetector(shape=(1, 2), px_size=3, binning=4, tilt=5, azimuthal=2, twist=1.02, pc=[0.421, 0.779, 0.505])
    assert repr(det) == (
        "EBSDDetector\n"
        "  shape (Ny, Nx):     (1, 2)\n"
        "  pc (PCx, PCy, PCz): (0.421, 0.779, 0.505)\n"
        "  sample_tilt:        70.0\N{DEGREE SIGN}\n"
        "  tilt:               5.0\N{DEGREE SIGN}\n"
        "  azimuthal:          2.0\N{DEGREE SIGN}\n"
        "  twist:              1.02\N{DEGREE SIGN}\n"
        "  binning:            4\n"
        "  px_size:            3.0 um"
    )
    one = ka.EBSDDetector(pc=[0.421, 0.779, 0.505])
    two = one.deepcopy()
    one.pcx += 0.1
    assert np.allclose(one.pcx, 0.521) and np.allclose(two.pcx, 0.421)
    # ranges and scales are what the gnomonic bounds say
    det = ka.EBSDDetector(shape=(60, 80), pc=np.tile([0.4, 0.6, 0.5], (2, 3, 1)))
    assert det.x_range.shape == (2, 3, 2) and det.x_scale.shape == (2, 3) and det.r_max.shape == (2, 3)
    assert np.allclose(det.x_range[..., 1] - det.x_range[..., 0], det.x_scale * 79)
    assert np.allclose(det.gnomonic_bounds, np.concatenate([det.x_range, det.y_range], axis=-1))
    assert np.array_equal(det.bounds, [0, 79, 0, 59])
