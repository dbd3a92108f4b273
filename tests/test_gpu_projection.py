"""GPU parity of the on-device dictionary generation: the
projection kernel through the C ABI against the reference's own outputs
(tests/golden/projection.npz) and the oracle.

Tolerances: the reference computes in f64 (Numba, fastmath) and rounds once to
float32; the kernel does the same arithmetic in f64 with its own libm, so
float32 outputs agree to a few float32 ulp (3e-7 relative to the pattern's
range), direction cosines to 1e-14, and the indexing scores to 1e-5."""

import numpy as np
import pytest

from conftest import load_golden
from oracle import kpdi_oracle as ko

pytestmark = pytest.mark.gpu
ATOL = 1e-5

DET60 = dict(shape=(60, 60), pc=(0.4210, 0.7794, 0.5049), sample_tilt=70.0)
DET48 = dict(shape=(48, 60), pc=(0.52, 0.71, 0.63), sample_tilt=69.5, tilt=5.0, azimuthal=3.0, twist=1.5)


@pytest.fixture(scope="module")
def g():
    return load_golden("projection.npz")


@pytest.fixture(scope="module")
def ctx():
    from kikuchipy_amd import _lib

    c = _lib.Context(0)
    yield c
    c.close()


def set_detector(ctx, d):
    m = ko.sample_to_detector_matrix(d["sample_tilt"], d.get("tilt", 0), d.get("azimuthal", 0), d.get("twist", 0))
    ctx.set_detector(ko.gnomonic_bounds(d["shape"], d["pc"]), d["pc"][2], d["shape"][0], d["shape"][1], m.T)


@pytest.mark.parametrize("name,d", [("det60", DET60), ("det48x60", DET48)])
def test_direction_cosines(ctx, g, name, d):
    set_detector(ctx, d)
    dc = ctx.get_direction_cosines()
    assert np.allclose(dc, g[f"{name}__dc"], rtol=0, atol=1e-14)


CASES = {
    # name: (master pattern dtype, lower, rescale, omin, omax, dtype_out)
    "u8mp_f32": (np.uint8, "lower", True, -1, 1, np.float32),
    "f32mp_f32": (np.float32, "lower", False, 1, 2, np.float32),
    "hemis_f32": (np.float32, "inverted", False, 1, 2, np.float32),
    "f32mp_u8": (np.float32, "lower", True, 0, 255, np.uint8),
    "u8mp_u8": (np.uint8, "lower", False, 1, 2, np.uint8),
}


def master_arrays(g, dtype, lower):
    up = g["mp_upper"].astype(dtype)
    lo = g["mp_lower"].astype(dtype) if lower == "lower" else (255 - g["mp_upper"]).astype(dtype)
    return up, lo


@pytest.mark.parametrize("name", sorted(CASES))
def test_patterns_golden(ctx, g, name):
    mp_dtype, lower, rescale, omin, omax, dtype_out = CASES[name]
    ctx.set_master_pattern(*master_arrays(g, mp_dtype, lower))
    set_detector(ctx, DET60)
    got = ctx.project_patterns(g["rot8"], rescale, omin, omax, dtype_out)
    want = g[f"{name}__patterns"]
    assert got.dtype == want.dtype and got.shape == want.shape
    if dtype_out == np.float32:
        assert np.allclose(got, want, rtol=3e-7, atol=3e-7 * np.abs(want).max())
    else:  # truncation: a value within rounding of an integer may fall on either side
        diff = np.abs(got.astype(int) - want.astype(int))
        assert diff.max() <= 1 and np.mean(diff != 0) < 1e-3


def test_patterns_other_detector(ctx, g):
    ctx.set_master_pattern(*master_arrays(g, np.float32, "lower"))
    set_detector(ctx, DET48)
    got = ctx.project_patterns(g["rot8"][:4])
    assert np.allclose(got, g["det48x60_f32__patterns"], rtol=3e-7, atol=1e-4)


def test_one_pc_per_pattern(ctx, g):
    """`_project_patterns_from_master_pattern_with_varying_pc`: direction cosines formed on the device."""
    om = g["det60__s2d"].T
    ctx.set_master_pattern(*master_arrays(g, np.float32, "lower"))
    got = ctx.project_patterns_varying_pc(g["rot8"][:4], g["varpc__pcs"], (60, 60), om)
    assert np.allclose(got, g["varpc_f32__patterns"], rtol=3e-7, atol=1e-4)
    ctx.set_master_pattern(g["mp_upper"], g["mp_lower"])
    got = ctx.project_patterns_varying_pc(g["rot8"][:4], g["varpc__pcs"], (60, 60), om, True, -1, 1)
    assert np.allclose(got, g["varpc_u8mp_f32__patterns"], rtol=3e-7, atol=3e-7)
    # a detector above 4096 pixels (the recompute path) against the oracle
    rng = np.random.default_rng(9)
    up = rng.random((101, 101)).astype(np.float32)
    pcs = np.array([0.45, 0.6, 0.55]) + rng.uniform(-0.05, 0.05, (3, 3))
    q = rng.standard_normal((3, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    m = ko.sample_to_detector_matrix(70.0, 10.0, 2.0, 1.0)
    ctx.set_master_pattern(up)
    got = ctx.project_patterns_varying_pc(q, pcs, (70, 90), m.T, True, 0, 1)
    want = ko.project_patterns_varying_pc(q, pcs, (70, 90), m.T, up, up, True, 0, 1)
    assert np.allclose(got, want, rtol=3e-7, atol=3e-7)


def test_get_patterns_with_one_pc_per_rotation(g):
    import kikuchipy_amd as ka

    mp = ka.EBSDMasterPattern(np.stack([g["mp_upper"], g["mp_lower"]]))
    det = ka.EBSDDetector(shape=(60, 60), pc=g["varpc__pcs"].reshape(2, 2, 3))
    sim = mp.get_patterns(g["rot8"][:4].reshape(2, 2, 4), det, compute=True)
    assert sim.data.shape == (2, 2, 60, 60)
    assert np.allclose(sim.data.reshape(4, -1), g["varpc_u8mp_f32__patterns"], rtol=3e-7, atol=3e-7)


def test_lazy_dictionary_with_one_pc_per_rotation(g):
    """`get_patterns(rotations, detector)` with a PC for every rotation and `compute=False` - the branch
    `nav_shape_det != (1,)` of signals/ebsd_master_pattern.py:236-241, :274-283 - is a LAZY dictionary whose chunks are
    projected with their own PCs on the device inside the indexing loop (`kpdi_push_rotations_chunk_varying_pc`): its
    patterns are the reference's (golden `varpc_*`), and indexing against it equals indexing against the computed array."""
    import kikuchipy_amd as ka

    mp = ka.EBSDMasterPattern(np.stack([g["mp_upper"], g["mp_lower"]]))
    det = ka.EBSDDetector(shape=(60, 60), pc=g["varpc__pcs"])
    sim = mp.get_patterns(g["rot8"][:4], det)                      # compute=False
    assert type(sim.data).__name__ == "ProjectedDictionary" and sim.data.pcs.shape == (4, 3)
    assert np.allclose(sim.data.compute().reshape(4, -1), g["varpc_u8mp_f32__patterns"], rtol=3e-7, atol=3e-7)
    # a dictionary worth indexing: 700 rotations, PCs scattered around the detector's, chunks of 300
    rng = np.random.default_rng(21)
    q = rng.standard_normal((700, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    pcs = np.array([0.42, 0.78, 0.5]) + 0.02 * rng.standard_normal((700, 3))
    det = ka.EBSDDetector(shape=(60, 60), pc=pcs)
    lazy = mp.get_patterns(q, det, chunk_shape=300)
    dense = mp.get_patterns(q, det, compute=True)
    assert np.array_equal(lazy.data.compute(), dense.data)
    exp = np.clip((dense.data[::70] + 1) * 100 + 5 * rng.standard_normal((10, 60, 60)), 0, 255).astype(np.uint8)  # (float32 output is rescaled to [-1, 1])
    s = ka.EBSD(exp)
    a = s.dictionary_indexing(lazy, keep_n=5, verbose=False)
    b = s.dictionary_indexing(dense, keep_n=5, verbose=False)
    assert np.array_equal(a.scores, b.scores) and np.array_equal(a.simulation_indices, b.simulation_indices)
    assert list(a.simulation_indices[:, 0]) == list(range(0, 700, 70))
    c = s.dictionary_indexing(lazy, keep_n=5, devices=[0, 0], verbose=False)   # ... and over a group's members
    assert np.array_equal(a.scores, c.scores) and np.array_equal(a.simulation_indices, c.simulation_indices)


def test_single_hemisphere_and_f64_output(ctx, g):
    up = g["mp_upper"].astype(np.float32)
    ctx.set_master_pattern(up)  # lower = upper
    set_detector(ctx, DET60)
    got = ctx.project_patterns(g["rot8"], dtype_out=np.float64)
    want = ko.project_patterns(g["rot8"], g["det60__dc"], up, up, dtype_out=np.float64)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-10)


@pytest.mark.parametrize("shape", [(8, 8), (61, 67), (80, 100)])
def test_vs_oracle_detector_sizes(ctx, g, shape):
    """Small, odd-sized and > 4096-pixel detectors (the recompute path), random
    master pattern with distinct hemispheres, with rescale."""
    rng = np.random.default_rng(shape[0])
    up = rng.random((101, 101)).astype(np.float32)
    lo = rng.random((101, 101)).astype(np.float32)
    d = dict(shape=shape, pc=(0.45, 0.6, 0.55), sample_tilt=70.0, tilt=10.0)
    q = rng.standard_normal((5, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    ctx.set_master_pattern(up, lo)
    set_detector(ctx, d)
    dc = ko.detector_direction_cosines(**d)
    for rescale in (False, True):
        got = ctx.project_patterns(q, rescale, -1, 1)
        want = ko.project_patterns(q, dc, up, lo, rescale, -1, 1)
        assert np.allclose(got, want, rtol=3e-7, atol=3e-7)


def test_dictionary_on_device_end_to_end(ctx, g):
    """Rotations in, best matches out; the dictionary only ever exists in HBM."""
    from kikuchipy_amd import _lib

    ctx.set_master_pattern(g["mp_upper"], g["mp_lower"])
    set_detector(ctx, DET60)
    rot, exp = g["di_rot"], g["di_exp"]
    ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 10)
    ctx.set_experimental(exp)
    for a in range(0, len(rot), 500):
        ctx.push_rotations_chunk(rot[a:a + 500], a, True, -1, 1)
    s, i = ctx.finalize(10)
    ko.assert_topk_parity(s, i, g["di_ncc_k10__scores"], g["di_ncc_k10__indices"], atol=ATOL)
    assert np.array_equal(i[:, 0], g["di_picks"])

    mask = ~ko.circular_window((60, 60)).astype(bool)
    ctx.set_problem(60, 60, mask, _lib.METRIC_NDP, 10)
    ctx.set_experimental(exp)
    ctx.push_rotations_chunk(rot, 0, True, -1, 1)
    s, i = ctx.finalize(10)
    ko.assert_topk_parity(s, i, g["di_ndp_k10_circ__scores"], g["di_ndp_k10_circ__indices"], atol=ATOL)


def test_generated_equals_pushed(ctx, g):
    """push_rotations_chunk == project_patterns + push_dictionary_chunk, bit for bit."""
    from kikuchipy_amd import _lib

    ctx.set_master_pattern(g["mp_upper"], g["mp_lower"])
    set_detector(ctx, DET60)
    rot, exp = g["di_rot"][:700], g["di_exp"]
    dic = ctx.project_patterns(rot, True, -1, 1)
    ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 20)
    ctx.set_experimental(exp)
    ctx.push_dictionary_chunk(dic, 0)
    a = ctx.finalize(20)
    ctx.set_experimental(exp)
    ctx.push_rotations_chunk(rot, 0, True, -1, 1)
    b = ctx.finalize(20)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_generated_dictionary_in_float64_arithmetic(ctx, g):
    """float64 arithmetic (KPDI_COMPUTE_F64) over a dictionary that is simulated on the device: the rescoring kernel
    reads the simulated float32 patterns where they were generated."""
    from kikuchipy_amd import _lib

    ctx.set_master_pattern(g["mp_upper"], g["mp_lower"])
    set_detector(ctx, DET60)
    rot, exp = g["di_rot"][:700], g["di_exp"]
    dic = ctx.project_patterns(rot, True, -1, 1)
    ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 20, _lib.COMPUTE_F64)
    ctx.set_experimental(exp)
    for start in (0, 300):
        ctx.push_rotations_chunk(rot[start:start + 400 if start else 300], start, True, -1, 1)
    scores, idx = ctx.finalize(20)
    ref_s, ref_i = ko.dictionary_indexing(exp, dic, metric="ncc", keep_n=20, dtype=np.float64)
    assert scores.dtype == np.float64 and np.abs(scores - ref_s).max() <= 1e-12 and np.array_equal(idx, ref_i)
    assert ctx.counters()["uncertified_patterns"] == 0
    ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 20)  # the module's context goes back to float32 arithmetic


def test_error_paths(g):
    from kikuchipy_amd import _lib

    with _lib.Context(0) as c:
        with pytest.raises(_lib.KpdiError, match="kpdi_set_detector"):
            c._dc_npix = 4
            c.project_patterns(g["rot8"])
        set_detector(c, DET60)
        with pytest.raises(_lib.KpdiError, match="kpdi_set_master_pattern"):
            c.project_patterns(g["rot8"])
        c.set_master_pattern(g["mp_upper"])
        with pytest.raises(_lib.KpdiError, match="out_max > out_min"):
            c.project_patterns(g["rot8"], True, 1, 1)
        c.set_problem(30, 30, None, _lib.METRIC_NCC, 1)
        c.set_experimental(np.ones((2, 30, 30), np.uint8))
        with pytest.raises(_lib.KpdiError, match="3600 pixels"):
            c.push_rotations_chunk(g["rot8"], 0)


# ------------------------------------------------------------------ Python interface
def test_get_patterns_and_dictionary_indexing_api(g):
    """`mp.get_patterns(rot, det)` -> `s.dictionary_indexing(sim)` as a kikuchipy
    user writes it; the lazy dictionary is generated chunk by chunk in HBM."""
    import kikuchipy_amd as ka

    det = ka.EBSDDetector(**DET60)
    mp = ka.EBSDMasterPattern(np.stack([g["mp_upper"], g["mp_lower"]]), phase_name="ni")
    sim = mp.get_patterns(g["di_rot"], det, chunk_shape=500)
    assert isinstance(sim.data, ka.ProjectedDictionary)
    s = ka.EBSD(g["di_exp"])
    res = s.dictionary_indexing(sim, keep_n=10, verbose=False)
    ko.assert_topk_parity(res.scores, res.simulation_indices, g["di_ncc_k10__scores"],
                          g["di_ncc_k10__indices"], atol=ATOL)
    assert res.phase_name == "ni"
    assert np.array_equal(res.rotations, g["di_rot"][res.simulation_indices])
    # n_per_iteration overrides the lazy chunk size; signal mask + ndp
    mask = ~ko.circular_window((60, 60)).astype(bool)
    res = s.dictionary_indexing(sim, metric="ndp", keep_n=10, n_per_iteration=333, signal_mask=mask,
                                verbose=False)
    ko.assert_topk_parity(res.scores, res.simulation_indices, g["di_ndp_k10_circ__scores"],
                          g["di_ndp_k10_circ__indices"], atol=ATOL)
    # compute=True gives the patterns themselves
    pats = mp.get_patterns(g["di_rot"][::100], det, compute=True).data
    assert pats.shape == (12, 60, 60) and pats.dtype == np.float32
    assert np.allclose(pats, g["di_dic_sample"], rtol=3e-7, atol=3e-7)
    # 2D rotation array, integer output dtype (materialised, truncated)
    p2 = mp.get_patterns(g["rot8"].reshape(2, 4, 4), det, dtype_out=np.uint8, compute=True).data
    assert p2.shape == (2, 4, 60, 60)
    diff = np.abs(p2.reshape(8, -1).astype(int) - g["u8mp_u8__patterns"].astype(int))
    assert diff.max() <= 1


def test_uint8_dictionary_takes_the_materialised_route(g):
    """A dictionary asked for as uint8 is generated, truncated and then matched
    like any other uint8 dictionary."""
    import kikuchipy_amd as ka

    det = ka.EBSDDetector(**DET60)
    mp = ka.EBSDMasterPattern(g["mp_upper"].astype(np.float32))
    rot = g["di_rot"][:300]
    sim = mp.get_patterns(rot, det, dtype_out=np.uint8)
    res = ka.EBSD(g["di_exp"]).dictionary_indexing(sim, keep_n=5, verbose=False)
    dic = sim.data.compute()
    assert dic.dtype == np.uint8 and dic.min() == 0 and dic.max() >= 254
    rs, ri = ko.dictionary_indexing(g["di_exp"], dic, keep_n=5)
    ko.assert_topk_parity(res.scores, res.simulation_indices, rs, ri, atol=ATOL)
