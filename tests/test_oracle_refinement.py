"""Pin the oracle's refinement restatement to the reference:
tests/golden/refinement.npz (reference run under SciPy 1.7.1) and
refinement_scipy115.npz (the same solver functions under SciPy 1.15.3, whose
Nelder-Mead treats bounds and `maxfev` differently - see gen_refinement)."""

import numpy as np
import pytest
import scipy

from conftest import load_golden
from oracle import kpdi_oracle as ko

NROWS = NCOLS = 60
SCIPY_MODERN = tuple(int(v) for v in scipy.__version__.split(".")[:2]) >= (1, 11)


@pytest.fixture(scope="module")
def g():
    return load_golden("refinement.npz")


@pytest.fixture(scope="module")
def g115():
    return load_golden("refinement_scipy115.npz")


@pytest.fixture(scope="module")
def master():
    p = load_golden("projection.npz")
    return ko.refinement_master_pattern(p["mp_upper"], p["mp_lower"])


def test_rotation_from_euler():
    assert np.allclose(ko.rotation_from_euler(0.1, 0.2, 0.3), [0.97517033, -0.09933467, 0.00996671, -0.19767681])
    q = ko.rotation_from_euler(3.0, 2.0, 3.0)
    assert q[0] >= 0 and np.isclose(np.sum(q**2), 1)


def test_master_pattern_rescale(g, master):
    mpu, mpl = master
    assert mpu.dtype == np.float32 and mpu.min() == -1 and mpu.max() == 1
    assert np.array_equal(mpu[::40, ::40], g["mp_f32_upper_sample"])
    f = np.ones((3, 3), np.float32)
    assert ko.refinement_master_pattern(f, f)[0] is f  # float32 is passed through


def test_prepare_pattern(g):
    keep = ko.circular_window((60, 60)).astype(bool).ravel()
    p, sq = ko.prepare_refinement_pattern(g["patterns"][0], False)
    assert p.dtype == np.float32 and np.allclose(p, g["prep_u8"], rtol=0, atol=1e-4)
    assert np.isclose(sq, g["prep_u8_sqnorm"], rtol=1e-6)
    p, sq = ko.prepare_refinement_pattern(g["patterns"][1][keep].astype(np.float32) * 0.37, True)
    assert np.allclose(p, g["prep_f32_masked"], rtol=0, atol=1e-6) and np.isclose(sq, g["prep_f32_masked_sqnorm"], rtol=1e-6)
    assert abs(float(p.mean())) < 1e-6


def test_prepare_pattern_known_answers():
    """tests/test_indexing/test_ebsd_refinement.py:55-77 of the reference: squared norms of
    the prepared test pattern (rescaled float32: 34.007; plain: 8.502 for its values)."""
    # the arithmetic identity those numbers rest on: rescaling to [-1, 1] scales the centred norm
    rng = np.random.default_rng(0)
    pat = rng.random(3600).astype(np.float32)
    p1, s1 = ko.prepare_refinement_pattern(pat, True)
    p2, s2 = ko.prepare_refinement_pattern(pat, False)
    span = float(pat.max() - pat.min())
    assert np.isclose(s1, s2 * (2 / span) ** 2, rtol=1e-4)


def objective_inputs(g, i, mask=None):
    pat = g["patterns"][i] if mask is None else g["patterns"][i][mask]
    return ko.prepare_refinement_pattern(pat, False)


def test_objective_values(g, master):
    mpu, mpl = master
    om = g["om_detector_to_sample"]
    vals = g["objective_values"]
    for i in range(4):
        p, sq = objective_inputs(g, i)
        dc = ko.direction_cosines_fixed_pc(ko.gnomonic_bounds((60, 60), g["pc0"][i]), g["pc0"][i][2], 60, 60, om)
        for j, o in enumerate(g["objective_offsets"]):
            x = np.concatenate([g["eu0"][i], g["pc0"][i]]) + o
            got = [
                ko.refinement_objective(x[:3], "ori", p, sq, mpu, mpl, direction_cosines=dc),
                ko.refinement_objective(x[3:], "pc", p, sq, mpu, mpl, rotation=ko.rotation_from_euler(*g["eu0"][i]),
                                        nrows=60, ncols=60, om_detector_to_sample=om),
                ko.refinement_objective(x, "ori_pc", p, sq, mpu, mpl, nrows=60, ncols=60, om_detector_to_sample=om),
            ]
            assert np.allclose(got, vals[i, j], rtol=0, atol=2e-6), (i, j, got, vals[i, j])
    keep = ko.circular_window((60, 60)).astype(bool).ravel()
    p, sq = objective_inputs(g, 2, keep)
    got = ko.refinement_objective(np.concatenate([g["eu0"][2], g["pc0"][2]]), "ori_pc", p, sq, mpu, mpl,
                                  signal_mask_keep=keep, nrows=60, ncols=60, om_detector_to_sample=om)
    assert abs(got - g["objective_masked"]) < 2e-6


def solver_close(got, want, nvar):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert abs(got[0] - want[0]) < 1e-5, (got, want)           # score
    assert got[1] == want[1], (got, want)                      # number of evaluations
    assert np.allclose(got[2:2 + nvar], want[2:2 + nvar], rtol=0, atol=1e-5), (got, want)
    if len(want) > 2 + nvar:
        assert got[2 + nvar] == want[2 + nvar]


@pytest.mark.parametrize("i", range(4))
def test_solver_orientation_unbounded(g, g115, master, i):
    """No bounds: SciPy 1.7.1 and 1.15.3 walk the same simplex path."""
    mpu, mpl = master
    om = g["om_detector_to_sample"]
    assert np.allclose(g["ori_nm"], g115["ori_nm"], rtol=0, atol=1e-9)
    dc = ko.direction_cosines_fixed_pc(ko.gnomonic_bounds((60, 60), g["pc0"][i]), g["pc0"][i][2], 60, 60, om)
    got = ko.refine_solver(g["patterns"][i], "ori", g["eu0"][i], mpu, mpl, False, direction_cosines=dc)
    solver_close(got, g["ori_nm"][i], 3)


@pytest.mark.skipif(not SCIPY_MODERN, reason="bounded Nelder-Mead goldens were made with SciPy >= 1.11")
def test_solver_variants_scipy115(g, g115, master):
    mpu, mpl = master
    om = g["om_detector_to_sample"]
    keep = ko.circular_window((60, 60)).astype(bool).ravel()
    tr = np.deg2rad(2.0)

    def dc(i, mask=None):
        return ko.direction_cosines_fixed_pc(ko.gnomonic_bounds((60, 60), g["pc0"][i]), g["pc0"][i][2], 60, 60, om, mask)

    for i in range(2):
        # trust region
        b = np.stack([g["eu0"][i] - tr, g["eu0"][i] + tr], axis=1)
        got = ko.refine_solver(g["patterns"][i], "ori", g["eu0"][i], mpu, mpl, False, bounds=b, direction_cosines=dc(i))
        solver_close(got, g115["ori_nm_bounds"][i], 3)
        # signal mask
        got = ko.refine_solver(g["patterns"][i][keep], "ori", g["eu0"][i], mpu, mpl, False, direction_cosines=dc(i, keep))
        solver_close(got, g115["ori_nm_masked"][i], 3)
        # evaluation budget
        got = ko.refine_solver(g["patterns"][i], "ori", g["eu0"][i], mpu, mpl, False, direction_cosines=dc(i),
                               method_kwargs=dict(options=dict(maxfev=30)))
        solver_close(got, g115["ori_nm_maxfev30"][i], 3)
        # pseudo-symmetry starts
        got = ko.refine_solver(g["patterns"][i], "ori", g["ori_nm_ps_starts"][i], mpu, mpl, False, direction_cosines=dc(i))
        solver_close(got, g115["ori_nm_ps"][i], 3)
        # projection centre
        got = ko.refine_solver(g["patterns"][i], "pc", g["pc0"][i], mpu, mpl, False,
                               rotation=ko.rotation_from_euler(*g["eu0"][i]), nrows=60, ncols=60,
                               om_detector_to_sample=om)
        solver_close(got, g115["pc_nm"][i], 3)
    # orientation + PC with bounds (6 control variables)
    i = 0
    x0 = np.concatenate([g["eu0"][i], g["pc0"][i]])
    tr6 = np.array([tr, tr, tr, 0.02, 0.02, 0.02])
    got = ko.refine_solver(g["patterns"][i], "ori_pc", x0, mpu, mpl, False, bounds=np.stack([x0 - tr6, x0 + tr6], axis=1),
                           nrows=60, ncols=60, om_detector_to_sample=om)
    solver_close(got, g115["ori_pc_nm_bounds"][i], 6)
    # refinement moved towards the truth and raised the score
    assert np.all(np.abs(g115["ori_nm"][:, 2:5] - g["eu_true"]).max(axis=1) < np.abs(g["eu0"] - g["eu_true"]).max(axis=1))
