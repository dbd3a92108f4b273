"""GPU parity of the refinement path through the C ABI:
the device Nelder-Mead against SciPy bit for bit on analytic objectives, the
pattern preparation and objective values against the reference's outputs
(tests/golden/refinement.npz), and complete refinements against the reference's
SciPy solvers (refinement_scipy115.npz: SciPy 1.15.3, the version whose
Nelder-Mead the engine restates)."""

import numpy as np
import pytest
import scipy.optimize

from conftest import load_golden
from oracle import kpdi_oracle as ko

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    return load_golden("refinement.npz")


@pytest.fixture(scope="module")
def g115():
    return load_golden("refinement_scipy115.npz")


@pytest.fixture(scope="module")
def ctx():
    from kikuchipy_amd import _lib

    c = _lib.Context(0)
    p = load_golden("projection.npz")
    c.set_master_pattern(*ko.refinement_master_pattern(p["mp_upper"], p["mp_lower"]))
    yield c
    c.close()


# ------------------------------------------------------------------ the optimiser alone
def rosen(x):
    acc = None
    for i in range(len(x) - 1):
        d = x[i + 1] - x[i] * x[i]
        e = 1.0 - x[i]
        t = 100.0 * (d * d) + e * e
        acc = t if acc is None else acc + t
    return acc


def bowl(x):
    acc = None
    for i in range(len(x)):
        d = x[i] - 0.3 * float(i + 1)
        t = float(i + 1) * (d * d)
        acc = t if acc is None else acc + t
    return acc


NM_CASES = [
    # kind, x0, bounds, options
    (0, [-1.2, 1.0], None, {}),
    (0, [1.3, 0.7, 0.8], None, {}),
    (0, [1.3, 0.7, 0.8, 1.9, 1.2, 0.5], None, {}),
    (0, [0.0, 0.0, 0.0], None, {}),                                   # zero entries: the 0.00025 step
    (1, [2.0, -1.0, 0.5], None, dict(xatol=1e-8, fatol=1e-8)),
    (1, [2.0, 1.0, 0.5], ([0.5, 0.0, 0.0], [2.05, 1.5, 0.6]), {}),      # minimum outside the box; reflection at ub
    (1, [0.2, 0.5, 1.0, 1.1, 1.6, 1.7], ([0, 0, 0, 0, 0, 0], [1, 1, 1.02, 1.2, 1.65, 2]), {}),
    (0, [1.3, 0.7, 0.8], None, dict(maxfev=37)),                       # evaluation budget hit mid-iteration
    (0, [1.3, 0.7, 0.8], None, dict(maxiter=25)),
    (0, [1.3, 0.7, 0.8, 1.9, 1.2, 0.5], None, dict(maxfev=11, maxiter=500)),
    (0, [1.3, 0.7, 0.8], None, dict(maxfev=3)),                        # budget smaller than the initial simplex
]


@pytest.mark.parametrize("case", range(len(NM_CASES)))
def test_nelder_mead_matches_scipy_bit_for_bit(ctx, case):
    kind, x0, bounds, opt = NM_CASES[case]
    fun = (rosen, bowl)[kind]
    kw = {}
    if bounds is not None:
        kw["bounds"] = list(zip(*bounds))
    want = scipy.optimize.minimize(fun, np.array(x0, dtype=np.float64), method="Nelder-Mead", options=dict(opt), **kw)
    got = ctx.nelder_mead_selftest(kind, x0, *(bounds or (None, None)), xatol=opt.get("xatol", 1e-4),
                                   fatol=opt.get("fatol", 1e-4), maxiter=opt.get("maxiter", 0),
                                   maxfev=opt.get("maxfev", 0))
    assert got[1] == want.nfev and got[2] == want.nit, (got, want)
    assert np.array_equal(got[3:], want.x), (got[3:], want.x)
    assert got[0] == want.fun


# ------------------------------------------------------------------ preparation and objective
def test_prepare_pattern(ctx, g):
    keep = ko.circular_window((60, 60)).astype(bool)
    om = g["om_detector_to_sample"]
    ctx.refine_set_patterns(g["patterns"].reshape(-1, 60, 60), None, False, om)
    pat, sqn = ctx.refine_get_prepared()
    # float32 mean of 3600 values around 128: the reference's pairwise float32 sum and the engine's
    # f64 sum differ by a few float32 ulp of the mean
    assert np.allclose(pat[0], g["prep_u8"], rtol=0, atol=1e-4)
    assert np.isclose(sqn[0], g["prep_u8_sqnorm"], rtol=1e-6)
    f32 = (g["patterns"][1].astype(np.float32) * 0.37).reshape(1, 60, 60)
    ctx.refine_set_patterns(f32, ~keep, True, om)
    pat, sqn = ctx.refine_get_prepared()
    assert pat.shape == (1, int(keep.sum()))
    assert np.allclose(pat[0], g["prep_f32_masked"], rtol=0, atol=1e-6)
    assert np.isclose(sqn[0], g["prep_f32_masked_sqnorm"], rtol=1e-6)


def test_objective_values(ctx, g):
    from kikuchipy_amd import _lib

    om = g["om_detector_to_sample"]
    ctx.refine_set_patterns(g["patterns"].reshape(-1, 60, 60), None, False, om)
    offs, vals = g["objective_offsets"], g["objective_values"]
    idx = np.repeat(np.arange(4), len(offs))
    x = (np.concatenate([g["eu0"], g["pc0"]], axis=1)[:, None, :] + offs[None]).reshape(-1, 6)
    pc0 = np.repeat(g["pc0"], len(offs), axis=0)
    q0 = np.repeat(np.array([ko.rotation_from_euler(*e) for e in g["eu0"]]), len(offs), axis=0)
    got = np.stack([
        ctx.refine_objective(_lib.REFINE_ORI, idx, x[:, :3], pc0),
        ctx.refine_objective(_lib.REFINE_PC, idx, x[:, 3:], q0),
        ctx.refine_objective(_lib.REFINE_ORI_PC, idx, x),
    ], axis=1).reshape(4, len(offs), 3)
    # float32 simulated values, f64 sums here vs float32 pairwise sums in the reference
    assert np.allclose(got, vals, rtol=0, atol=2e-6), np.abs(got - vals).max()
    keep = ko.circular_window((60, 60)).astype(bool)
    ctx.refine_set_patterns(g["patterns"].reshape(-1, 60, 60), ~keep, False, om)
    got = ctx.refine_objective(_lib.REFINE_ORI_PC, [2], np.concatenate([g["eu0"][2], g["pc0"][2]]))
    assert abs(got[0] - g["objective_masked"]) < 2e-6


# ------------------------------------------------------------------ complete refinements
def report(name, got, want, nvar):
    """got / want rows: score, nfev, x..."""
    ds = np.abs(got[:, 0] - want[:, 0]).max()
    dx = np.abs(got[:, 2:2 + nvar] - want[:, 2:2 + nvar]).max()
    print(f"{name}: |dscore| {ds:.2e}  |dx| {dx:.2e}  nfev {got[:, 1].astype(int)} vs {want[:, 1].astype(int)}")
    return ds, dx


def rows(res):
    """engine result (n, 1, 3 + nvar): fun, nfev, nit, x  ->  reference rows: ncc, nfev, x"""
    r = res[:, 0]
    return np.column_stack([1 - r[:, 0], r[:, 1], r[:, 3:]])


def test_refine_orientation(ctx, g, g115):
    from kikuchipy_amd import _lib

    om = g["om_detector_to_sample"]
    pats = g["patterns"].reshape(-1, 60, 60)
    ctx.refine_set_patterns(pats, None, False, om)
    x0, pc = g["eu0"][:, None, :], g["pc0"][:, None, :]
    got = rows(ctx.refine_solve(_lib.REFINE_ORI, x0, pc))
    ds, dx = report("ori", got, g115["ori_nm"], 3)
    # same simplex path as SciPy unless a comparison between two nearly equal objective values
    # (they differ by ~1e-7 between the float32 and f64 sums) flips; both end within the
    # optimiser's own tolerances (xatol = fatol = 1e-4)
    assert ds < 1e-4 and dx < 5e-4
    assert np.array_equal(got[:, 1], g115["ori_nm"][:, 1])

    tr = np.deg2rad(2.0)
    got = rows(ctx.refine_solve(_lib.REFINE_ORI, x0, pc, x0 - tr, x0 + tr))
    ds, dx = report("ori bounds", got, g115["ori_nm_bounds"], 3)
    assert ds < 1e-4 and dx < 5e-4

    ctx.refine_set_patterns(pats[:2], None, False, om)
    got = rows(ctx.refine_solve(_lib.REFINE_ORI, x0[:2], pc[:2], maxfev=30))
    ds, dx = report("ori maxfev30", got, g115["ori_nm_maxfev30"], 3)
    assert np.array_equal(got[:, 1], [30, 30]) and ds < 1e-4 and dx < 5e-4

    keep = ko.circular_window((60, 60)).astype(bool)
    ctx.refine_set_patterns(pats[:2], ~keep, False, om)
    got = rows(ctx.refine_solve(_lib.REFINE_ORI, x0[:2], pc[:2]))
    ds, dx = report("ori masked", got, g115["ori_nm_masked"], 3)
    assert ds < 1e-4 and dx < 5e-4


def test_refine_pseudo_symmetry_starts(ctx, g, g115):
    from kikuchipy_amd import _lib

    pats = g["patterns"].reshape(-1, 60, 60)[:2]
    ctx.refine_set_patterns(pats, None, False, g["om_detector_to_sample"])
    starts = g["ori_nm_ps_starts"]
    res = ctx.refine_solve(_lib.REFINE_ORI, starts, np.repeat(g["pc0"][:2, None, :], 2, axis=1))
    assert res.shape == (2, 2, 6)
    ncc = 1 - res[:, :, 0]
    best = np.argmax(ncc, axis=1)
    want = g115["ori_nm_ps"]
    # both starts of this fixture (3 degrees apart) fall into the same optimum, so the winner is
    # decided by score differences at the 1e-7 level: compare the run the reference picked
    for i in range(2):
        assert best[i] == want[i, 5] or abs(ncc[i, 0] - ncc[i, 1]) < 1e-5
    pick = res[np.arange(2), want[:, 5].astype(int)]
    assert np.allclose(1 - pick[:, 0], want[:, 0], atol=1e-4) and np.allclose(pick[:, 3:], want[:, 2:5], atol=5e-4)
    assert np.array_equal(pick[:, 1], want[:, 1])


def test_refine_pc_and_orientation_pc(ctx, g, g115):
    from kikuchipy_amd import _lib

    pats = g["patterns"].reshape(-1, 60, 60)[:2]
    ctx.refine_set_patterns(pats, None, False, g["om_detector_to_sample"])
    q0 = np.array([ko.rotation_from_euler(*e) for e in g["eu0"][:2]])[:, None, :]
    got = rows(ctx.refine_solve(_lib.REFINE_PC, g["pc0"][:2, None, :], q0))
    ds, dx = report("pc", got, g115["pc_nm"], 3)
    assert ds < 1e-4 and dx < 5e-4
    x0 = np.concatenate([g["eu0"][:2], g["pc0"][:2]], axis=1)[:, None, :]
    tr = np.deg2rad(2.0)
    tr6 = np.array([tr, tr, tr, 0.02, 0.02, 0.02])
    got = rows(ctx.refine_solve(_lib.REFINE_ORI_PC, x0, None, x0 - tr6, x0 + tr6))
    ds, dx = report("ori_pc bounds", got, g115["ori_pc_nm_bounds"], 6)
    # six variables, ~300 evaluations on a flat valley (PC and orientation trade off): the
    # endpoint is defined to the optimiser's tolerance only
    assert ds < 5e-4
    # the refined score beats the start and the truth is approached
    start = ctx.refine_objective(_lib.REFINE_ORI_PC, [0, 1], x0[:, 0])
    assert np.all(1 - start < got[:, 0])


def test_error_paths(g):
    from kikuchipy_amd import _lib

    with _lib.Context(0) as c:
        with pytest.raises(_lib.KpdiError, match="kpdi_refine_set_patterns"):
            c.refine_solve(_lib.REFINE_ORI, np.zeros((1, 1, 3)), np.zeros((1, 1, 3)))
        c.refine_set_patterns(g["patterns"].reshape(-1, 60, 60), None, False, np.eye(3))
        with pytest.raises(_lib.KpdiError, match="kpdi_set_master_pattern"):
            c.refine_solve(_lib.REFINE_ORI, np.zeros((4, 1, 3)), np.zeros((4, 1, 3)))
        c.set_master_pattern(np.ones((11, 11), np.float32))
        with pytest.raises(_lib.KpdiError, match="4 patterns were set"):
            c.refine_solve(_lib.REFINE_ORI, np.zeros((3, 1, 3)), np.zeros((3, 1, 3)))
        with pytest.raises(_lib.KpdiError, match="lower bounds is greater"):
            c.refine_solve(_lib.REFINE_ORI, np.zeros((4, 1, 3)), np.zeros((4, 1, 3)), np.ones((4, 1, 3)),
                           np.zeros((4, 1, 3)))
        with pytest.raises(_lib.KpdiError, match="out of range"):
            c.refine_objective(_lib.REFINE_ORI_PC, [7], np.zeros((1, 6)))


# ------------------------------------------------------------------ Python interface
@pytest.fixture(scope="module")
def api_inputs(g):
    import kikuchipy_amd as ka
    from kikuchipy_amd.indexing._refinement import rotation_from_euler

    p = load_golden("projection.npz")
    mp = ka.EBSDMasterPattern(np.stack([p["mp_upper"], p["mp_lower"]]), phase_name="ni")
    pats = g["patterns"].reshape(2, 2, 60, 60)
    det = ka.EBSDDetector(shape=(60, 60), pc=g["pc0"].reshape(2, 2, 3), sample_tilt=70)
    rot0 = rotation_from_euler(g["eu0"]).reshape(2, 2, 4)
    return ka.EBSD(pats), det, mp, rot0


def test_refine_orientation_api(api_inputs, g, g115, capsys):
    """`s.refine_orientation(xmap, detector, master_pattern, energy)` as in the reference:
    one PC per map point, uint8 master pattern (rescaled to float32 inside)."""
    from kikuchipy_amd.indexing._refinement import euler_from_rotation

    s, det, mp, rot0 = api_inputs
    # the reference starts from orix's Euler angles of the indexed rotations: same values here
    assert np.allclose(euler_from_rotation(rot0).reshape(-1, 3), g["eu0"], atol=1e-12)
    res = s.refine_orientation(rot0, det, mp, energy=20)
    out = capsys.readouterr().out
    assert "Method: Nelder-Mead (local) from SciPy" in out and "Refining 4 orientation(s):" in out
    assert "Refinement speed:" in out
    want = g115["ori_nm"]
    assert res.shape == (2, 2) and res.size == 4 and res.rotations.shape == (4, 4)
    assert np.allclose(res.scores, want[:, 0], atol=1e-4)
    assert np.allclose(res.euler, want[:, 2:5], atol=5e-4)
    assert np.array_equal(res.num_evals, want[:, 1])
    assert res.scores.mean() > 0.8 and res.pseudo_symmetry_index is None
    # trust region + navigation mask + signal mask
    nav = np.array([[False, True], [False, False]])
    keep = ko.circular_window((60, 60)).astype(bool)
    res2 = s.refine_orientation(rot0, det, mp, trust_region=[2, 2, 2], navigation_mask=nav, signal_mask=~keep,
                                verbose=False)
    assert res2.size == 3 and np.array_equal(res2.is_in_data, ~nav.ravel())
    assert np.allclose(res2.euler, g115["ori_nm"][[0, 2, 3], 2:5], atol=2e-3)
    # the k best rotations of dictionary indexing: only the best is refined
    res3 = s.refine_orientation(np.stack([rot0, rot0[::-1]], axis=2), det, mp, verbose=False,
                                method_kwargs=dict(method="Nelder-Mead", options=dict(maxfev=30)))
    assert np.all(res3.num_evals == 30)


def test_compute_false_equals_compute_true(api_inputs, g115):
    """`compute=False` (indexing/_refinement/_refinement.py:355-437): the deferred result, computed later, is what
    `compute=True` returns; its rows are the reference's (score, number of evaluations, Euler angles)."""
    import kikuchipy_amd.indexing as ki

    s, det, mp, rot0 = api_inputs
    now = s.refine_orientation(rot0, det, mp, energy=20, verbose=False)
    later = s.refine_orientation(rot0, det, mp, energy=20, compute=False, verbose=False)
    rows = later.compute()
    assert rows.shape == (4, 5)
    assert np.array_equal(rows[:, 0], now.scores) and np.array_equal(rows[:, 1], now.num_evals)
    assert np.array_equal(rows[:, 2:5], now.euler)
    assert np.allclose(rows[:, 2:5], g115["ori_nm"][:, 2:5], atol=5e-4)
    res = ki.compute_refine_orientation_results(later, rot0, mp)
    assert np.array_equal(res.rotations, now.rotations)
    d_pc = s.refine_projection_center(rot0, det, mp, compute=False, verbose=False)
    scores, new_det, num_evals = ki.compute_refine_projection_center_results(d_pc, det)
    scores2, new_det2, num_evals2 = s.refine_projection_center(rot0, det, mp, verbose=False)
    assert np.array_equal(scores, scores2) and np.array_equal(new_det.pc, new_det2.pc) and np.array_equal(num_evals, num_evals2)


def test_refine_pseudo_symmetry_api(api_inputs, g):
    """A 'pseudo-symmetry' operator that is a real 20 degree rotation: the indexed
    orientation (index 0) must win everywhere and the bookkeeping must hold."""
    s, det, mp, rot0 = api_inputs
    op = np.array([[np.cos(np.deg2rad(10)), 0, 0, np.sin(np.deg2rad(10))]])
    res = s.refine_orientation(rot0, det, mp, pseudo_symmetry_ops=op, verbose=False)
    assert np.array_equal(res.pseudo_symmetry_index, [0, 0, 0, 0])
    base = s.refine_orientation(rot0, det, mp, verbose=False)
    assert np.array_equal(res.scores, base.scores) and np.array_equal(res.num_evals, base.num_evals)
    # start from the displaced orientations instead: the operator's inverse leads back
    from kikuchipy_amd.indexing._refinement import quaternion_multiply

    inv = op * [1, -1, -1, -1]
    displaced = quaternion_multiply(inv[0], rot0)
    res = s.refine_orientation(displaced, det, mp, pseudo_symmetry_ops=op, verbose=False)
    assert np.array_equal(res.pseudo_symmetry_index, [1, 1, 1, 1])
    assert np.allclose(res.scores, base.scores, atol=1e-3)


def test_refine_pc_and_both_api(api_inputs, g, g115):
    s, det, mp, rot0 = api_inputs
    scores, new_det, num_evals = s.refine_projection_center(rot0, det, mp, verbose=False)
    assert scores.shape == (4,) and new_det.pc.shape == (2, 2, 3) and num_evals.shape == (4,)
    assert np.allclose(scores[:2], g115["pc_nm"][:, 0], atol=1e-4)
    assert np.allclose(new_det.pc_flattened[:2], g115["pc_nm"][:, 2:5], atol=5e-4)
    assert np.array_equal(num_evals[:2], g115["pc_nm"][:, 1])
    assert not np.allclose(new_det.pc, det.pc)
    res, new_det = s.refine_orientation_projection_center(rot0, det, mp, trust_region=[2, 2, 2, 0.02, 0.02, 0.02],
                                                          verbose=False)
    assert np.allclose(res.scores[:2], g115["ori_pc_nm_bounds"][:, 0], atol=5e-4)
    assert new_det.pc.shape == (2, 2, 3) and res.rotations.shape == (4, 4)
    # refining both can only do better than refining the orientation alone from the same start
    only = s.refine_orientation(rot0, det, mp, trust_region=[2, 2, 2], verbose=False)
    assert np.all(res.scores > only.scores - 1e-4)


def test_other_optimisers_run_on_the_host_with_the_device_objective(api_inputs, g, capsys):
    """Every optimiser of the reference other than plain Nelder-Mead (indexing/_refinement/_solvers.py:179-207):
    the SciPy call runs on the host as in the reference, the objective on the device.  Against the oracle's own
    SciPy run of the same method from the same start (`ko.refine_solver`)."""
    s, det, mp, rot0 = api_inputs
    p = load_golden("projection.npz")
    mpu, mpl = ko.refinement_master_pattern(p["mp_upper"], p["mp_lower"])
    pats = g["patterns"].reshape(4, -1)
    base = s.refine_orientation(rot0, det, mp, verbose=False)

    def oracle(i, method_kwargs, bounds=None):
        dc = ko.detector_direction_cosines((60, 60), g["pc0"][i])
        return ko.refine_solver(pats[i], "ori", g["eu0"][i], mpu, mpl, False, bounds=bounds, method_kwargs=method_kwargs,
                                direction_cosines=dc)

    # a local SciPy method
    res = s.refine_orientation(rot0, det, mp, method_kwargs=dict(method="Powell"))
    out = capsys.readouterr().out
    assert "Method: Powell (local) from SciPy" in out and "{'method': 'Powell'}" in out
    for i in range(4):
        want = oracle(i, dict(method="Powell"))
        assert abs(res.scores[i] - want[0]) < 2e-4 and np.abs(res.euler[i] - want[2:5]).max() < 2e-3
        assert abs(int(res.num_evals[i]) - want[1]) <= 0.25 * want[1]
    assert np.all(res.scores > base.scores - 2e-3)
    # with bounds
    tr = np.deg2rad(2)
    res = s.refine_orientation(rot0, det, mp, method_kwargs=dict(method="L-BFGS-B"), trust_region=[2, 2, 2], verbose=False)
    for i in range(4):
        bounds = np.column_stack([g["eu0"][i] - tr, g["eu0"][i] + tr])
        want = oracle(i, dict(method="L-BFGS-B"), bounds=bounds)
        # (finite-difference gradients with SciPy's 1e-8 step on an objective with 1e-7 of float32 noise: where the
        # search ends is noise on both sides - here only: not worse than the oracle's run, and inside the bounds)
        assert res.scores[i] > want[0] - 5e-3
        assert np.all(res.euler[i] >= bounds[:, 0] - 1e-12) and np.all(res.euler[i] <= bounds[:, 1] + 1e-12)
    # a global method inside the trust region, seeded: reaches the optimum the local searches find
    res = s.refine_orientation(rot0, det, mp, method="differential_evolution", trust_region=[1, 1, 1], verbose=False,
                               method_kwargs=dict(seed=1, maxiter=12, popsize=8, tol=1e-6))
    assert np.all(res.scores > base.scores - 2e-3) and np.all(res.num_evals > 100)
    with pytest.raises(ValueError, match="trust region"):
        s.refine_orientation(rot0, det, mp, method="dual_annealing", verbose=False)
    # the PC and the combined refinement go the same way
    scores, new_det, num_evals = s.refine_projection_center(rot0, det, mp, method_kwargs=dict(method="Powell"), verbose=False)
    ref_scores, ref_det, _ = s.refine_projection_center(rot0, det, mp, verbose=False)
    assert np.allclose(scores, ref_scores, atol=2e-3) and np.allclose(new_det.pc, ref_det.pc, atol=5e-3)


def test_float32_patterns_are_rescaled(api_inputs, g):
    """Patterns given as float32 go through the [-1, 1] rescale (_refinement.py:956); NCC is
    invariant to it up to rounding, so the refinement result is the same."""
    import kikuchipy_amd as ka

    s, det, mp, rot0 = api_inputs
    sf = ka.EBSD(s.data.astype(np.float32) * 0.5 + 3)
    a = s.refine_orientation(rot0, det, mp, verbose=False)
    b = sf.refine_orientation(rot0, det, mp, verbose=False)
    assert np.allclose(a.scores, b.scores, atol=1e-4) and np.allclose(a.euler, b.euler, atol=1e-3)


def test_many_patterns_against_the_scipy_loop(ctx):
    """48 synthetic experiments (smooth random master pattern, 40x40 detector, noise, starts up to
    1 degree off): the device's simplex search against the oracle's SciPy loop, pattern by pattern."""
    from kikuchipy_amd import _lib
    from kikuchipy_amd.indexing._refinement import rotation_from_euler

    rng = np.random.default_rng(99)
    f = np.fft.rfft2(rng.standard_normal((201, 201)))
    ky, kx = np.meshgrid(np.fft.fftfreq(201), np.fft.rfftfreq(201), indexing="ij")
    mpd = np.fft.irfft2(f * np.exp(-(kx**2 + ky**2) / (2 * 0.04**2)), s=(201, 201)).astype(np.float32)
    n, shape, pc = 48, (40, 40), np.array([0.45, 0.7, 0.55])
    m = ko.sample_to_detector_matrix(70.0, 0, 0, 0)
    dc = ko.detector_direction_cosines(shape, pc)
    eu = np.column_stack([rng.uniform(0.3, 6, n), rng.uniform(0.3, 2.8, n), rng.uniform(0.3, 6, n)])
    sim = ko.project_patterns(rotation_from_euler(eu), dc, mpd, mpd)
    noisy = sim + 0.3 * sim.std() * rng.standard_normal(sim.shape).astype(np.float32)
    pats = ((noisy - noisy.min()) / (noisy.max() - noisy.min()) * 255).astype(np.uint8).reshape(n, *shape)
    eu0 = eu + np.deg2rad(rng.uniform(-1, 1, eu.shape))
    with _lib.Context(0) as c:
        c.set_master_pattern(mpd)
        c.refine_set_patterns(pats, None, False, m.T)
        res = c.refine_solve(_lib.REFINE_ORI, eu0[:, None, :], np.tile(pc, (n, 1, 1)))[:, 0]
    same_path = 0
    for i in range(n):
        want = ko.refine_solver(pats[i].ravel(), "ori", eu0[i], mpd, mpd, False, direction_cosines=dc)
        assert abs((1 - res[i, 0]) - want[0]) < 2e-4, (i, 1 - res[i, 0], want[0])
        assert np.abs(res[i, 3:6] - np.array(want[2:5])).max() < 2e-3
        same_path += int(res[i, 1] == want[1])
    # The paths part where two objective values differ by less than their float32 noise (the reference
    # sums in float32, the kernel in f64): near convergence that is common on this smooth landscape, so
    # identical evaluation counts are the exception to count, not the rule to demand - what has to
    # agree, and does for every pattern, is the optimum within the optimiser's own tolerances.
    assert same_path >= 0.25 * n, same_path
    assert np.median(np.abs(res[:, 3:6] - eu)) < np.median(np.abs(eu0 - eu)) / 3


def test_refinement_over_a_group_of_devices(api_inputs):
    """`devices=[...]`: the points are independent, every GPU of a group refines a contiguous block of them from a host
    thread of its own (members sharing device 0 here) - the rows are those of the single-device run, bit for bit, for the
    device search, a host optimiser (objective on the device), PC refinement with a navigation mask, and `compute=False`."""
    s, det, mp, rot0 = api_inputs
    one = s.refine_orientation(rot0, det, mp, energy=20, verbose=False)
    grp = s.refine_orientation(rot0, det, mp, energy=20, devices=[0, 0, 0], verbose=False)
    assert np.array_equal(one.scores, grp.scores) and np.array_equal(one.euler, grp.euler)
    assert np.array_equal(one.num_evals, grp.num_evals)
    assert (0, 0, 0) in s._groups and len(s._groups[(0, 0, 0)]) == 3
    powell = dict(method="Powell", options=dict(maxfev=120))
    a = s.refine_orientation(rot0, det, mp, method_kwargs=powell, verbose=False)
    b = s.refine_orientation(rot0, det, mp, method_kwargs=powell, devices=[0, 0], verbose=False)
    assert np.array_equal(a.scores, b.scores) and np.array_equal(a.num_evals, b.num_evals)
    nav = np.array([[False, True], [False, False]])
    sa, da, na = s.refine_projection_center(rot0, det, mp, navigation_mask=nav, verbose=False)
    sb, db, nb = s.refine_projection_center(rot0, det, mp, navigation_mask=nav, devices=[0] * 5, verbose=False)  # more members than points
    assert np.array_equal(sa, sb) and np.array_equal(da.pc, db.pc) and np.array_equal(na, nb)
    later = s.refine_orientation_projection_center(rot0, det, mp, devices=[0, 0], compute=False, verbose=False)
    rows = later.compute()
    res, new_det = s.refine_orientation_projection_center(rot0, det, mp, verbose=False)
    assert np.array_equal(rows[:, 0], res.scores) and np.array_equal(rows[:, -3:], new_det.pc.reshape(-1, 3))


def test_only_points_in_the_crystal_maps_data_are_refined(api_inputs, g115):
    """signals/util/_crystal_map.py:111-161: the points to refine are `xmap.is_in_data`, further reduced by the navigation
    mask; `xmap.rotations` of an orix CrystalMap holds rows for the points in the data only, this package's
    holders store the whole map - either is taken.  Equal to refining those points alone."""
    from types import SimpleNamespace

    s, det, mp, rot0 = api_inputs
    in_data = np.array([True, False, True, True])
    whole = s.refine_orientation(rot0, det, mp, verbose=False,
                                 navigation_mask=~in_data.reshape(2, 2))           # the same points, by mask
    full_rows = rot0.reshape(4, 4).copy()
    full_rows[~in_data] = [1, 0, 0, 0]                                              # (never looked at)
    for rotations in (full_rows, full_rows[in_data]):                               # holder style, orix style
        xmap = SimpleNamespace(rotations=rotations, is_in_data=in_data, shape=(2, 2))
        res = s.refine_orientation(xmap, det, mp, verbose=False)
        assert res.size == 3 and np.array_equal(res.is_in_data, in_data)
        assert np.array_equal(res.scores, whole.scores) and np.array_equal(res.euler, whole.euler)
    # ... combined with a navigation mask
    nav = np.array([[False, False], [True, False]])
    res = s.refine_orientation(SimpleNamespace(rotations=full_rows, is_in_data=in_data, shape=(2, 2)), det, mp,
                               navigation_mask=nav, verbose=False)
    assert res.size == 2 and np.array_equal(res.is_in_data, [True, False, False, True])
    assert np.array_equal(res.scores, whole.scores[[0, 2]])
    # a refined map (rows for its three points) goes straight into the next refinement, as in the reference's tutorial
    scores, new_det, num_evals = s.refine_projection_center(whole, det, mp, verbose=False,
                                                            method_kwargs=dict(method="Nelder-Mead", options=dict(maxfev=40)))
    assert scores.shape == (3,) and new_det.pc.shape == (3, 3) and (scores >= whole.scores - 1e-4).all()
    with pytest.raises(ValueError, match=r"Crystal map shape \(3, 2\) and signal's navigation shape \(2, 2\) must be the same"):
        s.refine_orientation(SimpleNamespace(rotations=rot0, is_in_data=np.ones(6, bool), shape=(3, 2)), det, mp)
    with pytest.raises(ValueError, match="No point is both in the crystal map's data"):
        s.refine_orientation(SimpleNamespace(rotations=full_rows, is_in_data=in_data, shape=(2, 2)), det, mp,
                             navigation_mask=in_data.reshape(2, 2))


def test_in_data_rows_with_several_rotations_per_point(api_inputs):
    """A masked map with keep_n rotations per point whose row count could be read either way (2 points x 2 rotations =
    4 rows = the 4 points of the map): the first axis says which - `size` rows are the points in the data."""
    from types import SimpleNamespace

    s, det, mp, rot0 = api_inputs
    in_data = np.array([False, True, False, True])
    flat = rot0.reshape(4, 4)
    best_then_other = np.stack([flat[in_data], flat[~in_data]], axis=1)           # (2 points, 2 rotations, 4)
    res = s.refine_orientation(SimpleNamespace(rotations=best_then_other, is_in_data=in_data, shape=(2, 2)), det, mp,
                               verbose=False)
    alone = s.refine_orientation(rot0, det, mp, navigation_mask=~in_data.reshape(2, 2), verbose=False)
    assert res.size == 2 and np.array_equal(res.scores, alone.scores) and np.array_equal(res.euler, alone.euler)
