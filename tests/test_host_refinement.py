"""Host logic of the refinement interface (no GPU): rotation conversions,
bounds, optimiser options and messages (indexing/_refinement/_refinement.py of
the reference)."""

import numpy as np
import pytest

from oracle import kpdi_oracle as ko

import kikuchipy_amd as ka
from kikuchipy_amd.indexing import _refinement as rf


def random_quaternions(n, seed=0):
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    q[q[:, 0] < 0] *= -1
    return q


def test_rotation_from_euler_matches_oracle():
    rng = np.random.default_rng(1)
    eu = np.column_stack([rng.uniform(0, 2 * np.pi, 50), rng.uniform(0, np.pi, 50), rng.uniform(0, 2 * np.pi, 50)])
    want = np.array([ko.rotation_from_euler(*e) for e in eu])
    assert np.allclose(rf.rotation_from_euler(eu), want, rtol=0, atol=1e-15)
    assert rf.rotation_from_euler(eu.reshape(5, 10, 3)).shape == (5, 10, 4)


def test_euler_round_trip():
    q = random_quaternions(500)
    eu = rf.euler_from_rotation(q)
    assert eu[:, 0].min() >= 0 and eu[:, 0].max() < 2 * np.pi
    assert eu[:, 1].min() >= 0 and eu[:, 1].max() <= np.pi
    assert eu[:, 2].min() >= 0 and eu[:, 2].max() < 2 * np.pi
    assert np.allclose(rf.rotation_from_euler(eu), q, atol=1e-12)
    # and the other way around
    rng = np.random.default_rng(2)
    eu = np.column_stack([rng.uniform(0, 2 * np.pi, 100), rng.uniform(0.01, np.pi - 0.01, 100),
                          rng.uniform(0, 2 * np.pi, 100)])
    assert np.allclose(rf.euler_from_rotation(rf.rotation_from_euler(eu)), eu, atol=1e-10)


def test_euler_degenerate_cases():
    # Phi = 0 and Phi = pi: phi2 = 0 by convention, the rotation is reproduced
    for eu in ([1.2, 0.0, 0.0], [4.0, np.pi, 0.0], [0.0, 0.0, 0.0]):
        q = rf.rotation_from_euler(np.array(eu))
        back = rf.euler_from_rotation(q)
        assert np.allclose(back, eu, atol=1e-12), (eu, back)
    q = rf.rotation_from_euler(np.array([0.7, 0.0, 0.5]))  # phi1 + phi2 folded into phi1
    assert np.allclose(rf.euler_from_rotation(q), [1.2, 0, 0], atol=1e-12)


def test_quaternion_multiply_composes_rotations():
    p, q = random_quaternions(20, 3), random_quaternions(20, 4)
    v = np.random.default_rng(5).standard_normal((7, 3))
    for a, b in zip(p, q):
        # rotating by b then by a == rotating by a * b (ko.rotate_vector is the reference's rotate_vector)
        want = ko.rotate_vector(a, ko.rotate_vector(b, v))
        got = ko.rotate_vector(rf.quaternion_multiply(a, b), v)
        assert np.allclose(got, want, atol=1e-12)
    assert rf.quaternion_multiply(p[None, :3, :], q[:5, None, :]).shape == (5, 3, 4)


def test_bounds_follow_the_reference():
    """get_bound_constraints (indexing/_refinement/_refinement.py:1178-1242)."""
    x0 = np.array([[[0.01, 3.1, 6.2]]])
    lo, hi = rf._bounds("ori", x0, [2, 2, 2])
    tr, lee = np.deg2rad(2), np.deg2rad(5)
    assert np.allclose(lo, [[[0.01 - tr, 3.1 - tr, 6.2 - tr]]])
    assert np.allclose(hi, [[[0.01 + tr, 3.1 + tr, min(6.2 + tr, 2 * np.pi + lee)]]])
    lo, hi = rf._bounds("ori", np.array([[[0.0, 0.0, 6.36]]]), [10, 10, 10])
    assert np.allclose(lo[0, 0, :2], -lee) and np.isclose(hi[0, 0, 2], 2 * np.pi + lee)
    lo, hi = rf._bounds("pc", np.array([[[0.4, 0.5, 1.99]]]), [0.05, 0.05, 0.05])
    assert np.allclose(lo, [[[0.35, 0.45, 1.94]]]) and np.allclose(hi, [[[0.45, 0.55, 2.0]]])
    tr6 = [1, 1, 1, 0.02, 0.02, 0.02]
    lo, hi = rf._bounds("ori_pc", np.array([[[1.0, 1.0, 1.0, 0.4, 0.5, 0.6]]]), tr6)
    assert np.allclose(hi - lo, 2 * np.array([np.deg2rad(1)] * 3 + [0.02] * 3))
    assert tr6 == [1, 1, 1, 0.02, 0.02, 0.02]  # caller's list untouched
    assert rf._bounds("ori", x0, None) == (None, None)


def test_optimiser_options():
    nm, shown = rf._nelder_mead_options("minimize", None, None, None)
    assert nm == dict(xatol=1e-4, fatol=1e-4, maxiter=None, maxfev=None) and shown == {"method": "Nelder-Mead"}
    nm, shown = rf._nelder_mead_options("MINIMIZE", dict(method="Nelder-Mead", tol=1e-3, options=dict(maxfev=50)), None, None)
    assert nm == dict(xatol=1e-3, fatol=1e-3, maxiter=None, maxfev=50)
    with pytest.raises(ValueError, match="not in the list of supported methods"):
        rf._nelder_mead_options("powell", None, None, None)
    for method in ("ln_neldermead", "basinhopping", "differential_evolution", "dual_annealing", "shgo"):
        with pytest.raises(NotImplementedError, match="not available on the GPU engine"):
            rf._nelder_mead_options(method, None, None, None)
    with pytest.raises(NotImplementedError, match="only 'Nelder-Mead'"):
        rf._nelder_mead_options("minimize", dict(method="BFGS"), None, None)
    with pytest.raises(NotImplementedError, match="option"):
        rf._nelder_mead_options("minimize", dict(options=dict(adaptive=True)), None, None)


def test_info_message():
    msg = rf._info_message("ori", [1, 1, 1], {"method": "Nelder-Mead"}, 2)
    assert msg == ("Refinement information:\n  Method: Nelder-Mead (local) from SciPy\n  Trust region (+/-): [1 1 1]\n"
                   "  Keyword arguments passed to method: {'method': 'Nelder-Mead'}\n  No. pseudo-symmetry operators: 2")
    assert "Trust region (+/-): None" in rf._info_message("pc", None, {}, 0)


def test_master_pattern_data_is_float32():
    up = np.arange(25, dtype=np.uint8).reshape(5, 5)
    mp = ka.EBSDMasterPattern(up)
    a, b = rf._master_pattern_data(mp, None)
    want = ko.refinement_master_pattern(up, up)[0]
    assert a.dtype == np.float32 and np.array_equal(a, want) and a.min() == -1 and a.max() == 1
    f = np.linspace(0, 1, 25, dtype=np.float32).reshape(5, 5)
    assert np.array_equal(rf._master_pattern_data(ka.EBSDMasterPattern(f), None)[0], f)


def test_argument_validation():
    det = ka.EBSDDetector(shape=(6, 6))
    mp = ka.EBSDMasterPattern(np.zeros((11, 11), np.float32))
    pats = np.zeros((2, 3, 6, 6), np.uint8)
    rot = np.tile([1.0, 0, 0, 0], (2, 3, 1))
    with pytest.raises(ValueError, match="Detector shape"):
        rf.refine("ori", pats, rot, ka.EBSDDetector(shape=(5, 6)), mp)
    with pytest.raises(ValueError, match="exactly one projection center"):
        rf.refine("ori", pats, rot, ka.EBSDDetector(shape=(6, 6), pc=np.full((4, 3), 0.5)), mp)
    with pytest.raises(ValueError, match="Signal mask shape"):
        rf.refine("ori", pats, rot, det, mp, signal_mask=np.zeros((5, 5), bool))
    with pytest.raises(ValueError, match="one rotation per pattern"):
        rf.refine("ori", pats, rot[:1], det, mp)
    with pytest.raises(ValueError, match="Navigation mask shape"):
        rf.refine("ori", pats, rot, det, mp, navigation_mask=np.zeros((3, 2), bool))
    with pytest.raises(ValueError, match="at least one pattern"):
        rf.refine("ori", pats, rot, det, mp, navigation_mask=np.ones((2, 3), bool))
    with pytest.raises(NotImplementedError, match="square Lambert"):
        rf.refine("ori", pats, rot, det, ka.EBSDMasterPattern(np.zeros((11, 11)), projection="stereographic"))
    with pytest.raises(ValueError, match="mode must be"):
        rf.refine("both", pats, rot, det, mp)


class _SolveRecorder:
    """Stands in for the engine context of `refine`: records the calls, returns a fixed solve result."""

    def __init__(self):
        self.calls = []

    def set_master_pattern(self, up, lo):
        self.calls.append("master")

    def refine_set_patterns(self, pats, signal_mask, rescale, om):
        self.calls.append(("patterns", len(pats)))

    def refine_solve(self, mode, x0, fixed, lower, upper, xatol, fatol, maxiter, maxfev):
        self.calls.append(("solve", mode))
        out = np.zeros(x0.shape[:2] + (3 + x0.shape[2],))
        out[:, :, 0] = 0.25 + 0.01 * np.arange(x0.shape[1])   # 1 - ncc: the first start wins
        out[:, :, 1] = 77
        out[:, :, 3:] = x0 + 0.001
        return out


def test_compute_false_defers_the_solve(capsys):
    """compute=False (indexing/_refinement/_refinement.py:355, :429-437, :58-290): validation, set-up and the
    information message happen at the call; nothing runs until `.compute()` / `compute_refine_*_results`, which
    return the reference's result rows / what compute=True returns."""
    import kikuchipy_amd.indexing as ki

    det = ka.EBSDDetector(shape=(6, 6), pc=(0.4, 0.6, 0.5))
    mp = ka.EBSDMasterPattern(np.random.default_rng(0).random((11, 11), dtype=np.float32))
    pats = np.random.default_rng(1).integers(0, 255, (2, 3, 6, 6), dtype=np.uint8)
    rot = np.tile([1.0, 0, 0, 0], (2, 3, 1))
    nav = np.zeros((2, 3), bool)
    nav[0, 1] = True
    ctx = _SolveRecorder()
    s = ka.EBSD(pats)
    s._ctx = ctx  # (the engine context the signal would create lazily)
    with pytest.raises(ValueError, match="Signal mask shape"):   # validation is NOT deferred
        s.refine_orientation(rot, det, mp, signal_mask=np.zeros((5, 5), bool), compute=False)
    d = s.refine_orientation(rot, det, mp, navigation_mask=nav, trust_region=[1, 1, 1], compute=False)
    assert isinstance(d, ki.DeferredRefinement) and ctx.calls == [] and "not computed" in repr(d)
    out = capsys.readouterr().out
    assert "Refinement information:" in out and "Refining" not in out
    rows = d.compute()
    assert ctx.calls == ["master", ("patterns", 5), ("solve", 0)]
    assert rows.shape == (5, 5) and np.allclose(rows[:, 0], 0.75) and np.all(rows[:, 1] == 77)
    out = capsys.readouterr().out
    assert "Refining 5 orientation(s):" in out and "Refinement speed:" in out
    res = ki.compute_refine_orientation_results(d, rot, mp, nav)
    assert ctx.calls == ["master", ("patterns", 5), ("solve", 0)]          # computed once
    assert isinstance(res, ki.RefinementResult) and res.scores.shape == (5,) and np.array_equal(res.is_in_data, ~nav.ravel())
    assert np.allclose(res.euler, rows[:, 2:5])
    with pytest.raises(ValueError, match="'ori' refinement, not 'pc'"):
        ki.compute_refine_projection_center_results(d, det)
    with pytest.raises(TypeError, match="compute=False"):
        ki.compute_refine_orientation_results(rows)
    # the other two modes; pseudo-symmetry adds the index column
    ops = np.array([[0.0, 1, 0, 0]])
    d2 = s.refine_orientation_projection_center(rot, det, mp, pseudo_symmetry_ops=ops, compute=False, verbose=False)
    rows2 = d2.compute()
    assert rows2.shape == (6, 9) and np.all(rows2[:, -1] == 0)
    res2, det2 = ki.compute_refine_orientation_projection_center_results(d2, det, pseudo_symmetry_checked=True)
    assert det2.pc.shape == (2, 3, 3) and np.all(res2.pseudo_symmetry_index == 0)
    with pytest.raises(ValueError, match="pseudo_symmetry_checked"):
        ki.compute_refine_orientation_projection_center_results(d2, det, pseudo_symmetry_checked=False)
    d3 = s.refine_projection_center(rot, det, mp, compute=False, verbose=False)
    scores, det3, nev = ki.compute_refine_projection_center_results(d3, det)
    assert scores.shape == (6,) and np.all(nev == 77) and np.allclose(det3.pc, 0.001 + np.asarray([0.4, 0.6, 0.5]))


# ------------------------------------------------------------------ optimisers driven from the host
class _QuadraticContext:
    """Stands in for the device objective: f(x) = sum(w (x - c_i)^2) + 0.1 for pattern i."""

    def __init__(self, centres, weights):
        self.centres, self.weights, self.calls = np.asarray(centres, float), np.asarray(weights, float), 0

    def refine_objective(self, mode, pattern_index, x, fixed=None):
        self.calls += len(pattern_index)
        x = np.asarray(x, float)
        return np.array([np.sum(self.weights * (xx - self.centres[i]) ** 2) + 0.1 for i, xx in zip(pattern_index, x)])


@pytest.mark.parametrize("method,kwargs", [
    ("minimize", dict(method="Powell")),
    ("minimize", dict(method="L-BFGS-B")),
    ("minimize", dict(method="Nelder-Mead", options=dict(adaptive=True))),   # an option the device search lacks
    ("differential_evolution", dict(seed=3, maxiter=15, tol=1e-8)),
    ("dual_annealing", dict(seed=3, maxiter=40)),
    ("shgo", dict()),
    ("basinhopping", dict(seed=3, niter=3, minimizer_kwargs=dict(method="Powell"))),
])
def test_host_driven_optimisers_call_scipy_like_the_reference(method, kwargs):
    """indexing/_refinement/_solvers.py:179-207: minimize(fun, x0, bounds=..., **kw), <global>(func, bounds=...,
    **kw), basinhopping(func, x0, **kw) - same results as calling SciPy directly on the same objective."""
    import scipy.optimize

    centres = np.array([[0.3, -0.2, 0.5], [1.0, 0.4, -0.7]])
    ctx = _QuadraticContext(centres, [1.0, 2.0, 0.5])
    nm, host, plan = rf._optimization_plan(method, kwargs, None, 1e-4, None, "ori")
    assert nm is None and host is not None and plan["package"] == "scipy"
    x0 = np.zeros((2, 1, 3))
    lower, upper = x0 - 2.0, x0 + 2.0
    res = rf._host_solve(ctx, 0, host, x0, np.zeros((2, 1, 3)), lower, upper)
    assert res.shape == (2, 1, 6) and ctx.calls > 0
    for i in range(2):
        fun = lambda x, i=i: float(np.sum(np.array([1.0, 2.0, 0.5]) * (np.asarray(x) - centres[i]) ** 2) + 0.1)  # noqa: E731
        bounds = list(zip(lower[i, 0], upper[i, 0]))
        solver = getattr(scipy.optimize, method)
        if method == "minimize":
            want = solver(fun, x0[i, 0], bounds=bounds, **kwargs)
        elif method == "basinhopping":
            want = solver(fun, x0[i, 0], **kwargs)
        else:
            want = solver(fun, bounds=bounds, **kwargs)
        assert res[i, 0, 0] == want.fun and res[i, 0, 1] == want.nfev and np.array_equal(res[i, 0, 3:], want.x)
        assert np.abs(res[i, 0, 3:] - centres[i]).max() < 2e-2


def test_a_population_is_one_objective_call():
    """differential_evolution(vectorized=True): SciPy hands the objective a whole generation, (variables, S); the
    device evaluates it in ONE kpdi_refine_objective launch (S evaluations of the same pattern)."""
    import scipy.optimize

    centres = np.array([[0.3, -0.2, 0.5]])
    w = np.array([1.0, 2.0, 0.5])

    class Ctx(_QuadraticContext):
        launches = 0

        def refine_objective(self, mode, pattern_index, x, fixed=None):
            Ctx.launches += 1
            return super().refine_objective(mode, pattern_index, x, fixed)

    ctx = Ctx(centres, w)
    kwargs = dict(seed=3, maxiter=12, tol=1e-8, vectorized=True, updating="deferred", polish=False)
    nm, host, plan = rf._optimization_plan("differential_evolution", kwargs, None, 1e-4, None, "ori")
    x0 = np.zeros((1, 1, 3))
    res = rf._host_solve(ctx, 0, host, x0, np.zeros((1, 1, 3)), x0 - 2.0, x0 + 2.0)
    fun = lambda x: np.sum(w[:, None] * (np.asarray(x) - centres[0][:, None]) ** 2, axis=0) + 0.1  # noqa: E731
    want = scipy.optimize.differential_evolution(fun, bounds=[(-2.0, 2.0)] * 3, **kwargs)
    assert res[0, 0, 0] == want.fun and res[0, 0, 1] == want.nfev and np.array_equal(res[0, 0, 3:], want.x)
    # one launch per generation (SciPy counts a vectorised call as one evaluation), 45 points each
    assert Ctx.launches == want.nfev == 13 and ctx.calls == 13 * 45


def test_global_methods_need_a_trust_region_and_messages():
    nm, host, plan = rf._optimization_plan("differential_evolution", None, None, 1e-4, None, "pc")
    with pytest.raises(ValueError, match="trust region"):
        rf._host_solve(_QuadraticContext([[0, 0, 0]], [1, 1, 1]), 1, host, np.zeros((1, 1, 3)), np.zeros((1, 1, 4)), None, None)
    msg = rf._info_message("pc", [0.1, 0.1, 0.1], plan["kwargs"], 0, plan)
    assert msg == ("Refinement information:\n  Method: differential_evolution (global) from SciPy\n"
                   "  Trust region (+/-): [0.1 0.1 0.1]\n  Keyword arguments passed to method: {}")
    nm, host, plan = rf._optimization_plan("basinhopping", None, None, 1e-4, None, "ori")
    msg = rf._info_message("ori", None, plan["kwargs"], 0, plan)
    assert "Method: basinhopping (global) from SciPy" in msg and "Trust region" not in msg
    assert "minimizer_kwargs" in msg  # added like the reference does (_refinement.py:1132-1133)
    nm, host, plan = rf._optimization_plan("minimize", dict(method="Powell"), None, 1e-4, None, "ori")
    assert "Method: Powell (local) from SciPy" in rf._info_message("ori", [1, 1, 1], plan["kwargs"], 0, plan)
    # the plain Nelder-Mead stays on the device
    nm, host, plan = rf._optimization_plan(None, None, None, 1e-4, None, "ori")
    assert host is None and nm["xatol"] == 1e-4


def test_nlopt_is_driven_like_the_reference(monkeypatch):
    """LN_NELDERMEAD (indexing/_refinement/_solvers.py:464-536, _refinement.py:1098-1124): NLopt is not installable
    in the build image, so the calls are checked against a stand-in module that records them."""
    import sys
    import types

    with pytest.raises(ImportError, match="nlopt"):
        rf._optimization_plan("ln_neldermead", None, None, 1e-4, None, "ori")

    log = []

    class Opt:
        def __init__(self, name, n):
            log.append(("opt", name, n))
            self.n, self.evals = n, 0

        def set_ftol_rel(self, v): log.append(("ftol_rel", v))
        def set_initial_step(self, v): log.append(("initial_step", list(v)))
        def set_maxeval(self, v): log.append(("maxeval", v))
        def set_lower_bounds(self, v): log.append(("lower", list(v)))
        def set_upper_bounds(self, v): log.append(("upper", list(v)))
        def set_min_objective(self, f): self.f = f

        def optimize(self, x0):  # a few coordinate steps are enough to see the objective being called
            x, best = np.array(x0, float), None
            for _ in range(20):
                for d in range(self.n):
                    for step in (0.05, -0.05):
                        y = x.copy()
                        y[d] += step
                        self.evals += 1
                        fy = self.f(y, None)
                        if best is None or fy < best:
                            best, x = fy, y
            self.best = best
            return x

        def last_optimum_value(self): return self.best
        def get_numevals(self): return self.evals

    monkeypatch.setitem(sys.modules, "nlopt", types.SimpleNamespace(opt=Opt))
    nm, host, plan = rf._optimization_plan("LN_NELDERMEAD", None, [0.5, 0.01], 1e-3, 250, "ori_pc")
    assert host.method_name == "LN_NELDERMEAD" and plan["package"] == "nlopt"
    centres = np.array([[0.2, 0.1, -0.1, 0.4, 0.5, 0.6]])
    ctx = _QuadraticContext(centres, np.ones(6))
    x0 = np.zeros((1, 1, 6))
    res = rf._host_solve(ctx, 2, host, x0, None, x0 - 1, x0 + 1)
    assert ("opt", "LN_NELDERMEAD", 6) in log and ("ftol_rel", 1e-3) in log and ("maxeval", 250) in log
    assert ("initial_step", [0.5, 0.5, 0.5, 0.01, 0.01, 0.01]) in log and ("lower", [-1.0] * 6) in log
    assert res[0, 0, 1] == 240 and res[0, 0, 0] < 0.1 + np.sum(centres**2)
    msg = rf._info_message("ori_pc", [1] * 6, plan["kwargs"], 1, plan)
    assert "Method: LN_NELDERMEAD (local) from NLopt" in msg and "Relative tolerance: 0.001" in msg
    assert "Initial step(s): [0.5, 0.5, 0.5, 0.01, 0.01, 0.01]" in msg and "Max. function evaulations: 250" in msg
    with pytest.raises(ValueError, match="initial step"):
        rf._optimization_plan("ln_neldermead", None, [1, 2, 3], 1e-4, None, "ori")
