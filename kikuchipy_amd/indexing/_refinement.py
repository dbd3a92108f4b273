"""Refinement of orientations and/or projection centres on the GPU engine.

Counterpart of `EBSD.refine_orientation`, `EBSD.refine_projection_center` and
`EBSD.refine_orientation_projection_center` (signals/ebsd.py:1986-2700 of the
reference) with their set-up (`_RefinementSetup`,
indexing/_refinement/_refinement.py:851-1286) for the default optimiser,
`scipy.optimize.minimize(method="Nelder-Mead")`.  The reference hands every
pattern to SciPy, which calls a Numba objective a few hundred times; here the
whole map is ONE kernel launch (`kpdi_refine_solve`): a workgroup per (pattern,
start) evaluates the objective and walks SciPy's simplex on the device.

Every other optimiser the reference can dispatch to (SciPy's local methods and their options, its global
methods, NLopt's LN_NELDERMEAD) keeps ITS optimiser on the host, called with the reference's arguments, with
the objective evaluated on the device (`_HostOptimizer`); `compute=False` returns a `DeferredRefinement`
in place of the reference's lazy Dask array.
"""

import time

import numpy as np

from kikuchipy_amd import _lib
from kikuchipy_amd.indexing._dictionary_indexing import MapData

MODES = {"ori": _lib.REFINE_ORI, "pc": _lib.REFINE_PC, "ori_pc": _lib.REFINE_ORI_PC}
# indexing/_refinement/__init__.py:33-66
SUPPORTED_OPTIMIZATION_METHODS = ["minimize", "ln_neldermead", "basinhopping", "differential_evolution",
                                  "dual_annealing", "shgo"]


# --------------------------------------------------------------------------- rotations
def rotation_from_euler(euler):
    """(..., 3) Bunge Euler angles in radians -> (..., 4) unit quaternions with a
    non-negative scalar part (_utils/numba.py:43-57; `Rotation.from_euler`)."""
    euler = np.asarray(euler, dtype=np.float64)
    sigma = 0.5 * (euler[..., 0] + euler[..., 2])
    delta = 0.5 * (euler[..., 0] - euler[..., 2])
    c, s = np.cos(0.5 * euler[..., 1]), np.sin(0.5 * euler[..., 1])
    q = np.stack([c * np.cos(sigma), -s * np.cos(delta), -s * np.sin(delta), -c * np.sin(sigma)], axis=-1)
    return np.where(q[..., :1] < 0, -q, q)


def euler_from_rotation(q):
    """(..., 4) unit quaternions -> (..., 3) Bunge Euler angles, phi1 and phi2 in
    [0, 2 pi), Phi in [0, pi]: the inverse of `rotation_from_euler`, i.e. what
    orix's `Rotation.to_euler()` returns (un-vendored third party; eq. A.6 of
    Rowenhorst et al. 2015 with orix's passive convention).  phi2 = 0 where Phi
    is 0 or pi."""
    q = np.asarray(q, dtype=np.float64)
    a, b, c, d = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    q03, q12 = a * a + d * d, b * b + c * c
    chi = np.sqrt(q03 * q12)
    sigma = np.arctan2(-d, a)   # (phi1 + phi2) / 2
    delta = np.arctan2(-c, -b)  # (phi1 - phi2) / 2
    phi1 = np.where(chi == 0, np.where(q12 == 0, 2 * sigma, 2 * delta), sigma + delta)
    Phi = np.where(chi == 0, np.where(q12 == 0, 0.0, np.pi), 2 * np.arctan2(np.sqrt(q12), np.sqrt(q03)))
    phi2 = np.where(chi == 0, 0.0, sigma - delta)
    two_pi = 2 * np.pi
    out = np.stack([np.mod(phi1, two_pi), Phi, np.mod(phi2, two_pi)], axis=-1)
    out[out == two_pi] = 0.0  # np.mod of a tiny negative number
    return out


def quaternion_multiply(p, q):
    """Hamilton product p * q, broadcasting (orix `Quaternion.__mul__`)."""
    p, q = np.asarray(p, dtype=np.float64), np.asarray(q, dtype=np.float64)
    a1, b1, c1, d1 = p[..., 0], p[..., 1], p[..., 2], p[..., 3]
    a2, b2, c2, d2 = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    return np.stack([
        a1 * a2 - b1 * b2 - c1 * c2 - d1 * d2,
        a1 * b2 + b1 * a2 + c1 * d2 - d1 * c2,
        a1 * c2 - b1 * d2 + c1 * a2 + d1 * b2,
        a1 * d2 + b1 * c2 - c1 * b2 + d1 * a2,
    ], axis=-1)


# --------------------------------------------------------------------------- results
class RefinementResult(MapData):
    """What the reference puts into the refined `CrystalMap`
    (indexing/_refinement/_refinement.py:58-131): per refined point the
    `scores` (NCC), `num_evals`, `rotations` (quaternions from the refined Euler
    angles) and, with pseudo-symmetry operators, `pseudo_symmetry_index`
    (0 = the indexed orientation itself).  `is_in_data` marks the refined points
    in the flattened map."""

    def __init__(self, scores, num_evals, euler, is_in_data, nav_shape, pseudo_symmetry_index=None,
                 patterns_per_second=None):
        self.scores = scores
        self.num_evals = num_evals
        self.euler = euler
        self.rotations = None if euler is None else rotation_from_euler(euler)
        self.is_in_data = is_in_data
        self.shape = tuple(nav_shape)
        self.pseudo_symmetry_index = pseudo_symmetry_index
        self.patterns_per_second = patterns_per_second

    @property
    def size(self):
        return int(np.count_nonzero(self.is_in_data))

    @property
    def prop(self):
        out = {"scores": self.scores, "num_evals": self.num_evals}
        if self.pseudo_symmetry_index is not None:
            out["pseudo_symmetry_index"] = self.pseudo_symmetry_index
        return out

    def to_crystal_map(self, phase_list=None, step_sizes=None):
        """The refined `orix.crystal_map.CrystalMap` as the reference assembles it
        (indexing/_refinement/_refinement.py:104-131; needs orix, which is not a
        dependency of this package)."""
        from orix.crystal_map import CrystalMap, create_coordinate_arrays
        from orix.quaternion import Rotation

        if self.rotations is None:
            raise ValueError("a projection-centre refinement has no rotations")
        kw, _ = create_coordinate_arrays(self.shape, step_sizes)
        n_all = self.is_in_data.size
        rot = np.zeros((n_all, 4))
        rot[:, 0] = 1
        rot[self.is_in_data] = self.rotations
        prop = {}
        for name, values in self.prop.items():
            full = np.zeros(n_all, dtype=np.asarray(values).dtype)
            full[self.is_in_data] = values
            prop[name] = full
        return CrystalMap(rotations=Rotation(rot), phase_list=phase_list, prop=prop, is_in_data=self.is_in_data, **kw)


# --------------------------------------------------------------------------- set-up
# method -> (type, supports bounds, package), `SUPPORTED_OPTIMIZATION_METHODS` of indexing/_refinement/__init__.py:32-70
_METHOD_INFO = {
    "minimize": ("local", True, "scipy"),
    "ln_neldermead": ("local", True, "nlopt"),
    "basinhopping": ("global", False, "scipy"),
    "differential_evolution": ("global", True, "scipy"),
    "dual_annealing": ("global", True, "scipy"),
    "shgo": ("global", True, "scipy"),
}


def _nelder_mead_options(method, method_kwargs, initial_step, maxeval):
    """`_RefinementSetup.set_optimization_parameters`
    (indexing/_refinement/_refinement.py:1053-1139) for the method that runs ENTIRELY on the
    device: SciPy's Nelder-Mead with its plain options.  Raises NotImplementedError for
    everything else - `_optimization_plan` then drives the optimiser on the host."""
    method = (method or "minimize").lower()
    if method not in SUPPORTED_OPTIMIZATION_METHODS:
        raise ValueError(
            f"Method {method!r} not in the list of supported methods {SUPPORTED_OPTIMIZATION_METHODS}"
        )
    if method != "minimize":
        raise NotImplementedError(
            f"Method {method!r} is not available on the GPU engine; only 'minimize' with SciPy's "
            "Nelder-Mead (the reference's default) is"
        )
    kwargs = dict(method_kwargs or {})
    name = kwargs.pop("method", "Nelder-Mead")
    if str(name).lower() not in ("nelder-mead", "neldermead"):
        raise NotImplementedError(f"minimize(method={name!r}) is not available on the GPU engine, only 'Nelder-Mead'")
    options = dict(kwargs.pop("options", None) or {})
    tol = kwargs.pop("tol", None)
    if kwargs:
        raise NotImplementedError(f"unsupported keyword argument(s) to minimize: {sorted(kwargs)}")
    if tol is not None:  # scipy.optimize.minimize: tol sets xatol and fatol of Nelder-Mead
        options.setdefault("xatol", tol)
        options.setdefault("fatol", tol)
    unknown = set(options) - {"xatol", "fatol", "maxiter", "maxfev", "disp", "adaptive", "return_all"}
    if unknown or options.get("adaptive") or options.get("return_all"):
        raise NotImplementedError(f"unsupported Nelder-Mead option(s): {sorted(unknown) or ['adaptive/return_all']}")
    shown = {"method": "Nelder-Mead"}
    if method_kwargs:
        shown.update({k: v for k, v in method_kwargs.items() if k != "method"})
    return dict(xatol=float(options.get("xatol", 1e-4)), fatol=float(options.get("fatol", 1e-4)),
                maxiter=options.get("maxiter"), maxfev=options.get("maxfev")), shown


class _HostOptimizer:
    """Every optimiser of the reference other than plain Nelder-Mead (SciPy local methods with their
    options, the SciPy global methods, NLopt's LN_NELDERMEAD): the OPTIMISER runs on the host exactly as
    in the reference (indexing/_refinement/_solvers.py:79-250, :464-600: same call, same keyword
    arguments), the OBJECTIVE - master-pattern projection + NCC of one pattern - is one call into the
    device per evaluation (`kpdi_refine_objective`).  Slow next to the on-device simplex search (one
    launch and one synchronisation per evaluation) but complete."""

    def __init__(self, method, method_kwargs, initial_step, rtol, maxeval, mode):
        self.method = method
        self.type, self.supports_bounds, self.package = _METHOD_INFO[method]
        self.mode = mode
        self.rtol, self.maxeval, self.initial_step = rtol, maxeval, None
        self.kwargs = dict(method_kwargs or {})
        if self.package == "nlopt":
            try:
                import nlopt  # noqa: F401
            except ImportError as err:  # verify_dependency_or_raise("nlopt", ...) in the reference
                raise ImportError(f"Optimization method {method.upper()!r} requires the optional dependency 'nlopt'") from err
            self.method_name = method.upper()
            if initial_step is not None:
                step = np.atleast_1d(initial_step)
                if step.size != {"ori": 1, "pc": 1, "ori_pc": 2}[mode]:
                    raise ValueError("The initial step must be a single number when refining orientations or PCs and a "
                                     "list of two numbers when refining both")
                self.initial_step = [float(v) for v in np.repeat(step, 3)]
        else:
            if method == "minimize" and "method" not in self.kwargs:
                self.kwargs["method"] = "Nelder-Mead"
            self.method_name = self.kwargs.get("method", method) if method == "minimize" else method
            if method == "basinhopping":
                self.kwargs.setdefault("minimizer_kwargs", {})

    @property
    def shown_kwargs(self):
        return self.kwargs

    def run(self, fun, x0, bounds):
        """One optimisation; returns (fun, nfev, nit, x)."""
        if self.package == "nlopt":
            import nlopt

            opt = nlopt.opt(self.method_name, len(x0))
            opt.set_ftol_rel(self.rtol)
            if self.initial_step is not None:
                opt.set_initial_step(self.initial_step)
            if self.maxeval is not None:
                opt.set_maxeval(self.maxeval)
            opt.set_min_objective(lambda x, grad: fun(x))
            if bounds is not None:
                opt.set_lower_bounds([b[0] for b in bounds])
                opt.set_upper_bounds([b[1] for b in bounds])
            x = opt.optimize(np.asarray(x0, dtype=np.float64))
            return opt.last_optimum_value(), opt.get_numevals(), 0, np.asarray(x)
        import scipy.optimize

        solver = getattr(scipy.optimize, self.method)
        kwargs = dict(self.kwargs)
        if self.method == "minimize":
            if bounds is not None:
                kwargs["bounds"] = bounds
            res = solver(fun=fun, x0=x0, **kwargs)
        elif self.supports_bounds:
            if bounds is None:
                raise ValueError(f"Method {self.method!r} optimises within bounds: pass a trust region")
            res = solver(func=fun, bounds=bounds, **kwargs)
        else:  # basinhopping
            kwargs["minimizer_kwargs"] = dict(kwargs["minimizer_kwargs"])
            res = solver(func=fun, x0=x0, **kwargs)
        return float(res.fun), int(res.nfev), int(getattr(res, "nit", 0) or 0), np.asarray(res.x, dtype=np.float64)


def _optimization_plan(method, method_kwargs, initial_step, rtol, maxeval, mode):
    """(device options | None, host optimiser | None, what the info message shows)."""
    try:
        nm, shown = _nelder_mead_options(method, method_kwargs, initial_step, maxeval)
        return nm, None, dict(method_name="Nelder-Mead", type="local", package="scipy", supports_bounds=True, kwargs=shown)
    except NotImplementedError:
        host = _HostOptimizer((method or "minimize").lower(), method_kwargs, initial_step, rtol, maxeval, mode)
        return None, host, dict(method_name=host.method_name, type=host.type, package=host.package,
                                supports_bounds=host.supports_bounds, kwargs=host.shown_kwargs, host=host)


def _host_solve(ctx, mode_code, host, x0, fixed, lower, upper):
    """The host-driven counterpart of `Context.refine_solve`: same (n, starts, 3 + nvar) result."""
    n, starts, nvar = x0.shape
    res = np.empty((n, starts, 3 + nvar))
    for i in range(n):
        for s in range(starts):
            fx = None if fixed is None else fixed[i, s][None]

            def fun(x, _i=i, _fx=fx):
                x = np.asarray(x, dtype=np.float64)
                if x.ndim == 2:
                    # a whole population at once, (variables, S) -> (S,): ONE launch per generation
                    # (differential_evolution(vectorized=True, updating="deferred"); an extension - the reference's
                    # objective takes one point)
                    pop = np.ascontiguousarray(x.T)
                    fxs = None if _fx is None else np.repeat(_fx, len(pop), axis=0)
                    return ctx.refine_objective(mode_code, np.full(len(pop), _i), pop, fxs)
                return float(ctx.refine_objective(mode_code, [_i], x[None], _fx)[0])

            bounds = None if lower is None else list(zip(lower[i, s], upper[i, s]))
            f, nfev, nit, x = host.run(fun, x0[i, s], bounds)
            res[i, s, 0], res[i, s, 1], res[i, s, 2], res[i, s, 3:] = f, nfev, nit, x
    return res


def _bounds(mode, x0, trust_region):
    """`_RefinementSetup.get_bound_constraints` (:1178-1242)."""
    if trust_region is None:
        return None, None
    angle_leeway = np.deg2rad(5)
    eu_lower = 3 * [-angle_leeway]
    eu_upper = [2 * np.pi + angle_leeway, np.pi + angle_leeway, 2 * np.pi + angle_leeway]
    pc_lower, pc_upper = 3 * [-2], 3 * [2]
    trust_region = np.asarray(trust_region, dtype=np.float64).copy()
    if mode == "ori":
        trust_region = np.deg2rad(trust_region)
        lower_abs, upper_abs = eu_lower, eu_upper
    elif mode == "pc":
        lower_abs, upper_abs = pc_lower, pc_upper
    else:
        trust_region[:3] = np.deg2rad(trust_region[:3])
        lower_abs, upper_abs = eu_lower + pc_lower, eu_upper + pc_upper
    return np.fmax(x0 - trust_region, lower_abs), np.fmin(x0 + trust_region, upper_abs)


def _info_message(mode, trust_region, shown_kwargs, n_pseudo, plan=None):
    """`_RefinementSetup.get_info_message` (:1244-1286)."""
    plan = plan or dict(method_name="Nelder-Mead", type="local", package="scipy", supports_bounds=True)
    package = {"scipy": "SciPy", "nlopt": "NLopt"}[plan["package"]]
    info = f"Refinement information:\n  Method: {plan['method_name']} ({plan['type']}) from {package}"
    if plan["supports_bounds"]:
        tr_str = np.array_str(np.asarray(trust_region), precision=5)
        info += "\n  Trust region (+/-): " + tr_str
    if plan["package"] == "scipy":
        info += f"\n  Keyword arguments passed to method: {shown_kwargs}"
    else:
        host = plan["host"]
        info += f"\n  Relative tolerance: {host.rtol}"
        if host.initial_step:
            info += f"\n  Initial step(s): {host.initial_step}"
        if host.maxeval:
            info += f"\n  Max. function evaulations: {host.maxeval}"
    if n_pseudo > 0:
        info += f"\n  No. pseudo-symmetry operators: {n_pseudo}"
    return info


def _master_pattern_data(master_pattern, energy):
    """`_get_master_pattern_data` (:1288-1320): float32 hemispheres; other data
    types are rescaled to [-1, 1] over each hemisphere's own range
    (pattern/_pattern.py:66-93)."""
    out = []
    for mp in master_pattern._get_master_pattern_arrays_from_energy(energy):
        if mp.dtype != np.float32:
            imin, imax = np.nanmin(mp), np.nanmax(mp)
            mp = ((mp - imin) / float(imax - imin) * 2 + -1).astype(np.float32)
        out.append(np.ascontiguousarray(mp))
    return out


def refine(mode, patterns, rotations, detector, master_pattern, energy=None, navigation_mask=None,
           signal_mask=None, pseudo_symmetry_ops=None, method="minimize", method_kwargs=None, trust_region=None,
           initial_step=None, rtol=1e-4, maxeval=None, context=None, device=0, verbose=True, comm=None, compute=True,
           contexts=None, is_in_data=None, xmap_shape=None):
    """Shared driver of the three refinements.

    is_in_data, xmap_shape
        Of the crystal map the rotations come from: only points that are in ITS data are refined - the navigation mask
        is combined with it (signals/util/_crystal_map.py:111-161) - and its shape must be the signal's navigation shape
        (`_xmap_is_compatible_with_signal`, :28-61).  `rotations` may then hold rows for the points in the data only
        (what orix's `CrystalMap.rotations` gives) or for every point of the map (what this package's holders store).

    contexts
        Several engine contexts (one per GPU - the members of a `kikuchipy_amd._lib.Group`): the points are independent,
        so each context refines a contiguous block of them from a host thread of its own (the library calls release the
        GIL) and the rows are concatenated - the results do not depend on the split.  The single-process counterpart of
        `comm`.

    compute
        False: validate, set everything up, print the information message and return a `DeferredRefinement` -
        what the reference returns as a lazy Dask array (indexing/_refinement/_refinement.py:355, :429-437);
        `.compute()` (or `compute_refine_*_results`) runs it.

    patterns
        (..., rows, cols) experimental patterns with 0-2 navigation axes.
    rotations
        Quaternions of the indexed orientations, navigation shape + (4,) or
        navigation shape + (k, 4) (best match first, as dictionary indexing
        returns them; only the best is refined, _refinement.py:963-966).
    detector
        `EBSDDetector` with one PC or one PC per navigation point.
    comm
        `kikuchipy_amd.parallel.Communicator`: every rank (one process per GPU)
        passes the same arguments, refines its contiguous block of the points and
        all ranks return the complete result.
    Returns `(RefinementResult, new_detector)`; `new_detector` is None in
    mode "ori".
    """
    if mode not in MODES:
        raise ValueError(f"mode must be one of {sorted(MODES)}")
    nm, host, plan = _optimization_plan(method, method_kwargs, initial_step, rtol, maxeval, mode)
    shown = plan["kwargs"]
    master_pattern._is_suitable_for_projection(raise_if_not=True)
    patterns = np.asarray(patterns)
    if patterns.ndim < 2 or patterns.ndim > 4:
        raise ValueError("patterns must have 0, 1 or 2 navigation axes and 2 signal axes")
    nav_shape, sig_shape = patterns.shape[:-2], patterns.shape[-2:]
    nav_size = int(np.prod(nav_shape)) if nav_shape else 1
    if sig_shape != detector.shape:
        raise ValueError(f"Detector shape {detector.shape} must be equal to the signal shape {sig_shape}")
    if detector.navigation_size not in (1, nav_size):
        raise ValueError(
            f"Detector must have exactly one projection center (PC), or one PC per pattern in an array of shape "
            f"signal's navigation shape + (3,) {nav_shape + (3,)}, but was {detector.pc.shape}"
        )
    if signal_mask is not None and signal_mask.shape != sig_shape:
        raise ValueError(
            f"Signal mask shape {signal_mask.shape} and signal's signal shape {sig_shape} must be the same shape"
        )
    if xmap_shape is not None and tuple(xmap_shape) != (nav_shape or (1,)) and tuple(xmap_shape) != nav_shape:
        raise ValueError(
            f"Crystal map shape {tuple(xmap_shape)} and signal's navigation shape {nav_shape} must be the same "
            "(see EBSD.axes_manager)"
        )
    rot = np.asarray(getattr(rotations, "data", rotations), dtype=np.float64)
    if rot.shape[-1] != 4:
        raise ValueError("`rotations` must be quaternions with a last axis of size 4")
    in_data = None
    if is_in_data is not None:
        in_data = np.asarray(is_in_data, dtype=bool).ravel()
        if in_data.size != nav_size:
            raise ValueError(
                f"Crystal map with {in_data.size} points and signal's navigation shape {nav_shape} must be the same "
                "(see EBSD.axes_manager)"
            )
        n_in = int(in_data.sum())
        if 0 < n_in < nav_size and rot.ndim >= 2 and rot.shape[0] == n_in:
            # a crystal map keeps its rotations flat, one row (of k) per point: `size` rows are the points in the data
            # only (orix) - laid out on the whole map here (identity elsewhere, never refined); `nav_size` rows are the
            # whole map (this package's holders)
            rows = rot.reshape(n_in, -1, 4)
            full = np.zeros((nav_size,) + rows.shape[1:])
            full[..., 0] = 1
            full[in_data] = rows
            rot = full
    if rot.size % (nav_size * 4) != 0 or rot.size == 0:
        raise ValueError(f"need one rotation per pattern ({nav_size}), got an array of shape {rot.shape}")
    # several rotations per point (the k best of dictionary indexing): refine the best
    rot = rot.reshape(nav_size, -1, 4)[:, 0]
    if navigation_mask is not None:
        if navigation_mask.shape != (nav_shape or (1,)):
            raise ValueError(
                f"Navigation mask shape {navigation_mask.shape} and crystal map shape {nav_shape} must be the same"
            )
        points = ~np.asarray(navigation_mask, dtype=bool).ravel()
        if not points.any():
            raise ValueError("The navigation mask must allow refinement of at least one pattern")
    else:
        points = np.ones(nav_size, dtype=bool)
    if in_data is not None:
        points = points & in_data
        if not points.any():
            raise ValueError("No point is both in the crystal map's data and allowed by the navigation mask")
    n = int(points.sum())
    rot = rot[points]
    pats = patterns.reshape((nav_size,) + sig_shape)[points]
    pc = detector.pc_flattened[points] if detector.navigation_size > 1 else np.tile(detector.pc_flattened, (n, 1))
    pc = pc.astype(np.float64)

    # starts: the indexed orientation, then its pseudo-symmetry equivalents (_refinement.py:968-973)
    n_pseudo = 0
    if pseudo_symmetry_ops is not None and mode != "pc":
        ops = np.asarray(getattr(pseudo_symmetry_ops, "data", pseudo_symmetry_ops), dtype=np.float64).reshape(-1, 4)
        n_pseudo = ops.shape[0]
        rot_starts = np.concatenate([rot[:, None, :], quaternion_multiply(ops[None, :, :], rot[:, None, :])], axis=1)
    else:
        rot_starts = rot[:, None, :]
    starts = rot_starts.shape[1]
    pc_starts = np.repeat(pc[:, None, :], starts, axis=1)
    if mode == "ori":
        x0, fixed = euler_from_rotation(rot_starts), pc_starts
    elif mode == "pc":
        x0, fixed = pc_starts, rot_starts
    else:
        x0, fixed = np.concatenate([euler_from_rotation(rot_starts), pc_starts], axis=2), None
    lower, upper = _bounds(mode, x0, trust_region)
    if verbose:
        print(_info_message(mode, trust_region, shown, n_pseudo, plan))

    def run():
        return _run_refinement(mode, n, starts, x0, fixed, lower, upper, pats, signal_mask, detector, master_pattern,
                               energy, nm, host, context, device, comm, verbose, n_pseudo, points, nav_shape,
                               navigation_mask, contexts)

    if not compute:
        return DeferredRefinement(mode, run, n_pseudo > 0)
    result, new_detector, _ = run()
    return result, new_detector


def _run_refinement(mode, n, starts, x0, fixed, lower, upper, pats, signal_mask, detector, master_pattern, energy, nm, host,
                    context, device, comm, verbose, n_pseudo, points, nav_shape, navigation_mask, contexts=None):
    """The solve + the assembly of the result (what `compute_refine_*_results` do in the reference,
    indexing/_refinement/_refinement.py:58-290).  Returns (RefinementResult, new detector | None, raw rows)."""
    lo_i, hi_i = 0, n
    if comm is not None and comm.world_size > 1:
        from kikuchipy_amd.parallel import shard_range

        lo_i, hi_i = shard_range(n, comm.rank, comm.world_size)
    part = slice(lo_i, hi_i)
    if verbose:
        what = {"ori": "orientation(s)", "pc": "projection center(s)", "ori_pc": "orientation(s) and projection center(s)"}
        print(f"Refining {n} {what[mode]}:")
    ctx = context if context is not None else (_lib.Context(device) if not contexts else None)

    def solve_block(c, part):
        c.set_master_pattern(*_master_pattern_data(master_pattern, energy))
        # rescale exactly when the patterns are float32 (_refinement.py:956)
        c.refine_set_patterns(pats[part], signal_mask, pats.dtype == np.float32, detector.detector_to_sample)
        if host is None:  # the whole simplex search on the device
            return c.refine_solve(MODES[mode], x0[part], None if fixed is None else fixed[part],
                                  None if lower is None else lower[part], None if upper is None else upper[part],
                                  nm["xatol"], nm["fatol"], nm["maxiter"] or 0, nm["maxfev"] or 0)
        # the reference's optimiser on the host, the objective on the device
        return _host_solve(c, MODES[mode], host, x0[part], None if fixed is None else fixed[part],
                           None if lower is None else lower[part], None if upper is None else upper[part])

    try:
        t0 = time.time()
        if hi_i > lo_i and contexts and len(contexts) > 1:
            # one block of the points per GPU, each driven from its own host thread
            from concurrent.futures import ThreadPoolExecutor

            from kikuchipy_amd.parallel import shard_range

            blocks = [shard_range(hi_i - lo_i, i, len(contexts)) for i in range(len(contexts))]
            jobs = [(c, slice(lo_i + a, lo_i + b)) for c, (a, b) in zip(contexts, blocks) if b > a]
            with ThreadPoolExecutor(len(jobs)) as pool:
                res = np.concatenate(list(pool.map(lambda job: solve_block(*job), jobs)), axis=0)
        elif hi_i > lo_i:
            res = solve_block(ctx if ctx is not None else contexts[0], part)
        else:
            res = np.empty((0, starts, 3 + x0.shape[2]))
        if comm is not None and comm.world_size > 1:
            res = comm.all_gather_rows(res)
        total = time.time() - t0
    finally:
        if context is None and ctx is not None:
            ctx.close()
    if verbose:
        print(f"Refinement speed: {n / total:.5f} patterns/s")

    ncc = 1 - res[:, :, 0]
    best = np.argmax(ncc, axis=1)  # first maximum, like int(np.argmax(ncc_all)) in the solvers
    pick = res[np.arange(n), best]
    scores, num_evals, x = ncc[np.arange(n), best], pick[:, 1].astype(np.int32), pick[:, 3:]
    ps_index = best.astype(np.int32) if n_pseudo > 0 else None
    euler = x[:, :3] if mode != "pc" else None
    result = RefinementResult(scores, num_evals, euler, points, nav_shape or (1,), ps_index, n / total)
    new_detector = None
    if mode != "ori":
        new_pc = x[:, -3:]
        new_detector = detector.deepcopy()
        if points.all() and nav_shape:
            new_pc = new_pc.reshape(nav_shape + (3,))
        new_detector.pc = new_pc
    # the reference's result rows (_refinement.py:120-128, :184-190, :284-290): score, number of evaluations, the refined
    # variables, [pseudo-symmetry index]
    raw = np.column_stack([scores, num_evals, x] + ([ps_index] if ps_index is not None else []))
    return result, new_detector, raw


class DeferredRefinement:
    """`refine_*(..., compute=False)`: everything is validated and set up, nothing has run.  The reference hands out a
    lazy Dask array here and finishes with `kikuchipy.indexing.compute_refine_*_results(results, ...)`
    (indexing/_refinement/_refinement.py:58-290); this object plays that array's part: `.compute()` returns the rows
    the Dask array would hold - (score, number of evaluations, refined variables[, pseudo-symmetry index]) per refined
    point - and the `compute_refine_*_results` functions of this module return what `compute=True` returns."""

    def __init__(self, mode, run, pseudo_symmetry_checked):
        self.mode = mode
        self._run = run
        self.pseudo_symmetry_checked = bool(pseudo_symmetry_checked)
        self._done = None

    def _finish(self):
        if self._done is None:
            self._done = self._run()
            self._run = None  # drops the patterns
        return self._done

    def compute(self):
        return self._finish()[2]

    def __repr__(self):
        state = "computed" if self._done is not None else "not computed"
        return f"DeferredRefinement(mode={self.mode!r}, {state})"


def _deferred(results, mode, pseudo_symmetry_checked=None):
    if not isinstance(results, DeferredRefinement):
        raise TypeError("`results` must be what refine_*(..., compute=False) returned")
    if results.mode != mode:
        raise ValueError(f"`results` come from a {results.mode!r} refinement, not {mode!r}")
    if pseudo_symmetry_checked is not None and bool(pseudo_symmetry_checked) != results.pseudo_symmetry_checked:
        raise ValueError("`pseudo_symmetry_checked` does not match the refinement that produced `results`")
    return results._finish()


def compute_refine_orientation_results(results, xmap=None, master_pattern=None, navigation_mask=None,
                                       pseudo_symmetry_checked=None):
    """indexing/_refinement/_refinement.py:58-131: the `RefinementResult` of `refine_orientation(compute=False)`
    (the other arguments are those of the reference; the deferred object already carries them)."""
    return _deferred(results, "ori", pseudo_symmetry_checked)[0]


def compute_refine_projection_center_results(results, detector=None, xmap=None, navigation_mask=None):
    """:133-197: `(scores, new detector, num_evals)`."""
    res, det, _ = _deferred(results, "pc")
    return res.scores, det, res.num_evals


def compute_refine_orientation_projection_center_results(results, detector=None, xmap=None, master_pattern=None,
                                                         navigation_mask=None, pseudo_symmetry_checked=None):
    """:199-290: `(RefinementResult, new detector)`."""
    res, det, _ = _deferred(results, "ori_pc", pseudo_symmetry_checked)
    return res, det
