"""Orientation sampling for dictionaries: `get_sample_fundamental(resolution, point_group)`.

The reference builds its dictionaries from `orix.sampling.get_sample_fundamental(resolution=..., point_group=...)`
(benchmarks/indexing/test_dictionary_indexing.py:37; doc/tutorials/pattern_matching.ipynb cell 8:
`method="cubochoric", resolution=3, point_group=ni.point_group`).  orix (>= 0.12.1, pyproject.toml:54) is a third-party
dependency that is neither vendored in the reference nor installed here; this module restates the published algorithm
behind that call - cubochoric sampling of SO(3), Rosca, Morawiec & De Graef, Modelling Simul. Mater. Sci. Eng. 22 (2014)
075013; Singh & De Graef, ibid. 24 (2016) 085013 - for the cases the reference uses:

  * a cubic grid of (2 n + 1)^3 points over the cube of semi-edge pi^(2/3) / 2, n = round(131.97049 / (resolution -
    0.03732)) (the grid INCLUDES the cube's surface), in lexicographic order (x slowest) - the order the dictionary is
    emitted in;
  * cube -> homochoric ball (the volume-preserving Lambert-type map) -> axis-angle -> unit quaternion (a, b, c, d), a >= 0;
  * kept where the Rodrigues vector lies in the fundamental zone of the point group's proper subgroup - 432 for
    m-3m / 432: |r_i| <= sqrt(2) - 1 and |r_1| + |r_2| + |r_3| <= 1.

Pinned by what the reference itself holds for this path: 30 443 orientations at `resolution=3` (the tutorial's printed
`Rotation (30443,)`) and the benchmark's known answer - nine Ni patterns against the 3557-orientation dictionary of
`resolution=6`: `scores.mean() = 0.1887 +- 1e-4` (tests/test_reference_benchmark.py, CPU through the oracle and GPU through
the engine)."""

import numpy as np

_SEMI_EDGE = np.pi ** (2 / 3) / 2


def resolution_to_semi_edge_steps(resolution):
    """Grid points per semi-edge of the cubochoric cube for an average disorientation of `resolution` degrees between
    neighbouring samples (EMsoft's fit, as orix uses it)."""
    if not resolution > 0.03732:
        raise ValueError("resolution must be positive (degrees)")
    return int(np.round(131.97049 / (resolution - 0.03732)))


def cubochoric_to_homochoric(xyz):
    """(N, 3) points of the cube [-pi^(2/3)/2, pi^(2/3)/2]^3 -> the ball of radius (3 pi / 4)^(1/3), volume preserving
    (Rosca et al. 2014, eqs. 2-4: scale the cube, map each square pyramid to a curved one, inverse Lambert projection)."""
    xyz = np.asarray(xyz, dtype=np.float64)
    pyramid = np.argmax(np.abs(xyz), axis=1)  # the axis of largest magnitude goes last
    order = np.array([[1, 2, 0], [2, 0, 1], [0, 1, 2]])[pyramid]
    x, y, z = (np.pi ** (5 / 6) / 6 ** (1 / 6) / np.pi ** (2 / 3) * np.take_along_axis(xyz, order, axis=1)).T
    r1, beta = (3 * np.pi / 4) ** (1 / 3), np.pi ** (5 / 6) / 6 ** (1 / 6) / 2
    prek, sr2 = r1 * 2 ** 0.25 / beta, np.sqrt(2.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        flat = np.abs(y) <= np.abs(x)
        q = np.where(flat, np.pi / 12 * y / x, np.pi / 12 * x / y)
        q = np.where(np.isfinite(q), q, 0.0)
        c, s = np.cos(q), np.sin(q)
        qq = np.where(flat, x, y) * prek / np.sqrt(sr2 - c)
        t1 = np.where(flat, (sr2 * c - 1) * qq, sr2 * s * qq)
        t2 = np.where(flat, sr2 * s * qq, (sr2 * c - 1) * qq)
        cc = t1 * t1 + t2 * t2
        shrink = np.sqrt(np.maximum(1 - np.pi * cc / (24 * z * z), 0))
        ball = np.stack([t1 * shrink, t2 * shrink, np.sqrt(6 / np.pi) * z - np.sqrt(np.pi) * cc / np.sqrt(24) / z], axis=1)
    on_axis = np.maximum(np.abs(x), np.abs(y)) == 0
    ball[on_axis] = np.stack([np.zeros(on_axis.sum()), np.zeros(on_axis.sum()), np.sqrt(6 / np.pi) * z[on_axis]], axis=1)
    ball[~np.isfinite(ball).all(axis=1)] = 0.0  # (the cube's centre)
    return np.take_along_axis(ball, np.argsort(order, axis=1), axis=1)


def homochoric_to_axis_angle(ho):
    """|ho| = (3/4 (w - sin w))^(1/3): (unit axes (N, 3), angles w in [0, pi]); bisection, exact to float64."""
    h = np.linalg.norm(ho, axis=1)
    lo, hi = np.zeros_like(h), np.full_like(h, np.pi)
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        below = (0.75 * (mid - np.sin(mid))) ** (1 / 3) < h
        lo, hi = np.where(below, mid, lo), np.where(below, hi, mid)
    axis = np.divide(ho, h[:, None], out=np.zeros_like(ho), where=h[:, None] > 0)
    return axis, 0.5 * (lo + hi)


_FUNDAMENTAL_ZONES = {"m-3m": "432", "432": "432", "1": "1", "-1": "1"}


def get_sample_fundamental(resolution=2, point_group="m-3m", method="cubochoric", semi_edge_steps=None):
    """(N, 4) unit quaternions (a, b, c, d), a >= 0, sampling the Rodrigues fundamental zone of `point_group` with an
    average disorientation of `resolution` degrees, in the sampler's (lexicographic) order.  `point_group`: its name
    ("m-3m", "432"; "1" = all of SO(3)) or an object with a `.name` (an orix `Symmetry`)."""
    if method != "cubochoric":
        raise NotImplementedError("only the cubochoric sampling of the reference's benchmark and tutorial is restated")
    name = getattr(point_group, "name", point_group)
    if name not in _FUNDAMENTAL_ZONES:
        raise NotImplementedError(f"point group {name!r}: only {sorted(_FUNDAMENTAL_ZONES)} have their fundamental zone restated")
    n = int(semi_edge_steps) if semi_edge_steps is not None else resolution_to_semi_edge_steps(resolution)
    g = np.arange(-n, n + 1) * (_SEMI_EDGE / n)
    out = []
    for x in g:  # one slab of the cube at a time: (2 n + 1)^2 points
        yy, zz = np.meshgrid(g, g, indexing="ij")
        xyz = np.column_stack([np.full(yy.size, x), yy.ravel(), zz.ravel()])
        axis, w = homochoric_to_axis_angle(cubochoric_to_homochoric(xyz))
        if _FUNDAMENTAL_ZONES[name] == "432":
            with np.errstate(invalid="ignore"):
                r = np.abs(axis * np.tan(w / 2)[:, None])
                keep = (r.max(axis=1) <= np.sqrt(2) - 1 + 1e-9) & (r.sum(axis=1) <= 1 + 1e-9)
            axis, w = axis[keep], w[keep]
        out.append(np.column_stack([np.cos(w / 2), axis * np.sin(w / 2)[:, None]]))
    return np.concatenate(out)
