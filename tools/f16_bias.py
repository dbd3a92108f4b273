"""Accumulation-error probe of the split-f16 mode (developer tool)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kikuchipy_amd import _lib  # noqa: E402

rng = np.random.default_rng(3)
for (s, metric, kind) in [(60, "ndp", "u8"), (60, "ncc", "u8"), (60, "ncc", "copy"), (44, "ndp", "u8"), (120, "ndp", "u8"), (60, "ndp", "f32")]:
    m, n = 64, 512
    if kind == "f32":
        exp = rng.random((m, s, s), dtype=np.float32)
        dic = rng.random((n, s, s), dtype=np.float32)
    else:
        exp = rng.integers(1, 252, (m, s, s), dtype=np.uint8)
        dic = rng.integers(1, 252, (n, s, s), dtype=np.uint8)
    if kind == "copy":
        dic[:m] = exp
    e = exp.reshape(m, -1).astype(np.float64)
    d = dic.reshape(n, -1).astype(np.float64)
    if metric == "ncc":
        e -= e.mean(1, keepdims=True)
        d -= d.mean(1, keepdims=True)
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    exact = e @ d.T
    out = {}
    for mode in (_lib.COMPUTE_F32, _lib.COMPUTE_F16X2):
        with _lib.Context(0) as c:
            c.set_problem(s, s, None, {"ncc": _lib.METRIC_NCC, "ndp": _lib.METRIC_NDP}[metric], 8, mode)
            c.set_experimental(exp)
            c.push_dictionary_chunk(dic, 0)
            sc, ix = c.finalize(8)
        err = sc - np.take_along_axis(exact, ix, 1)
        out[mode] = (err.mean(), np.abs(err).max(), sc.mean())
    print(f"{s}x{s} {metric} {kind}: f32 mean err {out[0][0]:+.2e} max {out[0][1]:.2e} | f16x2 mean err {out[1][0]:+.2e} max {out[1][1]:.2e} | mean score {out[1][2]:.3f}")
