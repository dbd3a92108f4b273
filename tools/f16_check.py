"""Split-f16 compute mode vs the f32 path (developer tool)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kikuchipy_amd import _lib  # noqa: E402

rng = np.random.default_rng(1)
for (m, n, s, k) in [(300, 3000, 60, 20), (48, 700, 33, 5), (1000, 20000, 60, 20)]:
    exp = rng.integers(0, 256, (m, s, s), dtype=np.uint8)
    dic = rng.random((n, s, s), dtype=np.float32)
    dic[7] = exp[3].astype(np.float32) * 0.01 + 2
    out = {}
    for mode in (_lib.COMPUTE_F32, _lib.COMPUTE_F16X2):
        with _lib.Context(0) as ctx:
            ctx.set_problem(s, s, None, _lib.METRIC_NCC, k, mode)
            ctx.set_experimental(exp)
            ctx.push_dictionary_chunk(dic, 0)
            out[mode] = ctx.finalize(k)
    (s0, i0), (s1, i1) = out[0], out[1]
    print(f"m={m} n={n} s={s}: max|dscore| {np.abs(s0 - s1).max():.2e}  index mismatches {np.count_nonzero(i0 != i1)} "
          f"of {i0.size}  best of row 3: {i1[3, 0]} {s1[3, 0]:.7f} (f32 {s0[3, 0]:.7f})")
