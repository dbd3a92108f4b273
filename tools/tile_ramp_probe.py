"""Fixed cost of a match launch: 4096 experimental x (2048 j) dictionary patterns, j = 1 .. 12 whole tiles per workgroup
(16 row blocks x 16 workgroups, match.hip) - match_ms against j; the intercept is what a launch costs beyond its tiles.

    python tools/tile_ramp_probe.py [wide]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wide = len(sys.argv) > 1 and sys.argv[1] == "wide"
os.environ["KPDI_F32_WIDE"] = "1" if wide else "0"
from kikuchipy_amd import _lib  # noqa: E402

rng = np.random.default_rng(3)
exp = rng.integers(0, 256, (4096, 60, 60), dtype=np.uint8)
unit = 4096 if wide else 2048
dic = rng.random((12 * unit, 60, 60), dtype=np.float32)
xs, ys = [], []
with _lib.Context(0) as ctx:
    ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 20)
    d_exp = ctx.dev_alloc(exp.nbytes)
    ctx.h2d(d_exp, exp)
    d_dic = ctx.dev_alloc(dic.nbytes)
    ctx.h2d(d_dic, dic)
    ctx.set_profiling(True)
    for j in list(range(1, 13)):
        n = unit * j
        for rep in range(13):
            if rep == 3:
                ctx.reset_counters()
            ctx.set_experimental_dev(d_exp, exp.dtype, 4096)
            ctx.push_dictionary_chunk_dev(d_dic, np.float32, n, 0)
            try:
                ctx.finalize(20)
            except _lib.KpdiError:  # (timing-only ablation builds of the kernel hand out empty lists)
                pass
        c = ctx.counters()
        ms = c["match_ms"] / 10
        xs.append(j)
        ys.append(ms)
        print(f"{j:2d} tiles per workgroup ({n} patterns): match {ms:.4f} ms = {ms / j:.4f} per tile, launches {c['match_launches'] / 10:.0f}", flush=True)
b, a = np.polyfit(xs, ys, 1)
print(f"fit: {a:.4f} ms + {b:.4f} ms per tile")
