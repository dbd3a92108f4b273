"""One size of the fused pre-kernel, twice (the command the counter passes of tools/pmc_any.sh wrap):
python tools/pk_probe.py <detector side> <patterns> [masked]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kikuchipy_amd import _lib  # noqa: E402

sy = sx = int(sys.argv[1])
m = int(sys.argv[2])
rng = np.random.default_rng(0)
bg = rng.integers(1, 256, (sy, sx)).astype(np.float32)
dic = rng.random((256, sy, sx), dtype=np.float32)
exp = rng.integers(0, 256, (m, sy, sx), dtype=np.uint8)
mask = None
if len(sys.argv) > 3:
    yy, xx = np.ogrid[:sy, :sx]
    mask = np.sqrt((yy - sy // 2) ** 2 + (xx - sx // 2) ** 2) > max(sy // 2, sx // 2)
with _lib.Context(0) as ctx:
    ctx.set_problem(sy, sx, mask, _lib.METRIC_NCC, 1)
    d_exp = ctx.dev_alloc(exp.nbytes)
    ctx.h2d(d_exp, exp)
    ctx.set_profiling(True)
    for rep in range(3):
        if rep == 1:
            ctx.reset_counters()
        ctx.set_experimental_dev(d_exp, exp.dtype, m)
        ctx.remove_static_background(bg, _lib.OP_SUBTRACT, False)
        ctx.remove_dynamic_background(_lib.OP_SUBTRACT, _lib.DOMAIN_FREQUENCY, 0.0, 4.0)
        ctx.push_dictionary_chunk(dic, 0)
        ctx.finalize(1)
    c = ctx.counters()
    print(f"{sy}x{sx} M={m}: {c['preproc_ms'] / c['preproc_launches']:.4f} ms per launch")
