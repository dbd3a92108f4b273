"""Time the fused background-removal + preparation kernel (csrc/preproc.hip) for a range of
pattern counts: python tools/prekernel_probe.py [sy sx]   (needs a GPU)

Prints per M: kernel ms (HIP events around the launch), algorithmic GB/s (pattern read + written
back + prepared row) and the fraction of the 8 TB/s HBM peak."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kikuchipy_amd import _lib  # noqa: E402

sy, sx = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (60, 60)
rng = np.random.default_rng(0)
yy, xx = np.ogrid[:sy, :sx]
mask = np.sqrt((yy - sy // 2) ** 2 + (xx - sx // 2) ** 2) > max(sy // 2, sx // 2)
bg = rng.integers(1, 256, (sy, sx)).astype(np.float32)
dic = rng.random((256, sy, sx), dtype=np.float32)
with _lib.Context(0) as ctx:
    for masked in (True, False):
        for m in (4096, 16384, 65536, 262144):
            if m * sy * sx > 2**31:
                continue
            exp = rng.integers(0, 256, (m, sy, sx), dtype=np.uint8)
            ctx.set_problem(sy, sx, mask if masked else None, _lib.METRIC_NCC, 1)
            d_exp = ctx.dev_alloc(exp.nbytes)
            ctx.h2d(d_exp, exp)
            ctx.set_profiling(True)
            for rep in range(4):
                if rep == 1:
                    ctx.reset_counters()
                ctx.set_experimental_dev(d_exp, exp.dtype, m)
                ctx.remove_static_background(bg, _lib.OP_SUBTRACT, False)
                ctx.remove_dynamic_background(_lib.OP_SUBTRACT, _lib.DOMAIN_FREQUENCY, 0.0, 4.0)
                ctx.push_dictionary_chunk(dic, 0)
                ctx.finalize(1)
            c = ctx.counters()
            ctx.set_profiling(False)
            ctx.dev_free(d_exp)
            ms = c["preproc_ms"] / c["preproc_launches"]
            nbytes = m * (2 * sy * sx + c["kpad"] * 4)
            print(f"{sy}x{sx} masked={masked} M={m}: {ms:.4f} ms, {nbytes / ms / 1e6:.1f} GB/s "
                  f"({nbytes / ms / 1e6 / 8000:.3f} of 8 TB/s), {m / ms / 1e3:.2f} M patterns/s", flush=True)
