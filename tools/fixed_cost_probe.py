"""Per-launch fixed cost of the match kernel (developer tool): chunks of several sizes pushed
into a cold sweep (empty rejection bound) and into a warm one."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kikuchipy_amd import _lib  # noqa: E402

rng = np.random.default_rng(0)
exp = rng.integers(0, 256, (4096, 60, 60), dtype=np.uint8)
dic = rng.random((50000, 60, 60), dtype=np.float32)
ctx = _lib.Context(0)
ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 20)
ctx.set_experimental(exp)
d = ctx.dev_alloc(dic.nbytes)
ctx.h2d(d, dic)
ctx.set_profiling(True)
row = 3600 * 4


def timed_push(n, start):
    ctx.synchronize()
    ctx.reset_counters()
    ctx.push_dictionary_chunk_dev(d + start * row, np.float32, n, start)
    ctx.synchronize()
    c = ctx.counters()
    return c["match_ms"], c["prep_ms"], c["merge_ms"]


for tiles in (16, 32, 48, 64, 96, 98, 112, 192):
    n = tiles * 128
    out = []
    for rep in range(3):
        ctx.reset_topk()
        cold = timed_push(n, 0)
        warm = timed_push(n, n)
        out.append((cold[0], warm[0], warm[1], warm[2]))
    cold, warm, prep, merge = np.min(np.array(out), axis=0)
    units = -(-tiles * 16 // 256)
    print(f"{tiles:4d} tiles ({units:2d}/WG): cold {cold:.3f} ms  warm {warm:.3f} ms  per unit {warm/units:.3f}  "
          f"prep {prep:.3f}  merge {merge:.3f}", flush=True)
