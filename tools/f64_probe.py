"""Cost of float64 arithmetic (COMPUTE_F64: f32 screen + float64 rescoring) at configs[1]'s shape (developer tool)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kikuchipy_amd import _lib  # noqa: E402

m, n, s, k = 4096, 100000, 60, 20
rng = np.random.default_rng(2024)
exp = rng.integers(0, 256, (m, s, s), dtype=np.uint8)
dic = rng.random((n, s, s), dtype=np.float32)
ctx = _lib.Context(0)
d = ctx.dev_alloc(dic.nbytes)
ctx.h2d(d, dic)
ctx.set_profiling(True)
for mode, name in ((_lib.COMPUTE_F32, "f32"), (_lib.COMPUTE_F64, "f64")):
    ctx.set_problem(s, s, None, _lib.METRIC_NCC, k, mode)
    ctx.set_experimental(exp)
    for rep in range(4):
        ctx.reset_topk()
        ctx.reset_counters()
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.push_dictionary_chunk_dev(d, np.float32, n, 0)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        c = ctx.counters()
    print(f"{name}: step {dt*1e3:.2f} ms  match {c['match_ms']:.2f}  prep {c['prep_ms']:.2f}  merge {c['merge_ms']:.2f}  "
          f"rescore {c['rescore_ms']:.2f}  launches {c['match_launches']}  extra passes {c['rescore_extra_passes']}  "
          f"uncertified {c['uncertified_patterns']}  -> {m/dt:.0f} patterns/s")
    sc, ix = ctx.finalize(k)
    print("   row0", ix[0, :4], sc[0, :4], sc.dtype)
