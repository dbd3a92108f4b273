"""Quick timing probe of the resident-data sweep (developer tool, not bench.py)."""
import argparse
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kikuchipy_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=4096)
ap.add_argument("--n", type=int, default=100000)
ap.add_argument("--s", type=int, default=60)
ap.add_argument("--k", type=int, default=20)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--mask", action="store_true")
ap.add_argument("--f16", action="store_true", help="opt-in split-f16 compute mode")
ap.add_argument("--half", action="store_true", help="opt-in plain float16 compute mode (reduced precision)")
a = ap.parse_args()

rng = np.random.default_rng(2024)
exp = rng.integers(0, 256, (a.m, a.s, a.s), dtype=np.uint8)
t0 = time.time()
dic = rng.random((a.n, a.s, a.s), dtype=np.float32)
print(f"host data {time.time() - t0:.1f}s", flush=True)
ctx = _lib.Context(0)
mask = None
if a.mask:
    yy, xx = np.ogrid[:a.s, :a.s]
    mask = np.sqrt((yy - a.s // 2) ** 2 + (xx - a.s // 2) ** 2) > a.s // 2
ctx.set_problem(a.s, a.s, mask, _lib.METRIC_NCC, a.k, _lib.COMPUTE_F16 if a.half else (_lib.COMPUTE_F16X2 if a.f16 else _lib.COMPUTE_F32))
d_dic = ctx.dev_alloc(dic.nbytes)
ctx.h2d(d_dic, dic)
ctx.set_experimental(exp)
ctx.set_profiling(True)
for rep in range(a.reps + 1):
    ctx.reset_topk()
    ctx.reset_counters()
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.push_dictionary_chunk_dev(d_dic, np.float32, a.n, 0)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    c = ctx.counters()
    tf = c["match_flops"] / (c["match_ms"] * 1e-3) / 1e12
    print(f"rep {rep}: wall {dt*1e3:.2f} ms  match {c['match_ms']:.2f} ms ({tf:.1f} TF/s, "
          f"{100*tf/157.3:.1f}% of f32 MFMA peak)  prep {c['prep_ms']:.2f} ms  merge {c['merge_ms']:.2f} ms  "
          f"grid {c['match_grid']} nsplit {c['match_nsplit']}  {a.m/dt:.0f} patterns/s", flush=True)
s, i = ctx.finalize(a.k)
print("row0", i[0, :5], s[0, :5])
