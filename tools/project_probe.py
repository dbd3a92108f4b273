"""Timing probe of on-device dictionary generation + sweep (developer tool)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kikuchipy_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=4096)
ap.add_argument("--n", type=int, default=100000)
ap.add_argument("--s", type=int, default=60)
ap.add_argument("--npx", type=int, default=401)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--rescale", type=int, default=1)
a = ap.parse_args()

rng = np.random.default_rng(1)
up = rng.random((a.npx, a.npx), dtype=np.float32)
lo = rng.random((a.npx, a.npx), dtype=np.float32)
q = rng.standard_normal((a.n, 4))
q /= np.linalg.norm(q, axis=1)[:, None]
exp = rng.integers(0, 256, (a.m, a.s, a.s), dtype=np.uint8)
ctx = _lib.Context(0)
ctx.set_master_pattern(up, lo)
pc = (0.42, 0.78, 0.5)
bounds = [-pc[0] / pc[2], (1 - pc[0]) / pc[2], -(1 - pc[1]) / pc[2], pc[1] / pc[2]]
t = np.deg2rad(70)
om = np.array([[0, 1, 0], [-np.cos(t) * 0 - np.sin(t) * 1, 0, np.cos(t)], [np.cos(t), 0, np.sin(t)]]).T
ctx.set_detector(bounds, pc[2], a.s, a.s, om)
ctx.set_problem(a.s, a.s, None, _lib.METRIC_NCC, 20)
ctx.set_experimental(exp)
ctx.set_profiling(True)
for rep in range(a.reps + 1):
    ctx.reset_topk()
    ctx.reset_counters()
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.push_rotations_chunk(q, 0, bool(a.rescale), -1, 1)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    c = ctx.counters()
    gpix = a.n * a.s * a.s / (c["project_ms"] * 1e-3) / 1e9
    print(f"rep {rep}: wall {dt*1e3:.2f} ms  project {c['project_ms']:.2f} ms ({gpix:.1f} Gpixel/s)  "
          f"prep {c['prep_ms']:.2f}  match {c['match_ms']:.2f}  merge {c['merge_ms']:.2f} ms", flush=True)
