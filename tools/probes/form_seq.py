"""One shape, the forced / automatic kernel choices in a given order (does a step's time depend on what ran before it?):
    python tools/probes/form_seq.py M N [first_M first_N]"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kikuchipy_amd import _lib  # noqa: E402

spec = importlib.util.spec_from_file_location("form_probe", os.path.join(ROOT, "tools", "form_probe.py"))
fp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fp)
m, n = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(5)
pool = rng.random(100000 * 3600, dtype=np.float32)
exp_pool = rng.integers(0, 256, 40000 * 3600, dtype=np.uint8)


def run(ctx, name, mm, nn):
    env = {"classic": "0", "wide": "1", "auto": None}[name]
    if env is None:
        os.environ.pop("KPDI_F32_WIDE", None)
    else:
        os.environ["KPDI_F32_WIDE"] = env
    ms, form = fp.time_step(ctx, d_exp, mm, d_dic, nn, 60, None, _lib.METRIC_NCC, 5)
    ctx.set_profiling(True)
    ctx.reset_counters()
    for _ in range(3):
        ctx.set_experimental_dev(d_exp, np.uint8, mm)
        ctx.push_dictionary_chunk_dev(d_dic, np.float32, nn, 0)
        ctx.finalize(20)
    c = ctx.counters()
    ctx.set_profiling(False)
    print(f"{mm:6d} x {nn:6d} {name:8s} {ms:8.4f} ms  form {form}  grid {c.get('match_grid')} nsplit {c.get('match_nsplit')}"
          f"  match {c['match_ms'] / 3:.4f} prep {c['prep_ms'] / 3:.4f} merge {c['merge_ms'] / 3:.4f} fixed {c.get('fixed_ms', 0) / 3:.4f} kpad {c.get('kpad')}", flush=True)


with _lib.Context(0) as ctx:
    ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 20)
    d_dic = ctx.dev_alloc(pool.nbytes)
    ctx.h2d(d_dic, pool)
    d_exp = ctx.dev_alloc(exp_pool.nbytes)
    ctx.h2d(d_exp, exp_pool)
    if len(sys.argv) > 4:
        for name in ("classic", "wide", "auto"):
            run(ctx, name, int(sys.argv[3]), int(sys.argv[4]))
    for name in ("classic", "wide", "auto", "classic", "auto", "wide", "auto", "auto", "classic"):
        run(ctx, name, m, n)
