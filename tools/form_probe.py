"""Which f32 match kernel should serve a sweep?  Both forced, and the automatic choice, over a grid of shapes.

    python tools/form_probe.py [out.json] [--quick]

For every (M experimental patterns, N dictionary patterns, K kept pixels) the resident-data step (preparation of both
sides + match + merge; the kernels read different operand layouts, so the preparation belongs to the comparison) is timed
with KPDI_F32_WIDE=0 (match.hip: 128 x 256 tiles, dynamic hand-out, quarter-tile tail launch), KPDI_F32_WIDE=1
(match16.hip's f32 form: 256 x 256 tiles, static hand-out with partial units) and with the variable unset (decide_form,
api.hip).  The JSON is what `decide_form`'s cost model is checked against (tests/test_gpu_engine.py reads the same
grid live) and what its constants were fitted on (csrc/form_model.h, written by --fit)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kikuchipy_amd import _lib  # noqa: E402

GRID_N = [6250, 12500, 25000, 37500, 50000, 100000, 300000]
GRID_M = [512, 4096, 10000, 40000]
GRID_K = [2819, 3600, 14400]


def circular_mask(s):
    yy, xx = np.ogrid[:s, :s]
    return np.sqrt((yy - s // 2) ** 2 + (xx - s // 2) ** 2) > s // 2


def time_step(ctx, d_exp, m, d_dic, n, side, mask, metric, reps):
    """best-of-`reps` wall time (ms) of one resident-data step, and the kernel that ran"""
    ctx.set_problem(side, side, mask, metric, 20)
    best = 1e9
    for r in range(reps + 1):
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.set_experimental_dev(d_exp, np.uint8, m)
        ctx.push_dictionary_chunk_dev(d_dic, np.float32, n, 0)
        ctx.finalize(20)
        if r:
            best = min(best, (time.perf_counter() - t0) * 1e3)
    return best, ctx.counters()["match_form"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out", nargs="?")
    ap.add_argument("--quick", action="store_true", help="a sparse sub-grid (what the GPU test runs)")
    ap.add_argument("--metric", default="ncc", choices=["ncc", "ndp"])
    ap.add_argument("--only", default="", help="M:N:K,M:N:K,... - just these points")
    a = ap.parse_args()
    grid_n, grid_m = (GRID_N[1::2], GRID_M[:2]) if a.quick else (GRID_N, GRID_M)
    if a.only:
        grid_m = sorted({int(p.split(":")[0]) for p in a.only.split(",")})
        grid_n = sorted({int(p.split(":")[1]) for p in a.only.split(",")})
    rng = np.random.default_rng(5)
    # one buffer of random floats serves every shape (100 000 patterns of 120 x 120 = 400 000 of 60 x 60)
    pool = rng.random(100000 * 14400, dtype=np.float32)
    exp_pool = rng.integers(0, 256, max(grid_m) * 14400, dtype=np.uint8)
    metric = _lib.METRIC_NCC if a.metric == "ncc" else _lib.METRIC_NDP
    rows = []
    with _lib.Context(0) as ctx:
        ctx.set_problem(60, 60, None, metric, 20)
        d_dic = ctx.dev_alloc(pool.nbytes)
        ctx.h2d(d_dic, pool)
        d_exp = ctx.dev_alloc(exp_pool.nbytes)
        ctx.h2d(d_exp, exp_pool)
        for k in GRID_K:
            side = 120 if k == 14400 else 60
            mask = circular_mask(60) if k == 2819 else None
            for m in grid_m:
                for n in grid_n:
                    if n * side * side > pool.size:
                        continue
                    if a.only and f"{m}:{n}:{k}" not in a.only.split(","):
                        continue
                    flop = 2.0 * m * n * k
                    reps = 2 if flop > 2e13 else 4
                    rec = {"M": m, "N": n, "K": k}
                    for name, env in (("classic", "0"), ("wide", "1"), ("auto", None)):
                        if env is None:
                            os.environ.pop("KPDI_F32_WIDE", None)
                        else:
                            os.environ["KPDI_F32_WIDE"] = env
                        ms, form = time_step(ctx, d_exp, m, d_dic, n, side, mask, metric, reps)
                        rec[name + "_ms"] = round(ms, 4)
                        if env is None:
                            rec["auto_chose"] = "wide" if form == 3 else "classic"
                    os.environ.pop("KPDI_F32_WIDE", None)
                    better = min(rec["classic_ms"], rec["wide_ms"])
                    rec["auto_over_better"] = round(rec["auto_ms"] / better, 4)
                    rec["wide_over_classic"] = round(rec["wide_ms"] / rec["classic_ms"], 4)
                    rows.append(rec)
                    print(rec, flush=True)
    worst = max(rows, key=lambda r: r["auto_over_better"])
    out = {"what": "resident-data step (ms, best of a few) with each f32 match kernel forced and with the automatic choice; "
                   f"metric {a.metric}, keep_n 20, one MI355X",
           "worst_auto_over_better": worst, "rows": rows}
    print("worst:", worst)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
