"""One rank's share of a dictionary-sharded job on ONE GPU: for N = 1, 2, 4, 8 ranks the sweep of rank 0's
shard (shard_range(n, 0, N) - a prefix of the dictionary) is timed with the inputs resident in HBM and
compared with the even share t_1 / N - what strong scaling over N GPUs can reach before the RCCL
all-gather (tens of microseconds for 4096 x 20 x 8 B per rank; 6.4 MB per rank at configs[3]).

    python tools/rank_share_probe.py [out.json] [--workload config2|config4|config5] [--compute f32|f16]
                                     [--ranks 1,2,4,8] [--reps 20] [--no-whole-tiles]

config4 = BASELINE.json configs[3] (40 000 x 300 000 x 60 x 60, ndp), config5 = configs[4] (4096 x 500 000 x
120 x 120).  `--pmc-shard N` runs ONLY rank 0's shard of an N-rank job, `--reps` times - the command the
rocprofv3 counter passes wrap (tools/collect_r03.sh)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from kikuchipy_amd import _lib  # noqa: E402
from kikuchipy_amd.parallel import shard_range  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("out", nargs="?")
ap.add_argument("--workload", default="config2", choices=["config2", "config4", "config5"])
ap.add_argument("--compute", default="f32", choices=["f32", "f16"])
ap.add_argument("--ranks", default="1,2,4,8")
ap.add_argument("--reps", type=int, default=0)
ap.add_argument("--no-whole-tiles", action="store_true", help="skip the second pass with KPDI_NO_TAIL=1")
ap.add_argument("--pmc-shard", type=int, default=0)
ap.add_argument("--pipeline", action="store_true", help="collect step i's result while step i + 1 runs (finalize_async / finalize_wait)")
ap.add_argument("--dict-dtype", default="f32", choices=["f32", "f16"], help="dtype the raw dictionary is resident in")
a = ap.parse_args()

w = bench.WORKLOADS[a.workload]
m, n, sy, sx, keep = w["m"], w["n"], w["sy"], w["sx"], w["keep_n"]
large = a.workload != "config2"
ranks_list = [int(x) for x in a.ranks.split(",")] if not a.pmc_shard else [a.pmc_shard]
rng = np.random.default_rng(2024)
exp = rng.integers(0, 256, (m, sy, sx), dtype=np.uint8)
n_need = max(shard_range(n, 0, r)[1] for r in ranks_list)  # rank 0's shards are prefixes of the dictionary
out = {"workload": w["name"] + f"; rank 0's shard on one MI355X, compute {a.compute}, raw dictionary resident as {a.dict_dtype}"
                   + ("; results collected while the next step runs (finalize_async / finalize_wait)" if a.pipeline else ""),
       "ranks": {}}
with _lib.Context(0) as ctx:
    metric = {"ncc": _lib.METRIC_NCC, "ndp": _lib.METRIC_NDP}[w["metric"]]
    compute = {"f32": _lib.COMPUTE_F32, "f16": _lib.COMPUTE_F16}[a.compute]
    ctx.set_problem(sy, sx, None, metric, keep, compute)
    d_exp = ctx.dev_alloc(exp.nbytes)
    ctx.h2d(d_exp, exp)
    t_gen = time.perf_counter()
    if large:
        d_dic = bench.upload_generated_shard(ctx, w, 0, n_need, None, np.float16 if a.dict_dtype == "f16" else np.float32)
    else:
        dic = rng.random((n, sy, sx), dtype=np.float32).astype(np.float16 if a.dict_dtype == "f16" else np.float32, copy=False)
        d_dic = ctx.dev_alloc(dic.nbytes)
        ctx.h2d(d_dic, dic)
    print(f"dictionary prefix of {n_need} patterns resident after {time.perf_counter() - t_gen:.1f} s", flush=True)
    t1 = None
    for tail in (True,) if (a.no_whole_tiles or a.pmc_shard or a.compute != "f32") else (True, False):
        if not tail:
            os.environ["KPDI_NO_TAIL"] = "1"
        for ranks in ranks_list:
            lo, hi = shard_range(n, 0, ranks)
            est = 2.0 * m * (hi - lo) * sy * sx / (1100e12 if a.compute == "f16" else 140e12)  # seconds per sweep
            reps = a.reps or int(max(3, min(20, 1.5 / est)))
            warm = 1 if a.pmc_shard else 3
            # the step is timed with events around the match launches only (an event record between two kernels idles the
            # GPU for ~6 us: 0.05 ms per step with every phase bracketed); the per-phase times come from untimed steps behind it
            ctx.set_profiling("match")
            pending = None
            for r in range(reps + warm):
                if r == warm:
                    if pending is not None:
                        ctx.finalize_wait(pending)
                        pending = None
                    ctx.reset_counters()
                    ctx.synchronize()
                    t0 = time.perf_counter()
                ctx.set_experimental_dev(d_exp, exp.dtype, m)
                ctx.push_dictionary_chunk_dev(d_dic, np.float16 if a.dict_dtype == "f16" else np.float32, hi - lo, lo)
                if a.pipeline:
                    ticket = ctx.finalize_async(keep)
                    if pending is not None:
                        ctx.finalize_wait(pending)
                    pending = ticket
                else:
                    ctx.finalize(keep)
            if pending is not None:
                ctx.finalize_wait(pending)
            dt = (time.perf_counter() - t0) / reps * 1e3
            c = ctx.counters()
            n_full = max(2, min(reps, 5))
            ctx.set_profiling(True)
            ctx.reset_counters()
            for r in range(n_full):
                ctx.set_experimental_dev(d_exp, exp.dtype, m)
                ctx.push_dictionary_chunk_dev(d_dic, np.float16 if a.dict_dtype == "f16" else np.float32, hi - lo, lo)
                ctx.finalize(keep)
            full = ctx.counters()
            for key in ("prep_ms", "merge_ms", "fixed_ms"):
                c[key] = full.get(key, 0.0) * reps / n_full
            ctx.set_profiling(False)
            if ranks == 1 and tail:
                t1 = dt
            key = f"{ranks}" + ("" if tail else "_whole_tiles_only")
            tile = 256 if c["match_form"] in (2, 3) else 128
            flops = c["match_flops"] / max(c["match_launches"], 1)
            mm = c["match_ms"] / max(c["match_launches"], 1)
            peak = 2500.0 if a.compute == "f16" else 157.3
            rec = {
                "shard_patterns": hi - lo, "tiles": -(-(hi - lo) // tile), "tile_patterns": tile,
                "kernel": {0: "match.hip (f32, 128 x 256 tiles)", 2: "match16.hip (float16)",
                           3: "match16.hip f32 form (256 x 256 tiles)"}.get(c["match_form"], str(c["match_form"])),
                "reps": reps, "ms_per_step": round(dt, 4),
                "match_ms": round(c["match_ms"] / reps, 4), "prep_ms": round(c["prep_ms"] / reps, 4),
                "merge_ms": round(c["merge_ms"] / reps, 4), "fixed_ms": round(c.get("fixed_ms", 0.0) / reps, 4),
                "other_ms": round(dt - (c["match_ms"] + c["prep_ms"] + c["merge_ms"] + c.get("fixed_ms", 0.0)) / reps, 4),
                "match_tflops": round(flops / (mm * 1e-3) / 1e12, 1) if mm > 0 else None,
                "match_frac_of_peak": round(flops / (mm * 1e-3) / 1e12 / peak, 4) if mm > 0 else None,
            }
            if t1 is not None:
                rec.update(even_share_ms=round(t1 / ranks, 4), step_over_even_share=round(dt / (t1 / ranks), 4),
                           efficiency_before_allgather=round((t1 / ranks) / dt, 4))
            out["ranks"][key] = rec
            print(key, rec, flush=True)
if a.out:
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
