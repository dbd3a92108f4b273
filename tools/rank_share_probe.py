"""One rank's share of a dictionary-sharded job on ONE GPU: for N = 1, 2, 4, 8 ranks the sweep of rank 0's
shard (shard_range(n, 0, N)) of configs[1] is timed with the inputs resident in HBM, and compared with
the even share t_1 / N - what strong scaling over N GPUs can reach before the RCCL all-gather (a few
hundred microseconds for 4096 x 20 x 8 B per rank).   python tools/rank_share_probe.py [out.json]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kikuchipy_amd import _lib  # noqa: E402
from kikuchipy_amd.parallel import shard_range  # noqa: E402

m, n, sy, sx, keep = 4096, 100000, 60, 60, 20
rng = np.random.default_rng(2024)
exp = rng.integers(0, 256, (m, sy, sx), dtype=np.uint8)
dic = rng.random((n, sy, sx), dtype=np.float32)
out = {"workload": "configs[1]: 4096 x 100 000 x 60 x 60, ncc, keep_n = 20; rank 0's shard on one MI355X", "ranks": {}}
with _lib.Context(0) as ctx:
    ctx.set_problem(sy, sx, None, _lib.METRIC_NCC, keep)
    d_exp = ctx.dev_alloc(exp.nbytes)
    ctx.h2d(d_exp, exp)
    d_dic = ctx.dev_alloc(dic.nbytes)
    ctx.h2d(d_dic, dic)
    t1 = None
    for tail in (True, False):
        if not tail:
            os.environ["KPDI_NO_TAIL"] = "1"
        for ranks in [int(x) for x in os.environ.get("RANKS", "1,2,4,8,16").split(",")]:
            lo, hi = shard_range(n, 0, ranks)
            ctx.set_profiling(True)
            reps = 20
            for r in range(reps + 3):
                if r == 3:
                    ctx.reset_counters()
                    ctx.synchronize()
                    t0 = time.perf_counter()
                ctx.set_experimental_dev(d_exp, exp.dtype, m)
                ctx.push_dictionary_chunk_dev(d_dic, np.float32, hi - lo, lo)
                ctx.finalize(keep)
            dt = (time.perf_counter() - t0) / reps * 1e3
            c = ctx.counters()
            ctx.set_profiling(False)
            if ranks == 1 and tail:
                t1 = dt
            if t1 is None:
                t1 = float(os.environ.get("T1_MS", "21.97"))
            key = f"{ranks}" + ("" if tail else "_whole_tiles_only")
            out["ranks"][key] = {
                "shard_patterns": hi - lo, "tiles": -(-(hi - lo) // 128), "ms_per_step": round(dt, 4),
                "match_ms": round(c["match_ms"] / reps, 4), "even_share_ms": round(t1 / ranks, 4),
                "step_over_even_share": round(dt / (t1 / ranks), 4),
                "efficiency_before_allgather": round((t1 / ranks) / dt, 4),
            }
            print(key, out["ranks"][key], flush=True)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        json.dump(out, f, indent=1)
