"""What the profiling events cost: one rank's share of configs[1] at N = 8 (4096 x 12 500 x 60 x 60, inputs resident), the
pipelined loop of bench.py, with `set_profiling` on and off - wall time per step.   python tools/probes/event_cost.py [n] [level ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kikuchipy_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
levels = [int(v) for v in sys.argv[2:]] or [0, 1, 0, 1]
rng = np.random.default_rng(3)
exp = rng.integers(0, 256, (4096, 60, 60), dtype=np.uint8)
dic = rng.random((n, 60, 60), dtype=np.float32)
with _lib.Context(0) as ctx:
    ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 20)
    d_exp = ctx.dev_alloc(exp.nbytes); ctx.h2d(d_exp, exp)
    d_dic = ctx.dev_alloc(dic.nbytes); ctx.h2d(d_dic, dic)
    for prof in levels:
        ctx.set_profiling(prof)
        for reps in (20, 200):
            ctx.reset_counters(); ctx.synchronize()
            t0 = time.perf_counter(); pending = None
            for _ in range(reps):
                ctx.set_experimental_dev(d_exp, exp.dtype, 4096)
                ctx.push_dictionary_chunk_dev(d_dic, np.float32, n, 0)
                t = ctx.finalize_async(20)
                if pending is not None: ctx.finalize_wait(pending)
                pending = t
            ctx.finalize_wait(pending); ctx.synchronize()
            dt = (time.perf_counter() - t0) / reps * 1e3
        print(f"profiling level {prof}: {dt:.4f} ms per step ({n} dictionary patterns)", flush=True)
