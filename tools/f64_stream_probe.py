import sys, time, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kikuchipy_amd import _lib
rng = np.random.default_rng(1)
exp = rng.integers(0, 256, (4096, 60, 60), dtype=np.uint8)
dic = rng.random((100000, 60, 60), dtype=np.float32)
with _lib.Context(0) as c:
    c.set_problem(60, 60, None, _lib.METRIC_NCC, 20, _lib.COMPUTE_F64)
    for rep in range(3):
        c.set_experimental(exp)
        c.synchronize()
        t0 = time.perf_counter()
        for s in range(0, 100000, 12500):
            c.push_dictionary_chunk(dic[s:s + 12500], s)
        sc, ix = c.finalize(20)
        dt = time.perf_counter() - t0
        print(f"float64, 8 host chunks of 12 500: {dt*1e3:.1f} ms", c.counters()["uncertified_patterns"], flush=True)
