"""The stand-alone driver as a user calls it: kikuchipy_amd.dictionary_indexing(exp, dictionary in HOST memory, ...) at
configs[1], whole dictionary / the tutorial's chunking / a quarter; wall time of the call (best of 3), identical results.
    python tools/standalone_call_probe.py [out.txt]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kikuchipy_amd as kpa  # noqa: E402

rng = np.random.default_rng(2024)
exp = rng.integers(0, 256, (4096, 60, 60), dtype=np.uint8)
dic = rng.random((100000, 60, 60), dtype=np.float32)
lines = []
ref = None
for per in (None, 25000, 3044):
    best = None
    for rep in range(4):
        t0 = time.perf_counter()
        r = kpa.dictionary_indexing(exp, dic, metric="ncc", keep_n=20, n_per_iteration=per, device=0, progress=False) \
            if "progress" in kpa.dictionary_indexing.__code__.co_varnames else \
            kpa.dictionary_indexing(exp, dic, metric="ncc", keep_n=20, n_per_iteration=per, device=0)
        dt = time.perf_counter() - t0
        if rep and (best is None or dt < best):
            best = dt
    if ref is None:
        ref = r
    same = np.array_equal(ref.scores, r.scores) and np.array_equal(ref.simulation_indices, r.simulation_indices)
    lines.append(f"n_per_iteration={per}: {best * 1e3:.2f} ms per call = {4096 / best / 1e3:.1f} k patterns/s (1.44 GB over the host link: "
                 f"{1.44 / best:.1f} GB/s); identical to the single pass: {same}")
    print(lines[-1], flush=True)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
