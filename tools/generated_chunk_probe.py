"""A dictionary SIMULATED on the device chunk by chunk (`get_patterns(..., chunk_shape=n)`: the lazy dictionary of the
reference's tutorial) at configs[1]'s sizes on ONE GPU: ms per `dictionary_indexing` call for a few chunk sizes, against
the single-chunk call.  `KPDI_NO_COALESCE=1` shows rounds 1-4 (every chunk swept on arrival).

    python tools/generated_chunk_probe.py [out.txt]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kikuchipy_amd as ka  # noqa: E402

rng = np.random.default_rng(7)
mp = ka.EBSDMasterPattern(rng.random((2, 401, 401)).astype(np.float32), hemisphere="both")
det = ka.EBSDDetector(shape=(60, 60), pc=(0.421, 0.7794, 0.5049))
q = rng.standard_normal((100000, 4))
q /= np.linalg.norm(q, axis=1)[:, None]
exp = rng.integers(0, 256, (64, 64, 60, 60), dtype=np.uint8)
lines = []
ref = None
with ka.EBSD(exp, device=0) as s:
    for chunk in (100000, 25000, 10000, 3044, 1000):
        sim = mp.get_patterns(q, det, compute=False, chunk_shape=chunk)
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            res = s.dictionary_indexing(sim, keep_n=20, verbose=False)
            best = min(best, time.perf_counter() - t0)
        if ref is None:
            ref = res
        same = bool(np.array_equal(res.scores, ref.scores) and np.array_equal(res.simulation_indices, ref.simulation_indices))
        lines.append(f"chunk_shape {chunk:6d} ({-(-100000 // chunk):3d} chunks): {best * 1e3:7.2f} ms per call = {4096 / best / 1e3:6.1f} k patterns/s; "
                     f"== the single-chunk result bit for bit: {same}")
        print(lines[-1], flush=True)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        f.write("4096 experimental patterns x 100 000 dictionary patterns SIMULATED on the device (60 x 60, 401 x 401 master pattern), one MI355X, "
                "best of 4 calls" + ("  [KPDI_NO_COALESCE=1]" if os.environ.get("KPDI_NO_COALESCE") else "") + "\n" + "\n".join(lines) + "\n")
