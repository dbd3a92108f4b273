"""Is the CPU timing port (oracle/cpu_port.py) a fair stand-in for the reference on a host?
CONTAINER ONLY (needs /root/reference):

    /opt/conda/bin/python3.9 -W ignore tools/cpu_baseline_crosscheck.py [M N]

Runs the REFERENCE's own `_dictionary_indexing` (modules loaded unmodified through
oracle/ref_shim.py, as for the golden vectors) and the port on the same inputs, in the same
interpreter (same NumPy, same BLAS, same threads), interleaved, and prints both wall times.
BASELINE.md section 2 workload by default: 4096 x 20 000 x 60 x 60, ncc, keep_n=20,
n_per_iteration=2000."""
import contextlib
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_shim  # noqa: E402

ref = ref_shim.load_reference()
from oracle import cpu_port  # noqa: E402

m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 20000)
rng = np.random.default_rng(0)
exp = rng.integers(0, 256, (m, 60, 60)).astype(np.uint8)
dic = rng.random((n, 60, 60)).astype(np.float32)


def run_reference():
    metric = ref["ncc"].NormalizedCrossCorrelationMetric()
    metric.n_experimental_patterns, metric.n_dictionary_patterns, metric.dtype = m, n, np.float32
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        xmap = ref["di"]._dictionary_indexing(
            experimental=exp, experimental_nav_shape=(m,), dictionary=dic, step_sizes=(1,),
            dictionary_xmap=ref_shim.FakeDictionaryXmap(), metric=metric, keep_n=20, n_per_iteration=2000)
    return np.asarray(xmap.kw["prop"]["scores"]), np.asarray(xmap.kw["prop"]["simulation_indices"])


times = {"reference": [], "port": []}
for rep in range(3):
    t = time.perf_counter()
    rs, ri = run_reference()
    times["reference"].append(time.perf_counter() - t)
    t = time.perf_counter()
    ps, pi = cpu_port.dictionary_indexing(exp, dic, "ncc", 20, 2000)
    times["port"].append(time.perf_counter() - t)
assert np.allclose(ps, rs, atol=1e-6) and np.mean(pi == ri) > 0.999
tr, tp = min(times["reference"]), min(times["port"])
print(f"numpy {np.__version__}, {os.cpu_count()} cpus, BLAS threads {cpu_port.blas_threads()}")
print(f"reference: {tr:.2f} s = {m / tr:.1f} patterns/s   (runs: {[round(x, 2) for x in times['reference']]})")
print(f"port     : {tp:.2f} s = {m / tp:.1f} patterns/s   (runs: {[round(x, 2) for x in times['port']]})")
print(f"port / reference time = {tp / tr:.3f}")
