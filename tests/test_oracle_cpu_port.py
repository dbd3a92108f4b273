"""The CPU TIMING ports behind bench.py's `cpu_baseline` (oracle/cpu_port.py: the reference's
executed operations in NumPy; oracle/kpdi_oracle_c.c: C + OpenMP) return the oracle's results."""
import numpy as np
import pytest

from oracle import c_oracle, cpu_port
from oracle import kpdi_oracle as ko


@pytest.mark.parametrize("metric,masked", [("ncc", False), ("ncc", True), ("ndp", False)])
def test_ports_equal_the_oracle(synth_inputs, metric, masked):
    exp, dic, g = synth_inputs
    mask = ~ko.circular_window((60, 60)).astype(bool) if masked else None
    rs, ri = ko.dictionary_indexing(exp, dic, metric=metric, keep_n=20, n_per_iteration=700, signal_mask=mask)
    s, i = cpu_port.dictionary_indexing(exp, dic, metric, 20, 700, mask)
    ko.assert_topk_parity(s, i, rs, ri, atol=1e-5)
    s, i, _ = cpu_port.run_parallel(exp, dic, 2, metric=metric, keep_n=20, n_per_iteration=700, signal_mask=mask)
    ko.assert_topk_parity(s, i, rs, ri, atol=1e-5)
    s, i = c_oracle.openmp_port(exp, dic, metric, 20, 700, mask)
    ko.assert_topk_parity(s, i, rs, ri, atol=2e-5)
    # and the float64 row checker against both
    rows = np.arange(0, len(exp), 5)
    s, i = c_oracle.rows_topk_f64(exp, dic, rows, metric, 20, mask)
    ko.assert_topk_parity(s, i, rs[rows], ri[rows], atol=1e-5)


def test_dask_topk_semantics():
    """k largest, descending (dask/array/chunk.py:167-258 as used at indexing/_dictionary_indexing.py:197-198)."""
    rng = np.random.default_rng(0)
    a = rng.random((50, 300)).astype(np.float32)
    assert np.array_equal(cpu_port.topk_dask(a, 7), np.sort(a, axis=1)[:, ::-1][:, :7])
    assert np.array_equal(np.take_along_axis(a, cpu_port.argtopk_dask(a, 7), axis=1), cpu_port.topk_dask(a, 7))
