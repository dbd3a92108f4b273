"""The plugin seam, pinned by data the reference itself produced.

`oracle/seam_check.py` (container only, python3.9 + the reference's unmodified `_dictionary_indexing`,
`_match_chunk` and `EBSD._prepare_metric`) drove this package's metric classes - mixed with the reference's ABC as
INTEGRATION.md section 1 shows - through every branch of the reference's loop, asserted equality with the
reference's stock metrics, asserted that `oracle.kpdi_oracle.plugin_loop` (the restatement of that loop) returns
exactly what the real loop returns, and stored the stock results in tests/golden/seam.npz.

Here the same metric classes go through `plugin_loop` and must reproduce those stored results: with the stand-in
engine (CPU, every round) and with the REAL engine (`-m gpu`) - single pass, chunked + host merge, a lazy dictionary
computed chunk by chunk inside the loop (indexing/_dictionary_indexing.py:106-108), navigation + signal masks,
float64, keep_n = 1, a last chunk shorter than keep_n."""

import numpy as np
import pytest

from conftest import load_golden, sha
from oracle import kpdi_oracle as ko

G = load_golden("seam.npz")
CASES = [str(c) for c in G["cases"]]


def case_arguments(name):
    """The inputs of oracle/seam_check.py::case_arguments, regenerated from the stored case definition and
    verified against the stored SHA-256."""
    metric, seed, nav_shape, n_dict, keep_n, n_it, lazy, masked, dtype = [str(v) for v in G[f"{name}__case"]]
    nav_shape = tuple(int(v) for v in nav_shape.strip("(),").replace(" ", "").split(",") if v)
    seed, n_dict, keep_n = int(seed), int(n_dict), int(keep_n)
    n_it = None if n_it == "None" else int(n_it)
    rng = np.random.default_rng(seed)
    exp = rng.integers(0, 256, nav_shape + (12, 10)).astype(np.uint8)
    dic = rng.random((n_dict, 12, 10)).astype(np.float32)
    assert sha(exp) == str(G[f"{name}__exp_sha"]) and sha(dic) == str(G[f"{name}__dic_sha"])
    sm = nm = None
    if masked == "True":
        sm = np.zeros((12, 10), dtype=bool)
        sm[:2] = True
        sm[5, 3:7] = True
        if len(nav_shape) == 2:
            nm = np.zeros(nav_shape, dtype=bool)
            nm[1, 2] = nm[4, 6] = nm[0, 0] = True
    return metric, exp, dic, keep_n, n_it, lazy == "True", sm, nm, np.dtype(dtype)


def run_case(name, make_context, score_atol):
    import kikuchipy_amd as kpa

    metric, exp, dic, keep_n, n_it, lazy, sm, nm, dtype = case_arguments(name)
    cls = {"ncc": kpa.NormalizedCrossCorrelationMetric, "ndp": kpa.NormalizedDotProductMetric}[metric]

    class HipMetric(cls):  # (INTEGRATION.md section 1 adds the reference's ABC as a second base: no behaviour of its own)
        pass

    n_per = n_it if n_it is not None else dic.shape[0]  # signals/ebsd.py:1925-1929
    m = ko.plugin_prepare_metric(HipMetric(context=make_context()), int(np.prod(exp.shape[:-2])), nm, sm, dtype, dic.shape[0])
    d_in = ko.LazyArray(dic, n_it) if lazy else dic
    dic_before = dic.copy()
    scores, idx, info = ko.plugin_loop(m, exp, exp.shape[:-2], d_in, keep_n, n_per)
    want_s, want_i = G[f"{name}__scores"], G[f"{name}__indices"]
    assert scores.shape == want_s.shape and scores.dtype == want_s.dtype == dtype
    assert idx.shape == want_i.shape and idx.dtype == want_i.dtype == np.int64
    k = min(keep_n, dic.shape[0])
    rows = slice(None) if nm is None else ~nm.ravel()
    ko.assert_topk_parity(scores.reshape(-1, k)[rows], idx.reshape(-1, k)[rows], want_s.reshape(-1, k)[rows],
                          want_i.reshape(-1, k)[rows], atol=score_atol)
    # the text the reference printed for its stock metric, the class name apart (:77-85, :206-237)
    stock = {"ncc": "NormalizedCrossCorrelationMetric", "ndp": "NormalizedDotProductMetric"}[metric]
    assert info == str(G[f"{name}__info"]).replace(stock, "HipMetric")
    assert repr(m) == str(G[f"{name}__repr"]).replace(stock, "HipMetric")
    assert np.array_equal(dic, dic_before)  # inputs are never modified (tests/test_indexing/test_dictionary_indexing.py:41-43)
    if lazy:  # every chunk was materialised exactly once, inside the loop
        assert d_in.computed == [(s, min(s + n_per, dic.shape[0])) for s in range(0, dic.shape[0], n_per)]
    return scores, idx


@pytest.mark.parametrize("name", CASES)
def test_seam_with_the_standin_engine(name):
    from _standin_engine import StandInContext

    run_case(name, lambda: StandInContext(0), 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_seam_with_the_gpu_engine(name):
    """north_star's bound (1e-5) for the float32 cases; the float64 case is float64 arithmetic on the device (the
    float32 screen + float64 rescoring of csrc/rescore.hip): 1e-12 against the reference's float64 metric."""
    from kikuchipy_amd import _lib

    ctxs = []

    def make():
        ctxs.append(_lib.Context(0))
        return ctxs[-1]

    try:
        run_case(name, make, 1e-12 if "float64" in name else 1e-5)
    finally:
        for c in ctxs:
            c.close()
