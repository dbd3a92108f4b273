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


# ---- the look-ahead of the plugin (similarity_metrics._LookAhead): the chunks the reference's loop is about to ask for are
# swept ahead of it; anything else cancels it.  CPU, stand-in engine.
def _lookahead_metric(n_exp, n_dict, keep_n=5):
    import kikuchipy_amd as kpa
    from _standin_engine import StandInContext

    m = kpa.NormalizedCrossCorrelationMetric(context=StandInContext(0))
    return ko.plugin_prepare_metric(m, n_exp, None, None, np.dtype("float32"), n_dict)


def test_lookahead_serves_the_chunks_of_the_loop_and_changes_nothing(monkeypatch):
    rng = np.random.default_rng(5)
    exp = rng.integers(0, 256, (6, 12, 10)).astype(np.uint8)
    dic = rng.random((103, 12, 10)).astype(np.float32)  # 11 chunks of 10 rows; the last one holds 3 < keep_n
    out = {}
    for on in ("1", "0"):
        monkeypatch.setenv("KPDI_SEAM_LOOKAHEAD", on)
        m = _lookahead_metric(6, 103)
        s, i, _ = ko.plugin_loop(m, exp, (6,), dic, 5, 10)
        out[on] = (s, i, m.lookahead_hits)
        assert m._lookahead is None  # (the worker has left when the last chunk was handed over)
        m.close()
    assert out["1"][2] == 10 and out["0"][2] == 0  # every chunk but the first came from the look-ahead
    assert np.array_equal(out["1"][0], out["0"][0]) and np.array_equal(out["1"][1], out["0"][1])
    # a second call of the loop on the same metric starts over (prepare_experimental resets the row count)
    monkeypatch.setenv("KPDI_SEAM_LOOKAHEAD", "1")
    m = _lookahead_metric(6, 103)
    a = ko.plugin_loop(m, exp, (6,), dic, 5, 10)
    b = ko.plugin_loop(m, exp, (6,), dic, 5, 25)
    assert m.lookahead_hits == 10 + 4 and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    m.close()


def test_lookahead_is_dropped_when_the_loop_asks_for_something_else():
    rng = np.random.default_rng(6)
    exp = rng.integers(0, 256, (4, 12, 10)).astype(np.uint8)
    dic = rng.random((60, 12, 10)).astype(np.float32)
    m = _lookahead_metric(4, 60)
    e = m.prepare_experimental(exp)
    flat = dic.reshape((60, -1))

    def best(rows, k=3):
        sim = m.match(e, m.prepare_dictionary(rows))
        return sim.topk(k, axis=-1), sim.argtopk(k, axis=-1)

    first = best(flat[0:10])
    assert m._lookahead is not None and m.lookahead_hits == 0
    skipped = best(flat[30:40])          # not the adjacent chunk: cancelled, served the ordinary way ...
    assert m.lookahead_hits == 0 and m._lookahead is None  # ... and nothing more is guessed during this call of the loop
    other_k = best(flat[40:50], k=2)
    copy = best(flat[10:20].copy())      # (same values, another buffer)
    assert m._lookahead is None and m.lookahead_hits == 0
    e = m.prepare_experimental(exp)      # the next call of the loop starts over
    again = best(flat[0:10])
    assert m._lookahead is not None
    nxt = best(flat[10:20])
    assert m.lookahead_hits == 1 and np.array_equal(again[0], first[0]) and np.array_equal(nxt[0], copy[0])
    m.close()
    # each result is what a fresh metric returns for that chunk alone
    for rows, k, got in ((flat[0:10], 3, first), (flat[30:40], 3, skipped), (flat[40:50], 2, other_k), (flat[10:20], 3, copy)):
        m2 = _lookahead_metric(4, 60)
        e2 = m2.prepare_experimental(exp)
        sim = m2.match(e2, m2.prepare_dictionary(rows.copy()))
        assert np.array_equal(sim.topk(k, axis=-1), got[0]) and np.array_equal(sim.argtopk(k, axis=-1), got[1])
        m2.close()


def test_lookahead_never_serves_rows_the_caller_has_rewritten():
    """a CUSTOM loop that refills a pre-allocated dictionary buffer between `match()` calls must not be
    served what the look-ahead swept before the rows changed: every chunk is fingerprinted when the worker reads it and
    checked against what `match()` is handed; on a mismatch the chunk is swept the ordinary way and nothing more is
    guessed during that call."""
    rng = np.random.default_rng(8)
    exp = rng.integers(0, 256, (4, 12, 10)).astype(np.uint8)
    buf = rng.random((40, 12, 10)).astype(np.float32)
    m = _lookahead_metric(4, 40)
    e = m.prepare_experimental(exp)
    flat = buf.reshape((40, -1))

    def best(rows, k=3):
        sim = m.match(e, m.prepare_dictionary(rows))
        return sim.topk(k, axis=-1), sim.argtopk(k, axis=-1)

    best(flat[0:10])
    la = m._lookahead
    assert la is not None
    while la._results.qsize() < 1:   # the worker has swept rows [10, 20) (as they were) ...
        pass
    flat[10:20] = rng.random((10, 120)).astype(np.float32)  # ... and now the caller refills them
    got = best(flat[10:20])
    assert m.lookahead_hits == 0 and m._lookahead is None and m._lookahead_off
    m2 = _lookahead_metric(4, 40)
    e2 = m2.prepare_experimental(exp)
    sim = m2.match(e2, m2.prepare_dictionary(flat[10:20].copy()))
    assert np.array_equal(sim.topk(3, axis=-1), got[0]) and np.array_equal(sim.argtopk(3, axis=-1), got[1])
    nxt = best(flat[20:30])          # the rest of this call is served the ordinary way
    assert m.lookahead_hits == 0 and m._lookahead is None
    m.close()
    m2.close()
    assert nxt[0].shape == (4, 3)


def test_lookahead_stays_inside_the_callers_buffer_and_ends_with_close():
    rng = np.random.default_rng(7)
    exp = rng.integers(0, 256, (3, 12, 10)).astype(np.uint8)
    big = rng.random((50, 12, 10)).astype(np.float32)
    m = _lookahead_metric(3, 40)         # the metric is told of 40 dictionary rows ...
    e = m.prepare_experimental(exp)
    tail = big[30:].reshape((20, -1))    # ... but this buffer ends 20 rows behind the first chunk's start
    m.match(e, m.prepare_dictionary(tail[0:10])).topk(2, axis=-1)
    assert m._lookahead is None          # rows [10, 40) of "the dictionary" would lie outside `big`: nothing is predicted
    m2 = _lookahead_metric(3, 40)
    e2 = m2.prepare_experimental(exp)
    m2.match(e2, m2.prepare_dictionary(big.reshape((50, -1))[0:10])).topk(2, axis=-1)
    la = m2._lookahead
    assert la is not None                # running ahead over rows [10, 40) ...
    m2.close()                           # ... until the loop is abandoned: the worker leaves the engine
    assert not la._thread.is_alive() and m2._lookahead is None


def test_lookahead_hands_a_failure_to_the_chunk_it_belongs_to():
    """A sweep that fails on the look-ahead's thread is raised in the caller's thread when the loop asks for THAT chunk -
    the chunks before it are served - and the engine is usable afterwards (no result slot left taken)."""
    import kikuchipy_amd as kpa
    from _standin_engine import StandInContext

    class Failing(StandInContext):
        pushes = 0

        def push_dictionary_chunk(self, patterns, global_start):
            Failing.pushes += 1
            if Failing.pushes == 3:
                raise RuntimeError("third chunk refused")
            return super().push_dictionary_chunk(patterns, global_start)

    rng = np.random.default_rng(8)
    exp = rng.integers(0, 256, (3, 12, 10)).astype(np.uint8)
    dic = rng.random((50, 12, 10)).astype(np.float32)
    flat = dic.reshape((50, -1))
    m = ko.plugin_prepare_metric(kpa.NormalizedCrossCorrelationMetric(context=Failing(0)), 3, None, None, np.dtype("float32"), 50)
    e = m.prepare_experimental(exp)
    best = lambda rows: m.match(e, m.prepare_dictionary(rows)).topk(2, axis=-1)  # noqa: E731
    best(flat[0:10])
    best(flat[10:20])                       # from the look-ahead
    with pytest.raises(RuntimeError, match="third chunk refused"):
        best(flat[20:30])                   # the worker's failure, here
    assert m.lookahead_hits == 1 and m._lookahead is None
    e = m.prepare_experimental(exp)         # the engine still works
    assert best(flat[0:10]).shape == (3, 2)
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("metric_name,dtype", [("ncc", "float32"), ("ndp", "float32"), ("ncc", "float64")])
def test_lookahead_on_the_gpu_engine_serves_the_loop_and_changes_nothing(monkeypatch, metric_name, dtype):
    """The real engine behind the plugin in the (restated) reference loop: 8 chunks of 700 rows + one of 100, a signal
    mask, keep_n = 20 - every chunk but the first comes from the look-ahead (pipelined finalize_async / finalize_wait on
    the worker thread; float64 arithmetic: one synchronous sweep at a time there), bit for bit what the same loop returns
    with the look-ahead switched off."""
    import kikuchipy_amd as kpa

    rng = np.random.default_rng(11)
    exp = rng.integers(0, 256, (300, 20, 20)).astype(np.uint8)
    dic = rng.random((5700, 20, 20)).astype(np.float32)
    sm = np.zeros((20, 20), dtype=bool)
    sm[:3] = True
    cls = {"ncc": kpa.NormalizedCrossCorrelationMetric, "ndp": kpa.NormalizedDotProductMetric}[metric_name]
    out = {}
    for on in ("1", "0"):
        monkeypatch.setenv("KPDI_SEAM_LOOKAHEAD", on)
        m = ko.plugin_prepare_metric(cls(device=0), 300, None, sm, np.dtype(dtype), 5700)
        try:
            s_, i_, _ = ko.plugin_loop(m, exp, (300,), dic, 20, 700)
            out[on] = (s_, i_, m.lookahead_hits)
        finally:
            m.close()
    assert out["1"][2] == 8 and out["0"][2] == 0
    assert np.array_equal(out["1"][0], out["0"][0]) and np.array_equal(out["1"][1], out["0"][1])
    # ... and what the oracle returns for the same call: north_star's 1e-5 (float64 arithmetic: 1e-12, one chunk at a time
    # on the look-ahead's thread - no pipelined hand-over there)
    assert out["1"][0].dtype == np.dtype(dtype)
    rs, ri = ko.dictionary_indexing(exp, dic, metric_name, 20, 700, None, sm, np.dtype(dtype).type)
    ko.assert_topk_parity(out["1"][0], out["1"][1], rs, ri, atol=1e-5 if dtype == "float32" else 1e-12,
                          tie=2e-5 if dtype == "float32" else 1e-11)
