"""Pin the plugin seam against the reference's REAL loop (container only).

TEST INFRASTRUCTURE - never imported by the product (kikuchipy_amd/).  Run with

    /opt/conda/bin/python3.9 -W ignore oracle/seam_check.py          # checks, then writes tests/golden/seam.npz

What INTEGRATION.md section 1 claims is that a maintainer can hand this package's metrics to an UNMODIFIED
kikuchipy: `class HipNCCMetric(kpa.NormalizedCrossCorrelationMetric, kikuchipy.indexing.SimilarityMetric)` passes
`EBSD._prepare_metric`'s `isinstance` gate (signals/ebsd.py:3065-3070) and is then driven by
`_dictionary_indexing` (indexing/_dictionary_indexing.py:36-169) and `_match_chunk` (:172-203).  tests/ can only
hold a restatement of that loop (the reference does not travel to the GPU box); THIS script runs the reference's own
code - both functions loaded unmodified by oracle/ref_shim.py, `_prepare_metric` executed from the source of
signals/ebsd.py - with the mixed-in metric, engine = tests/_standin_engine.StandInContext (the oracle behind the
`_lib.Context` interface: there is no GPU in this container), over every branch of the loop:

    single pass (`dictionary_size == n_per_iteration`, :88-93) . chunked with the host merge (:94-128) .
    a lazy `dask.array` dictionary whose chunks are `.compute()`d inside the loop (:106-108) .
    navigation + signal mask . dtype=float64 . keep_n = 1 . a last chunk shorter than keep_n

and asserts, per case, that the result equals what the reference's STOCK metric gives in the same loop (indices
equal outside near-ties, scores within 1e-6: the stand-in engine is float32 NumPy), that the reference printed the
same information text, and that `repr(metric)` is the stock metric's with the class name swapped.  The stock results
and the case definitions (inputs by seed) go to tests/golden/seam.npz; `tests/test_gpu_seam.py` reproduces them on
the GPU with the REAL engine behind the same metric classes, driven through the restated loop - so the chain
reference loop == restated loop == GPU engine is closed by data that the reference itself produced.
"""

import ast
import contextlib
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __future__  # noqa: E402

import ref_shim  # noqa: E402

ref = ref_shim.load_reference()
RefABC = ref["similarity_metric"].SimilarityMetric
RefNCC = ref["ncc"].NormalizedCrossCorrelationMetric
RefNDP = ref["ndp"].NormalizedDotProductMetric
di = ref["di"]

import dask.array as da  # noqa: E402

import kikuchipy_amd as kpa  # noqa: E402
from _standin_engine import StandInContext  # noqa: E402


# ---- INTEGRATION.md section 1, verbatim: this package's metric first, the reference's ABC mixed in
class HipNCCMetric(kpa.NormalizedCrossCorrelationMetric, RefABC):
    pass


class HipNDPMetric(kpa.NormalizedDotProductMetric, RefABC):
    pass


def reference_prepare_metric():
    """`EBSD._prepare_metric` (signals/ebsd.py:3049-3088) compiled from the reference's source file - the module as a
    whole needs hyperspy / orix - with the names it uses bound to the reference's own classes."""
    path = os.path.join(ref_shim.SRC, "signals", "ebsd.py")
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == "EBSD":
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name == "_prepare_metric":
                    mod = ast.Module(body=[item], type_ignores=[])
                    code = compile(mod, path, "exec", flags=__future__.annotations.compiler_flag, dont_inherit=True)
                    g = {"np": np, "SimilarityMetric": RefABC, "NormalizedCrossCorrelationMetric": RefNCC,
                         "NormalizedDotProductMetric": RefNDP}
                    exec(code, g)
                    return g["_prepare_metric"]
    raise KeyError("EBSD._prepare_metric")


PREPARE_METRIC = reference_prepare_metric()


def run_reference_loop(metric, exp, dic, keep_n, n_per_iteration, navigation_mask, signal_mask, dtype):
    """The reference's `EBSD._prepare_metric` + `_dictionary_indexing`, nothing restated."""
    nav_shape = exp.shape[:-2]
    signal = types.SimpleNamespace(axes_manager=types.SimpleNamespace(navigation_size=int(np.prod(nav_shape))))
    metric = PREPARE_METRIC(signal, metric, navigation_mask, signal_mask, dtype, False, dic.shape[0])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
        xmap = di._dictionary_indexing(
            experimental=exp, experimental_nav_shape=nav_shape, dictionary=dic, step_sizes=(1,) * len(nav_shape),
            dictionary_xmap=ref_shim.FakeDictionaryXmap(), metric=metric, keep_n=keep_n, n_per_iteration=n_per_iteration)
    prop = xmap.kw["prop"]
    text = buf.getvalue()
    # (the information block of :77-85; what follows is Dask's progress bar - stock metric only - and the speed line)
    info = text[:text.index("\n", text.index("signal mask:")) + 1]
    return np.asarray(prop["scores"]), np.asarray(prop["simulation_indices"]), info, repr(metric), metric


def inputs(seed, nav_shape, n_dict, sig=(12, 10)):
    """Seeded inputs (regenerated by tests/test_gpu_seam.py from the same seed)."""
    rng = np.random.default_rng(seed)
    exp = rng.integers(0, 256, nav_shape + sig).astype(np.uint8)
    dic = rng.random((n_dict,) + sig).astype(np.float32)
    return exp, dic


def masks(sig=(12, 10), nav_shape=(6, 7)):
    sm = np.zeros(sig, dtype=bool)
    sm[:2] = True
    sm[5, 3:7] = True
    nm = np.zeros(nav_shape, dtype=bool)
    nm[1, 2] = nm[4, 6] = nm[0, 0] = True
    return sm, nm


# name -> (metric, seed, nav_shape, n_dict, keep_n, n_per_iteration, lazy, masked, dtype)
CASES = {
    "ncc_single_pass": ("ncc", 11, (6, 7), 1000, 20, None, False, False, "float32"),
    "ndp_chunked": ("ndp", 12, (6, 7), 1000, 10, 300, False, False, "float32"),
    "ncc_lazy_dictionary": ("ncc", 13, (6, 7), 1000, 8, 250, True, False, "float32"),
    "ncc_masks_chunked": ("ncc", 14, (6, 7), 900, 12, 400, False, True, "float32"),
    "ndp_masks_single_pass": ("ndp", 15, (6, 7), 500, 5, None, False, True, "float32"),
    "ncc_float64_chunked": ("ncc", 16, (6, 7), 800, 6, 350, False, False, "float64"),
    "ndp_keep1_navmask_lazy": ("ndp", 17, (6, 7), 600, 1, 200, True, True, "float32"),
    "ncc_short_last_chunk": ("ncc", 18, (5,), 203, 5, 100, False, False, "float32"),  # last chunk: 3 < keep_n patterns
}


def case_arguments(name):
    metric, seed, nav_shape, n_dict, keep_n, n_it, lazy, masked, dtype = CASES[name]
    exp, dic = inputs(seed, nav_shape, n_dict)
    sm = nm = None
    if masked:
        sm, nm = masks(nav_shape=nav_shape) if len(nav_shape) == 2 else (masks()[0], None)
    return metric, exp, dic, keep_n, n_it, lazy, sm, nm, np.dtype(dtype)


def main():
    out = {"cases": np.array(sorted(CASES))}
    hip = {"ncc": HipNCCMetric, "ndp": HipNDPMetric}
    for name in sorted(CASES):
        metric, exp, dic, keep_n, n_it, lazy, sm, nm, dtype = case_arguments(name)
        d_in = da.from_array(dic, chunks=(n_it,) + dic.shape[1:]) if lazy else dic
        n_per = n_it if n_it is not None else dic.shape[0]  # signals/ebsd.py:1925-1929
        rs, ri, rinfo, rrepr, _ = run_reference_loop(metric, exp, d_in, keep_n, n_per, nm, sm, dtype)
        engine = StandInContext(0)
        m = hip[metric](context=engine)
        assert isinstance(m, RefABC) and isinstance(m, kpa.SimilarityMetric)
        hs, hi, hinfo, hrepr, m = run_reference_loop(m, exp, d_in, keep_n, n_per, nm, sm, dtype)
        # the reference's loop, our metric: same answer as its stock metric
        assert hs.shape == rs.shape and hi.shape == ri.shape and hs.dtype == rs.dtype, (name, hs.shape, rs.shape, hs.dtype)
        assert hi.dtype == ri.dtype == np.int64, (name, hi.dtype, ri.dtype)
        rows = slice(None) if nm is None else ~nm.ravel()
        k = min(keep_n, dic.shape[0])
        as2d = lambda x: x.reshape(-1, k)  # noqa: E731  (keep_n == 1 with a navigation mask is squeezed to 1-D, :155-158)
        a_s, a_i, b_s, b_i = as2d(hs)[rows], as2d(hi)[rows], as2d(rs)[rows], as2d(ri)[rows]
        assert np.abs(a_s - b_s).max() <= 1e-6, (name, np.abs(a_s - b_s).max())
        gap = np.abs(np.diff(b_s.astype(np.float64), axis=1)).min(axis=1) if k > 1 else np.ones(len(b_s))
        clear = gap > 4e-6  # rows without a near-tie must agree index for index
        assert np.array_equal(a_i[clear], b_i[clear]), name
        if nm is not None:
            # masked-out points: `np.empty` in the reference (:149-153) - no contract; zeroed in the fixture so that it
            # regenerates byte for byte
            rs, ri = rs.copy(), ri.copy()
            as2d(rs)[nm.ravel()] = 0
            as2d(ri)[nm.ravel()] = 0
        # the text the reference printed, and repr(metric): the stock one's with the class name swapped
        stock_name = {"ncc": "NormalizedCrossCorrelationMetric", "ndp": "NormalizedDotProductMetric"}[metric]
        assert hinfo == rinfo.replace(stock_name, type(m).__name__), (name, hinfo, rinfo)
        assert hrepr == rrepr.replace(stock_name, type(m).__name__), (name, hrepr, rrepr)
        # ... and the RESTATED loop (oracle/kpdi_oracle.py: plugin_loop, what tests/ can run on the GPU box) gives exactly
        # what the reference's own loop gives for the same metric object and engine - with NumPy and with lazy dictionaries
        import kpdi_oracle as ko

        m2 = ko.plugin_prepare_metric(hip[metric](context=StandInContext(0)), int(np.prod(exp.shape[:-2])), nm, sm, dtype, dic.shape[0])
        lazy_in = ko.LazyArray(dic, n_it) if lazy else dic
        ps, pi, pinfo = ko.plugin_loop(m2, exp, exp.shape[:-2], lazy_in, keep_n, n_per)
        keep_rows = np.ones(as2d(hs).shape[0], dtype=bool) if nm is None else ~nm.ravel()
        assert ps.shape == hs.shape and ps.dtype == hs.dtype and pi.dtype == hi.dtype, (name, ps.shape, hs.shape, ps.dtype, pi.dtype)
        assert np.array_equal(as2d(ps)[keep_rows], as2d(hs)[keep_rows]) and np.array_equal(as2d(pi)[keep_rows], as2d(hi)[keep_rows]), name
        assert pinfo == hinfo, (name, pinfo, hinfo)
        if lazy:
            assert lazy_in.computed == [(c * n_per, min((c + 1) * n_per, dic.shape[0])) for c in range(int(np.ceil(dic.shape[0] / n_per)))]
        # the engine was driven as the seam promises: prepared once, one push per chunk, chunk starts 0 (the loop adds them)
        n_chunks = int(np.ceil(dic.shape[0] / n_per))
        assert len(engine.pushed) == n_chunks and all(start == 0 for start, _ in engine.pushed), (name, engine.pushed)
        # the plugin's look-ahead (similarity_metrics._LookAhead) predicts the REAL loop's slices: every chunk of a NumPy
        # dictionary but the first was swept ahead - also of the lazy case here, whose `dask.array.from_array` over an
        # in-memory array hands out VIEWS of that array from `.compute()` (a dictionary that is really computed arrives
        # in fresh arrays and is not predicted); float64 arithmetic too (one chunk at a time: it has no pipelined hand-over)
        want_hits = n_chunks - 1
        assert m.lookahead_hits == want_hits, (name, m.lookahead_hits, want_hits)
        print(f"{name}: {n_chunks} chunk(s), {m.lookahead_hits} served from the look-ahead")
        assert [n for _, n in engine.pushed] == [min(n_per, dic.shape[0] - c * n_per) for c in range(n_chunks)], name
        out[f"{name}__scores"] = rs
        out[f"{name}__indices"] = ri
        out[f"{name}__info"] = np.array(rinfo)
        out[f"{name}__repr"] = np.array(rrepr)
        out[f"{name}__exp_sha"] = np.array(sha(exp))
        out[f"{name}__dic_sha"] = np.array(sha(dic))
        print(f"seam OK  {name:28s} {rs.shape} {rs.dtype}  max |d score| vs stock metric {np.abs(a_s - b_s).max():.1e}")
    # the gate itself: what is NOT an instance of the reference's ABC is refused by the reference's _prepare_metric
    try:
        signal = types.SimpleNamespace(axes_manager=types.SimpleNamespace(navigation_size=4))
        PREPARE_METRIC(signal, kpa.NormalizedCrossCorrelationMetric(context=StandInContext(0)), None, None, None, False, 10)
    except ValueError as e:
        assert "inheriting from SimilarityMetric" in str(e)
    else:
        raise AssertionError("the un-mixed metric must not pass the reference's isinstance gate")
    cases = {n: np.array([str(v) for v in CASES[n]]) for n in CASES}
    for n, v in cases.items():
        out[f"{n}__case"] = v
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "seam.npz"), **out)
    print("wrote tests/golden/seam.npz:", len(CASES), "cases")


def sha(a):
    import hashlib

    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


if __name__ == "__main__":
    main()
