"""CPU TIMING port of the reference's dictionary-indexing loop - TEST / BENCH INFRASTRUCTURE.

`oracle/kpdi_oracle.py` is the correctness oracle: it ranks with a full stable sort so that ties
come out by the engine's rule, which makes it a poor stopwatch.  This module mirrors what the
reference EXECUTES, operation for operation, so that its wall time stands for the reference's on
the same host (only bench.py's `cpu_baseline` leg and tools/ use it):

  prepare_experimental / prepare_dictionary   similarity_metrics/_normalized_cross_correlation.py:88-159
                                              (astype float32 -> reshape -> mask -> mean/norm, NumPy)
  match                                       ..._cross_correlation.py:161-183  einsum("ik,mk->im") -> sgemm
  argtopk + topk                              indexing/_dictionary_indexing.py:197-198 -> dask/array/chunk.py:167-258
                                              (dask 2021.10): np.argpartition / np.partition keep the k largest of a
                                              block, the aggregate sorts them and reverses - TWO passes over the
                                              (M, n_chunk) matrix, one for the indices and one for the scores
  merge                                       indexing/_dictionary_indexing.py:118-128: hstack, argsort(-scores)[:, :k]
                                              (NumPy's default unstable sort), take_along_axis
  the experimental side is LAZY               prepare_experimental starts with `da.asarray(patterns)` (:114), so the
                                              cast / reshape / mask / normalise of ALL experimental patterns is a Dask
                                              graph that is part of every iteration's `da.compute` (:117): it is
                                              re-evaluated for every dictionary chunk
  dask's wrapping of NumPy operands           `da.asarray` / `da.einsum` name the arrays they wrap by a CONTENT HASH
                                              (dask/base.py: tokenize -> normalize_array -> SHA-1 over the buffer when no
                                              faster hash library is installed, as in the reference's environment here):
                                              the raw experimental patterns once, every prepared dictionary chunk

Results equal the oracle's wherever scores are distinct (tests/test_oracle_cpu_port.py).
`run_parallel` spreads the experimental rows over processes so that ALL host cores work when the
BLAS library's own thread pool is smaller than the machine (OpenBLAS caps at 64 threads).
"""

import hashlib
import os
import time

import numpy as np

from oracle import kpdi_oracle as ko


def argtopk_dask(a, k):
    """dask.array.argtopk(k, axis=-1) of one block + aggregate (chunk.py:214-258)."""
    idx = np.argpartition(a, -k, axis=1)[:, -k:]
    vals = np.take_along_axis(a, idx, axis=1)
    order = np.argsort(vals, axis=1)[:, ::-1]
    return np.take_along_axis(idx, order, axis=1)


def topk_dask(a, k):
    """dask.array.topk(k, axis=-1) of one block + aggregate (chunk.py:167-211)."""
    part = np.partition(a, -k, axis=1)[:, -k:]
    return np.sort(part, axis=1)[:, ::-1]


def dictionary_indexing(exp, dic, metric="ncc", keep_n=20, n_per_iteration=None, signal_mask=None):
    """The loop of indexing/_dictionary_indexing.py:66-128 (see the module docstring)."""
    n = dic.shape[0]
    if n_per_iteration is None:
        n_per_iteration = n
    keep_n = min(keep_n, n)
    m_all = max(int(np.prod(exp.shape[:-2])), 1)
    hashlib.sha1(np.ascontiguousarray(exp)).hexdigest()  # da.asarray(patterns): name = content hash
    dic = dic.reshape(n, -1)
    indices = np.zeros((m_all, keep_n), dtype=np.int32)
    scores = np.full((m_all, keep_n), -1, dtype=np.float32)
    for start in range(0, n, n_per_iteration):
        end = min(start + n_per_iteration, n)
        y = ko.prepare_dictionary(dic[start:end], metric, signal_mask, np.float32)
        hashlib.sha1(np.ascontiguousarray(y)).hexdigest()  # da.einsum wraps the prepared chunk: name = content hash
        x = ko.prepare_experimental(exp, metric, m_all, None, signal_mask, np.float32)  # lazy graph, re-evaluated
        sim = ko.match(x, y)
        k = min(keep_n, end - start)
        idx_i = argtopk_dask(sim, k)
        scores_i = topk_dask(sim, k)
        idx_i = idx_i + start
        scores = np.hstack((scores, scores_i))
        indices = np.hstack((indices, idx_i))
        best = np.argsort(-scores, axis=1)[:, :keep_n]
        scores = np.take_along_axis(scores, best, axis=1)
        indices = np.take_along_axis(indices, best, axis=1)
    return scores, indices


def blas_threads():
    try:
        from threadpoolctl import threadpool_info

        t = [p["num_threads"] for p in threadpool_info() if p.get("user_api") == "blas"]
        return max(t) if t else 1
    except Exception:
        return 1


def _worker(args):
    exp, dic, kw = args
    return dictionary_indexing(exp, dic, **kw)


def run_parallel(exp, dic, n_proc, **kw):
    """`n_proc` processes (fork: the dictionary is shared copy-on-write), each with the BLAS
    library's own threads, over contiguous blocks of experimental rows.  Returns
    (scores, indices, seconds)."""
    t0 = time.perf_counter()
    if n_proc <= 1:
        s, i = dictionary_indexing(exp, dic, **kw)
        return s, i, time.perf_counter() - t0
    import multiprocessing as mp

    bounds = np.linspace(0, len(exp), n_proc + 1).astype(int)
    jobs = [(exp[a:b], dic, kw) for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
    with mp.get_context("fork").Pool(len(jobs)) as pool:
        parts = pool.map(_worker, jobs)
    s = np.concatenate([p[0] for p in parts])
    i = np.concatenate([p[1] for p in parts])
    return s, i, time.perf_counter() - t0
