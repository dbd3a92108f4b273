"""Cross-check the C restatement (oracle/kpdi_oracle_c.c) against the NumPy
oracle, which is itself pinned to the reference's golden vectors."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import kpdi_oracle as ko

LIB = os.path.join(ROOT, "oracle", "libkpdi_oracle.so")


@pytest.fixture(scope="module")
def clib():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    lib = C.CDLL(LIB)
    f32p, i64p = C.POINTER(C.c_float), C.POINTER(C.c_int64)
    lib.kpdi_c_normalize.argtypes = [f32p, C.c_int64, C.c_int64, C.c_int]
    lib.kpdi_c_match_topk.argtypes = [f32p, f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int64, f32p, i64p]
    lib.kpdi_c_init_topk.argtypes = [f32p, i64p, C.c_int64]
    return lib


def c_indexing(lib, exp, dic, metric, keep_n, chunk):
    f32p, i64p = C.POINTER(C.c_float), C.POINTER(C.c_int64)
    m, n = exp.shape[0], dic.shape[0]
    code = {"ncc": 0, "ndp": 1}[metric]
    x = np.array(exp.reshape(m, -1), dtype=np.float32, copy=True)
    lib.kpdi_c_normalize(x.ctypes.data_as(f32p), m, x.shape[1], code)
    scores = np.empty((m, keep_n), np.float32)
    idx = np.empty((m, keep_n), np.int64)
    lib.kpdi_c_init_topk(scores.ctypes.data_as(f32p), idx.ctypes.data_as(i64p), scores.size)
    for s in range(0, n, chunk):
        y = np.array(dic[s:s + chunk].reshape(min(chunk, n - s), -1), dtype=np.float32, copy=True)
        lib.kpdi_c_normalize(y.ctypes.data_as(f32p), y.shape[0], y.shape[1], code)
        lib.kpdi_c_match_topk(x.ctypes.data_as(f32p), y.ctypes.data_as(f32p), m, y.shape[0], y.shape[1], keep_n,
                              s, scores.ctypes.data_as(f32p), idx.ctypes.data_as(i64p))
    return scores, idx


@pytest.mark.parametrize("metric,keep_n,chunk", [("ncc", 20, 3000), ("ncc", 5, 700), ("ndp", 20, 1000)])
def test_c_vs_numpy_oracle_and_reference(clib, synth_inputs, metric, keep_n, chunk):
    exp, dic, g = synth_inputs
    s, i = c_indexing(clib, exp, dic, metric, keep_n, chunk)
    rs, ri = ko.dictionary_indexing(exp, dic, metric=metric, keep_n=keep_n, n_per_iteration=chunk)
    # plain sequential float32 accumulation in C: the 1e-5 contract of north_star applies
    ko.assert_topk_parity(s, i, rs, ri, atol=1e-5)
    name = {("ncc", 20): "ncc_k20", ("ncc", 5): "ncc_k5_it700", ("ndp", 20): "ndp_k20"}[(metric, keep_n)]
    ko.assert_topk_parity(s, i, g[f"{name}__scores"], g[f"{name}__indices"], atol=1e-5)
