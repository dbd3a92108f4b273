"""ctypes loader of oracle/libkpdi_oracle.so (kpdi_oracle_c.c) - TEST INFRASTRUCTURE.

Only tests/, `__graft_entry__.smoke()` and bench.py's checker / `cpu_baseline` legs may import
this module; nothing under kikuchipy_amd/ does (tests/test_no_oracle_in_product.py).

`rows_topk_f64` is the full-size checker: a sample of experimental rows against the WHOLE
dictionary, every dot product accumulated in float64 (OpenMP over rows), so hundreds of rows of
BASELINE.json's configs[1..4] are checked in seconds on the GPU box's host cores.
"""

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libkpdi_oracle.so")
_f32p, _i64p = C.POINTER(C.c_float), C.POINTER(C.c_int64)
_lib = None


def effective_cpus():
    """CPUs this process can actually use: visible CPUs, limited by the affinity mask and by the
    cgroup CPU quota (the GPU boxes show 256 logical CPUs under a 16-CPU quota; more runnable
    threads than that only get throttled)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(float(parts[0]) / float(parts[1]) + 0.5)))
            else:
                quota = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.run(["make", "-C", HERE], check=True, capture_output=True)
        lib = C.CDLL(LIB)
        lib.kpdi_c_normalize.argtypes = [_f32p, C.c_int64, C.c_int64, C.c_int]
        lib.kpdi_c_match_topk.argtypes = [_f32p, _f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int64, _f32p, _i64p]
        lib.kpdi_c_match_topk_rows_f64.argtypes = [_f32p, _i64p, C.c_int64, _f32p, C.c_int64, C.c_int64, C.c_int,
                                                   C.c_int64, _f32p, _i64p]
        lib.kpdi_c_prepare_f64.argtypes = [_f32p, C.c_int64, C.c_int64, _i64p, C.c_int64, C.c_int, _f32p]
        lib.kpdi_c_init_topk.argtypes = [_f32p, _i64p, C.c_int64]
        lib.kpdi_c_set_threads.argtypes = [C.c_int]
        lib.kpdi_c_set_threads(effective_cpus())
        lib.kpdi_c_match_topk_fast.argtypes = lib.kpdi_c_match_topk.argtypes
        lib.kpdi_c_prepare_f32.argtypes = lib.kpdi_c_prepare_f64.argtypes
        _lib = lib
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def prepare_f64(raw, metric, signal_mask=None):
    """(n, sy, sx) or (n, npix) raw patterns -> (n, K) float32 rows normalised in float64
    (`ncc`: zero mean, unit norm; `ndp`: unit norm) over the pixels the signal mask keeps."""
    lib = load()
    raw = np.ascontiguousarray(np.asarray(raw).reshape(len(raw), -1), dtype=np.float32)
    pix = None
    k = raw.shape[1]
    if signal_mask is not None:
        pix = np.ascontiguousarray(np.flatnonzero(~np.asarray(signal_mask, dtype=bool).ravel()), dtype=np.int64)
        k = pix.size
    out = np.empty((raw.shape[0], k), dtype=np.float32)
    lib.kpdi_c_prepare_f64(_p(raw, _f32p), raw.shape[0], raw.shape[1], None if pix is None else _p(pix, _i64p), k,
                           {"ncc": 0, "ndp": 1}[metric], _p(out, _f32p))
    return out


def rows_topk_f64(exp, dictionary_chunks, rows, metric, keep_n, signal_mask=None):
    """Best `keep_n` dictionary entries of the experimental patterns `rows`.

    exp: all experimental patterns (raw); dictionary_chunks: iterable of (start, raw chunk)
    covering the dictionary (or one array = one chunk at 0).  Returns (scores, indices) of shape
    (len(rows), keep_n), ties by lower dictionary index first."""
    lib = load()
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    x = prepare_f64(np.asarray(exp)[rows], metric, signal_mask)
    local = np.arange(len(rows), dtype=np.int64)
    scores = np.empty((len(rows), keep_n), np.float32)
    idx = np.empty((len(rows), keep_n), np.int64)
    lib.kpdi_c_init_topk(_p(scores, _f32p), _p(idx, _i64p), scores.size)
    if isinstance(dictionary_chunks, np.ndarray):
        dictionary_chunks = [(0, dictionary_chunks)]
    for start, chunk in dictionary_chunks:
        step = 25000  # bounds the prepared copy
        for s in range(0, len(chunk), step):
            y = prepare_f64(chunk[s:s + step], metric, signal_mask)
            lib.kpdi_c_match_topk_rows_f64(_p(x, _f32p), _p(local, _i64p), len(rows), _p(y, _f32p), y.shape[0],
                                           y.shape[1], keep_n, int(start) + s, _p(scores, _f32p), _p(idx, _i64p))
    return scores, idx


def openmp_port(exp, dic, metric="ncc", keep_n=20, n_per_iteration=None, signal_mask=None):
    """The TIMING variant in C: the reference's chunk loop with every stage on all host cores
    (OpenMP; AVX2 + FMA register-tiled dot products, float32) - bench.py's `cpu_baseline`
    "c_openmp".  Returns (scores, indices)."""
    lib = load()
    n = dic.shape[0]
    if n_per_iteration is None:
        n_per_iteration = n
    keep_n = min(keep_n, n)
    code = {"ncc": 0, "ndp": 1}[metric]
    pix, pp = None, None
    if signal_mask is not None:
        pix = np.ascontiguousarray(np.flatnonzero(~np.asarray(signal_mask, dtype=bool).ravel()), dtype=np.int64)
        pp = _p(pix, _i64p)

    def prep(raw):
        raw = np.ascontiguousarray(np.asarray(raw).reshape(len(raw), -1), dtype=np.float32)
        k = raw.shape[1] if pix is None else pix.size
        out = np.empty((raw.shape[0], k), dtype=np.float32)
        lib.kpdi_c_prepare_f32(_p(raw, _f32p), raw.shape[0], raw.shape[1], pp, k, code, _p(out, _f32p))
        return out

    x = prep(exp.reshape((-1,) + exp.shape[-2:]))
    scores = np.empty((x.shape[0], keep_n), np.float32)
    idx = np.empty((x.shape[0], keep_n), np.int64)
    lib.kpdi_c_init_topk(_p(scores, _f32p), _p(idx, _i64p), scores.size)
    for start in range(0, n, n_per_iteration):
        y = prep(dic[start:start + n_per_iteration])
        lib.kpdi_c_match_topk_fast(_p(x, _f32p), _p(y, _f32p), x.shape[0], y.shape[0], y.shape[1], keep_n, start,
                                   _p(scores, _f32p), _p(idx, _i64p))
    return scores, idx
