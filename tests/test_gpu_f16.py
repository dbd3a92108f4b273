"""The opt-in, REDUCED-PRECISION float16 arithmetic of the match kernel (KPDI_COMPUTE_F16, the
"fp16 MFMA, fp32 accumulate" variant of BASELINE.json configs[4]).

Two statements are tested:
 * what the mode computes is exactly the dot product of the operands rounded to float16: scores
   within 1e-5 of a float64 evaluation over `float16(2^12 * prepared value)` - so masks,
   normalisation, top-k, merge, multi-pass and chunking behave as in the other modes (3e-5 in the
   test: a value on a rounding boundary may round either way);
 * against the reference (float32 operands) the scores stay within 2e-3 (measured: a few 1e-5 at
   K = 3600, the rounding errors of the 2 x K operands average out) - OUTSIDE the 1e-5 contract of
   the default path, which is why the mode is opt-in - and well separated best matches are
   still found.
"""

import numpy as np
import pytest

from oracle import kpdi_oracle as ko

pytestmark = pytest.mark.gpu


def engine(exp, dic, metric="ncc", keep_n=20, chunk=None, signal_mask=None, navigation_mask=None):
    from kikuchipy_amd import _lib

    sy, sx = exp.shape[-2:]
    n = dic.shape[0]
    keep_n = min(keep_n, n)
    with _lib.Context(0) as ctx:
        ctx.set_problem(sy, sx, signal_mask, {"ncc": _lib.METRIC_NCC, "ndp": _lib.METRIC_NDP}[metric], keep_n,
                        _lib.COMPUTE_F16)
        ctx.set_experimental(exp.reshape(-1, sy, sx), navigation_mask)
        chunk = chunk or n
        for start in range(0, n, chunk):
            ctx.push_dictionary_chunk(dic[start:start + chunk], start)
        return ctx.finalize(keep_n)


def rounded_operand_topk(exp, dic, metric, keep_n, signal_mask=None, navigation_mask=None):
    """Top-k of the exact products of the float16-rounded prepared operands.  The operands are
    prepared in float64 here: the reference's float32 sums over a masked uint8 pattern are off by
    ~1e-5 relative (sequential float32 accumulation of a strided array), the engine's tree sums are
    not, and this test is about the products."""
    x = np.asarray(ko.prepare_experimental(exp, metric=metric, signal_mask=signal_mask, dtype=np.float64,
                                           navigation_mask=navigation_mask, n_experimental=len(exp)))
    y = np.asarray(ko.prepare_dictionary(dic.reshape(len(dic), -1), metric=metric, signal_mask=signal_mask,
                                         dtype=np.float64))
    xh = (x * 4096).astype(np.float32).astype(np.float16).astype(np.float64)
    yh = (y * 4096).astype(np.float32).astype(np.float16).astype(np.float64)
    s = (xh @ yh.T) * 2.0**-24
    order = np.lexsort((np.broadcast_to(np.arange(s.shape[1]), s.shape), -s), axis=1)[:, :keep_n]
    return np.take_along_axis(s, order, 1).astype(np.float32), order


@pytest.mark.parametrize("m,n,sy,sx,k,chunk,metric,masked", [
    (1, 1, 8, 8, 1, None, "ncc", False),
    (3, 130, 16, 12, 20, None, "ncc", False),      # ragged tile edges, K = 192 = 3 slabs of 64
    (130, 257, 20, 20, 8, 100, "ndp", True),       # K not a multiple of 64: zero-padded slab
    (260, 1000, 31, 33, 20, 333, "ncc", False),    # K = 1023, odd: scalar preparation path
    (40, 640, 60, 60, 33, 250, "ncc", True),       # keep_n > 32: multi-pass path; staged masked prep
    (257, 900, 6, 37, 64, None, "ncc", False),     # second pass = the bounded 32-entry form (once spilled an accumulator)
    (130, 700, 12, 12, 96, 400, "ndp", False),     # three passes
    (21, 300, 120, 120, 7, 170, "ncc", False),     # workgroup-per-pattern preparation
    (21, 300, 96, 80, 7, None, "ndp", True),
    (9, 200, 130, 130, 5, None, "ncc", False),     # generic preparation kernel
])
def test_computes_the_rounded_operand_products(m, n, sy, sx, k, chunk, metric, masked):
    rng = np.random.default_rng(m * 1000 + n)
    exp = rng.integers(0, 256, (m, sy, sx)).astype(np.uint8)
    dic = rng.random((n, sy, sx)).astype(np.float32)
    signal_mask = None
    if masked:
        yy, xx = np.mgrid[:sy, :sx]
        signal_mask = (yy - sy / 2) ** 2 + (xx - sx / 2) ** 2 > (min(sy, sx) / 2) ** 2
    nav = None
    if m > 20:
        nav = np.zeros(m, dtype=bool)
        nav[[2, m - 1]] = True
    s, i = engine(exp, dic, metric, k, chunk, signal_mask, nav)
    rs, ri = rounded_operand_topk(exp, dic, metric, min(k, n), signal_mask, nav)
    # (a value within float32 noise of a float16 rounding boundary may round the other way: one such
    # flip moves a score by ~2e-6 at K = 316, so the bound is a few of them - one MISSING pixel is 2e-3)
    ko.assert_topk_parity(s, i, rs, ri, atol=3e-5, tie=6e-5)
    # ... and stays within the documented bound of the float32 reference
    fs, fi = ko.dictionary_indexing(exp, dic, metric=metric, keep_n=k, signal_mask=signal_mask, navigation_mask=nav)
    assert np.abs(s - fs).max() < 2e-3


def test_planted_matches_and_bound_at_config2_shape():
    rng = np.random.default_rng(3)
    dic = rng.random((20000, 60, 60), dtype=np.float32)
    exp = rng.integers(0, 256, (512, 60, 60), dtype=np.uint8)
    planted = rng.choice(20000, 64, replace=False)
    for j, p in enumerate(planted):  # a noisy copy of a dictionary pattern: by far the best match
        exp[j] = np.clip(dic[p] * 200 + rng.normal(0, 20, (60, 60)), 0, 255).astype(np.uint8)
    s, i = engine(exp, dic, "ncc", 20, chunk=7000)
    assert np.array_equal(i[:64, 0], planted)
    fs, fi = ko.dictionary_indexing(exp, dic, keep_n=20, n_per_iteration=7000)
    err = np.abs(s - fs)
    assert err.max() < 1e-3 and 1e-7 < err.mean() < 2e-4  # reduced precision, and really a different arithmetic
    # the best 20 of the float32 path are (nearly all) among the best 20 found here
    overlap = np.mean([len(set(a) & set(b)) for a, b in zip(i, fi)]) / 20
    assert overlap > 0.8


def test_python_api_and_chunking_invariance():
    import kikuchipy_amd as ka

    rng = np.random.default_rng(8)
    dic = rng.random((3000, 24, 24), dtype=np.float32)
    exp = rng.integers(0, 256, (6, 7, 24, 24), dtype=np.uint8)
    a = ka.dictionary_indexing(exp, dic, "ncc", 10, compute="f16", verbose=False)
    b = ka.dictionary_indexing(exp, dic, "ncc", 10, n_per_iteration=700, compute="f16", verbose=False)
    assert np.array_equal(a.scores, b.scores) and np.array_equal(a.simulation_indices, b.simulation_indices)
    resident = ka.ResidentDictionary(dic, "ncc", compute="f16")
    c = ka.dictionary_indexing(exp, resident, "ncc", 10, verbose=False)
    assert np.array_equal(a.scores, c.scores) and np.array_equal(a.simulation_indices, c.simulation_indices)
    f = ka.dictionary_indexing(exp, dic, "ncc", 10, verbose=False)
    assert 0 < np.abs(a.scores - f.scores).max() < 2e-3
    with pytest.raises(ValueError, match="compute must be one of"):
        ka.dictionary_indexing(exp, dic, "ncc", 10, compute="bf16", verbose=False)


@pytest.mark.parametrize("sy,sx,n", [(120, 120, 700), (60, 60, 1500), (60, 60, 1501)])
def test_float16_dictionaries_skip_the_cast(sy, sx, n):
    """A dictionary HANDED OVER as float16 (KPDI_F16: half the bytes in host memory, over PCIe and in HBM) is cast exactly
    by the preparation kernels: the results of the same values handed over as float32 (bit for bit where the same
    kernel serves both), in the float16 arithmetic and in the default float32 one."""
    from kikuchipy_amd import _lib

    rng = np.random.default_rng(sy + n)
    exp = rng.integers(0, 256, (70, sy, sx), dtype=np.uint8)
    dic16 = rng.random((n, sy, sx), dtype=np.float32).astype(np.float16)
    yy, xx = np.mgrid[:sy, :sx]
    mask = (yy - sy / 2) ** 2 + (xx - sx / 2) ** 2 > (sy / 2) ** 2
    for compute in (_lib.COMPUTE_F16, _lib.COMPUTE_F32):
        for sm in (None, mask):
            got = []
            for dic in (dic16, dic16.astype(np.float32)):
                with _lib.Context(0) as ctx:
                    ctx.set_problem(sy, sx, sm, _lib.METRIC_NCC, 10, compute)
                    ctx.set_experimental(exp)
                    ctx.push_dictionary_chunk(dic, 0)
                    got.append(ctx.finalize(10))
            if sy <= 64:  # one wave per pattern: the same kernel, the same order of every sum
                assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
            else:
                # 120 x 120 unmasked float16 rows are read 8 pixels per load (16 bytes, as float32 rows are): the mean
                # and the norm are summed in another order than for the float32 copy - rounding-level differences
                # (which can flip a float16 rounding of a prepared value: 3e-5 in the float16 arithmetic)
                ko.assert_topk_parity(got[0][0], got[0][1], got[1][0], got[1][1], atol=3e-5 if compute == _lib.COMPUTE_F16 else 1e-6,
                                      tie=6e-5 if compute == _lib.COMPUTE_F16 else 2e-6)
