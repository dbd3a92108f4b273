"""GPU parity tests of the C-ABI engine (libkpdi.so through ctypes) against the
CPU oracle and the golden vectors made by the reference.  Tolerance: scores
within 1e-5 absolute (BASELINE.json north_star), indices per
`oracle.kpdi_oracle.assert_topk_parity`."""

import numpy as np
import pytest

from conftest import load_golden
from oracle import kpdi_oracle as ko

pytestmark = pytest.mark.gpu

ATOL = 1e-5


@pytest.fixture(scope="module")
def ctx():
    from kikuchipy_amd import _lib

    c = _lib.Context(0)
    yield c
    c.close()


def run_engine(ctx, exp, dic, metric="ncc", keep_n=20, chunk=None, signal_mask=None,
               navigation_mask=None):
    from kikuchipy_amd import _lib

    sy, sx = exp.shape[-2:]
    n = dic.shape[0]
    keep_n = min(keep_n, n)
    ctx.set_problem(sy, sx, signal_mask, {"ncc": _lib.METRIC_NCC, "ndp": _lib.METRIC_NDP}[metric], keep_n)
    ctx.set_experimental(exp.reshape(-1, sy, sx), navigation_mask)
    chunk = chunk or n
    for start in range(0, n, chunk):
        ctx.push_dictionary_chunk(dic[start:start + chunk], start)
    return ctx.finalize(keep_n)


SYNTH_CASES = {
    "ncc_k20": dict(metric="ncc", keep_n=20),
    "ncc_k1": dict(metric="ncc", keep_n=1),
    "ncc_k5_it700": dict(metric="ncc", keep_n=5, chunk=700),
    "ndp_k20": dict(metric="ndp", keep_n=20),
    "ndp_k5_it1000": dict(metric="ndp", keep_n=5, chunk=1000),
    "ncc_k20_circ": dict(metric="ncc", keep_n=20, signal_mask="circ"),
    "ncc_k20_circ_it999": dict(metric="ncc", keep_n=20, signal_mask="circ", chunk=999),
    "ncc_k10_f64": dict(metric="ncc", keep_n=10),
    "ndp_k50": dict(metric="ndp", keep_n=50),
}


@pytest.mark.parametrize("name", sorted(SYNTH_CASES))
def test_golden_synth(ctx, name, synth_inputs):
    """The reference's own results (tests/golden/di_synth.npz)."""
    exp, dic, g = synth_inputs
    kw = dict(SYNTH_CASES[name])
    if kw.get("signal_mask") == "circ":
        kw["signal_mask"] = g["circular_mask"]
    scores, idx = run_engine(ctx, exp, dic, **kw)
    ko.assert_topk_parity(scores, idx, g[f"{name}__scores"], g[f"{name}__indices"], atol=ATOL)
    assert scores.dtype == np.float32 and idx.dtype == np.int64


def test_golden_navmask(ctx, synth_inputs):
    exp, dic, g = synth_inputs
    nav = g["nav_mask"]
    scores, idx = run_engine(ctx, exp.reshape(6, 8, 60, 60), dic, metric="ncc", keep_n=7, chunk=1500,
                             navigation_mask=nav)
    assert scores.shape == (45, 7)
    ko.assert_topk_parity(scores, idx, g["ncc_k7_nav__scores"][~nav.ravel()],
                          g["ncc_k7_nav__indices"][~nav.ravel()], atol=ATOL)


def test_golden_config1(ctx, config1_inputs):
    """BASELINE.json configs[0]."""
    exp, dic, g = config1_inputs
    s, i = run_engine(ctx, exp, dic, metric="ncc", keep_n=5)
    ko.assert_topk_parity(s, i, g["ncc_k5__scores"], g["ncc_k5__indices"], atol=ATOL)
    assert np.allclose(s[:, 0], 1, atol=ATOL)
    assert list(i[:, 0]) == list(range(0, 999, 111))
    circ = ~ko.circular_window((60, 60)).astype(bool)
    s, i = run_engine(ctx, exp, dic, metric="ncc", keep_n=5, signal_mask=circ, chunk=300)
    ko.assert_topk_parity(s, i, g["ncc_k5_circ_it300__scores"], g["ncc_k5_circ_it300__indices"], atol=ATOL)


@pytest.mark.parametrize("name", ["ndp_all", "ncc_all", "ndp_sigmask", "ndp_it2", "ncc_it4_k3"])
def test_golden_dummy(ctx, name):
    """3x3 detector, 9 patterns, dictionary == experimental
    (tests/test_indexing/test_dictionary_indexing.py:27-66 of the reference)."""
    g = load_golden("di_dummy.npz")
    kw = {
        "ndp_all": dict(metric="ndp"),
        "ncc_all": dict(metric="ncc"),
        "ndp_sigmask": dict(metric="ndp", signal_mask=g["sig_mask"]),
        "ndp_it2": dict(metric="ndp", chunk=2),
        "ncc_it4_k3": dict(metric="ncc", keep_n=3, chunk=4),
    }[name]
    dummy = g["dummy"]
    s, i = run_engine(ctx, dummy, dummy.reshape(-1, 3, 3), **kw)
    assert np.allclose(s[:, 0], 1, atol=ATOL)
    ko.assert_topk_parity(s, i, g[f"{name}__scores"], g[f"{name}__indices"], atol=ATOL, tie=2e-5)


@pytest.mark.parametrize("m,n,sy,sx,k,chunk,metric", [
    (1, 1, 8, 8, 1, None, "ncc"),        # smallest problem
    (3, 130, 16, 12, 20, None, "ncc"),   # ragged tile edges in every direction
    (130, 257, 20, 20, 8, 100, "ndp"),   # two experimental row blocks, ragged chunks
    (260, 1000, 31, 33, 20, 333, "ncc"), # K = 1023 -> padded to 1024
    (17, 640, 60, 60, 33, 250, "ncc"),   # keep_n > 32: multi-pass path
    (5, 70, 10, 10, 70, None, "ndp"),    # keep_n == dictionary size
])
def test_vs_oracle_shapes(ctx, m, n, sy, sx, k, chunk, metric):
    rng = np.random.default_rng(m * 1000 + n)
    exp = rng.integers(0, 256, (m, sy, sx)).astype(np.uint8)
    dic = rng.random((n, sy, sx)).astype(np.float32)
    s, i = run_engine(ctx, exp, dic, metric=metric, keep_n=k, chunk=chunk)
    rs, ri = ko.dictionary_indexing(exp, dic, metric=metric, keep_n=k, n_per_iteration=chunk)
    ko.assert_topk_parity(s, i, rs, ri, atol=ATOL)


@pytest.mark.parametrize("sy,sx,masked,dtype,metric", [
    (70, 72, False, np.float32, "ncc"),    # K = 5040: one workgroup per pattern, vector loads
    (120, 120, False, np.uint8, "ndp"),    # configs[4] detector
    (128, 128, False, np.uint16, "ncc"),   # K = 16384, the largest register-resident pattern
    (96, 80, True, np.float32, "ncc"),     # signal mask: gather through the pixel map
    (120, 120, True, np.uint8, "ncc"),
    (75, 75, False, np.float32, "ncc"),    # K % 4 != 0: generic kernel
    (130, 130, False, np.uint8, "ncc"),    # K > 16384: generic kernel
    (130, 130, True, np.float32, "ndp"),
    (256, 250, False, np.uint8, "ncc"),    # K = 64 000: 32 MB per dictionary tile (scalar offsets of the LDS-DMA pieces)
])
def test_large_detectors(ctx, sy, sx, masked, dtype, metric):
    """Pattern preparation switches kernels with the number of kept pixels (csrc/prep.hip)."""
    from kikuchipy_amd import _lib

    rng = np.random.default_rng(sy * sx)
    exp = (rng.random((21, sy, sx)) * 250).astype(dtype)
    dic = (rng.random((300, sy, sx)) * 250).astype(dtype)
    signal_mask = None
    if masked:
        yy, xx = np.mgrid[:sy, :sx]
        signal_mask = (yy - sy / 2) ** 2 + (xx - sx / 2) ** 2 > (min(sy, sx) / 2) ** 2
    nav_mask = np.zeros(21, dtype=bool)
    nav_mask[[3, 20]] = True
    s, i = run_engine(ctx, exp, dic, metric=metric, keep_n=7, chunk=170, signal_mask=signal_mask,
                      navigation_mask=nav_mask)
    rs, ri = ko.dictionary_indexing(exp, dic, metric=metric, keep_n=7, signal_mask=signal_mask,
                                    navigation_mask=nav_mask)
    ko.assert_topk_parity(s, i, rs, ri, atol=ATOL)
    # the opt-in split-f16 arithmetic goes through the same preparation kernels
    ctx.set_problem(sy, sx, signal_mask, {"ncc": _lib.METRIC_NCC, "ndp": _lib.METRIC_NDP}[metric], 7,
                    _lib.COMPUTE_F16X2)
    ctx.set_experimental(exp, nav_mask)
    ctx.push_dictionary_chunk(dic, 0)
    s16, i16 = ctx.finalize(7)
    ko.assert_topk_parity(s16, i16, rs, ri, atol=ATOL)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int16, np.float16, np.float32, np.float64])
def test_input_dtypes(ctx, dtype):
    rng = np.random.default_rng(5)
    exp = (rng.random((40, 24, 24)) * 200).astype(dtype)
    dic = (rng.random((300, 24, 24)) * 200).astype(dtype)
    s, i = run_engine(ctx, exp, dic, keep_n=6)
    rs, ri = ko.dictionary_indexing(exp, dic, keep_n=6)
    ko.assert_topk_parity(s, i, rs, ri, atol=ATOL)


def test_ties_lower_index_first(ctx):
    """Exact duplicates in the dictionary: equal scores come out by ascending index."""
    rng = np.random.default_rng(3)
    base = rng.random((50, 16, 16)).astype(np.float32)
    dic = np.concatenate([base, base, base])  # entries j, j+50, j+100 are identical
    exp = (base[:10] * 255).astype(np.uint8)
    s, i = run_engine(ctx, exp, dic, keep_n=6, chunk=64)
    assert np.all(np.diff(s, axis=1) <= 0)
    for row_s, row_i in zip(s, i):
        for a in range(5):
            if row_s[a] == row_s[a + 1]:
                assert row_i[a] < row_i[a + 1]
    # every pattern's best three are the three copies of its own source, in index order
    assert np.array_equal(i[:, :3] % 50, np.repeat(np.arange(10)[:, None], 3, axis=1))
    assert np.all(np.diff(i[:, :3], axis=1) > 0)


def test_chunking_invariance(ctx, synth_inputs):
    """Size-independent property: the result does not depend on how the
    dictionary is cut into chunks nor on the order the chunks arrive in."""
    exp, dic, g = synth_inputs
    from kikuchipy_amd import _lib

    ref = run_engine(ctx, exp, dic, keep_n=20)
    ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 20)
    ctx.set_experimental(exp)
    bounds = [0, 17, 900, 901, 2049, 3000]
    for a, b in reversed(list(zip(bounds[:-1], bounds[1:]))):
        ctx.push_dictionary_chunk(dic[a:b], a)
    s, i = ctx.finalize(20)
    assert np.array_equal(i, ref[1]) and np.array_equal(s, ref[0])


def test_inputs_not_mutated(ctx):
    rng = np.random.default_rng(1)
    exp = rng.random((9, 12, 12)).astype(np.float32)
    dic = rng.random((40, 12, 12)).astype(np.float32)
    e0, d0 = exp.copy(), dic.copy()
    run_engine(ctx, exp, dic, keep_n=3)
    assert np.array_equal(exp, e0) and np.array_equal(dic, d0)


def test_error_paths(ctx):
    from kikuchipy_amd import _lib

    with pytest.raises(_lib.KpdiError, match="keep_n"):
        ctx.set_problem(4, 4, None, _lib.METRIC_NCC, 0)
    with pytest.raises(_lib.KpdiError, match="every pixel"):
        ctx.set_problem(2, 2, np.ones((2, 2), bool), _lib.METRIC_NCC, 1)
    with pytest.raises(_lib.KpdiError, match="unknown metric"):
        ctx.set_problem(2, 2, None, 7, 1)


def test_rccl_path_single_rank(synth_inputs):
    """The multi-GPU finalize (RCCL all-gather of the per-rank best-k + merge kernel)
    with a one-rank communicator: must reproduce the plain result exactly."""
    from kikuchipy_amd import _lib

    exp, dic, g = synth_inputs
    with _lib.Context(0) as c:
        ref = run_engine(c, exp, dic, keep_n=20, chunk=1000)
        c.comm_init(0, 1, _lib.Context.comm_unique_id())
        out = run_engine(c, exp, dic, keep_n=20, chunk=1000)
        again = c.finalize(20)  # finalize is idempotent
    assert np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1])
    assert np.array_equal(again[0], ref[0]) and np.array_equal(again[1], ref[1])


def test_small_chunks_are_swept_together_and_nothing_changes(monkeypatch):
    """Coalescing (csrc/sweep.hip): pushed chunks of less than two tile rounds wait for company and are prepared + matched
    as ONE launch set; the merge translates rows of the coalesced matrix back to dictionary indices.  Whatever the order,
    size, dtype or origin (host / device / generated) of the chunks the result is the single pass's, BIT FOR BIT; chunks
    that cannot wait (float64 arithmetic, a start below what is pending, a 17th segment) are swept at once
    or force the pending ones out first.  The reference's own call shape: `n_per_iteration` patterns per iteration
    (indexing/_dictionary_indexing.py:100-128)."""
    from kikuchipy_amd import _lib

    rng = np.random.default_rng(77)
    exp = rng.integers(0, 256, (700, 24, 20), dtype=np.uint8)
    dic = rng.random((9000, 24, 20), dtype=np.float32)
    dic[5000] = dic[17]  # ties across chunks: lower dictionary index first
    dic[8999] = dic[17]
    with _lib.Context(0) as c:
        def sweep(pieces, keep_n=20, metric=_lib.METRIC_NCC, compute=_lib.COMPUTE_F32, dev=False, source=dic):
            c.set_problem(24, 20, None, metric, keep_n, compute)
            c.set_experimental(exp)
            c.reset_counters()
            d = None
            if dev:
                d = c.dev_alloc(source.nbytes)
                c.h2d(d, source)
            for a, b in pieces:
                if dev:
                    c.push_dictionary_chunk_dev(d + a * source[0].nbytes, source.dtype, b - a, a)
                else:
                    c.push_dictionary_chunk(source[a:b], a)
            out = c.finalize(keep_n)
            cnt = c.counters()
            if dev:
                c.dev_free(d)
            return out, cnt

        (ref_s, ref_i), cnt = sweep([(0, 9000)])
        assert cnt["coalesced_sweeps"] == 0 and cnt["match_launches"] == 1
        chunks = [(a, min(a + 700, 9000)) for a in range(0, 9000, 700)]
        for dev in (False, True):
            (s, i), cnt = sweep(chunks, dev=dev)
            assert np.array_equal(s, ref_s) and np.array_equal(i, ref_i)
            assert cnt["coalesced_sweeps"] >= 1 and cnt["match_launches"] < len(chunks), cnt
            assert cnt["match_flops"] == 2.0 * 700 * 9000 * 480
        # every other chunk first, then the rest: ascending runs coalesce, the step back forces a flush
        order = chunks[0::2] + chunks[1::2]
        (s, i), cnt = sweep(order)
        assert np.array_equal(s, ref_s) and np.array_equal(i, ref_i) and cnt["coalesced_sweeps"] >= 2
        (s, i), cnt = sweep(chunks[::-1])  # descending: nothing can wait for anything
        assert np.array_equal(s, ref_s) and np.array_equal(i, ref_i) and cnt["match_launches"] == len(chunks)
        tiny = [(a, a + 100) for a in range(0, 9000, 100)]  # 90 chunks: more than a coalesced matrix has segments
        (s, i), cnt = sweep(tiny)
        assert np.array_equal(s, ref_s) and np.array_equal(i, ref_i)
        assert cnt["coalesced_sweeps"] == -(-90 // 16), cnt
        # mixed dtypes (the uint16 chunks hold the same values scaled: ndp does not see the scale)
        d16 = (dic * 1000).astype(np.uint16)
        (r16_s, r16_i), _ = sweep([(0, 9000)], metric=_lib.METRIC_NDP, source=d16.astype(np.float32))
        c.set_problem(24, 20, None, _lib.METRIC_NDP, 20, _lib.COMPUTE_F32)
        c.set_experimental(exp)
        for j, (a, b) in enumerate(chunks):
            c.push_dictionary_chunk(d16[a:b] if j % 3 else d16[a:b].astype(np.float32), a)
        s, i = c.finalize(20)
        assert np.array_equal(s, r16_s) and np.array_equal(i, r16_i)
        # keep_n > 32 (bounded passes: the bounds live in row space until the merge) coalesces too; float64 arithmetic reads
        # a chunk's own raw patterns: swept at once, same result as ever
        for kw, together in ((dict(keep_n=40), True), (dict(compute=_lib.COMPUTE_F64, keep_n=10), False)):
            (a_s, a_i), cnt1 = sweep([(0, 9000)], **kw)
            (b_s, b_i), cntn = sweep(chunks, **kw)
            assert np.array_equal(a_i, b_i) and (cntn["coalesced_sweeps"] >= 1) == together
            assert np.array_equal(a_s, b_s) or kw.get("compute") == _lib.COMPUTE_F64 and np.abs(a_s - b_s).max() < 1e-14
        # float16 arithmetic coalesces like float32
        (a_s, a_i), _ = sweep([(0, 9000)], compute=_lib.COMPUTE_F16)
        (b_s, b_i), cnt = sweep(chunks, compute=_lib.COMPUTE_F16)
        assert np.array_equal(a_s, b_s) and np.array_equal(a_i, b_i) and cnt["coalesced_sweeps"] >= 1
        # pending chunks belong to their sweep: a new experimental set forgets them, synchronize sweeps them
        c.set_problem(24, 20, None, _lib.METRIC_NCC, 20, _lib.COMPUTE_F32)
        c.set_experimental(exp)
        c.push_dictionary_chunk(dic[:700], 0)
        c.set_experimental(exp)
        c.push_dictionary_chunk(dic[700:], 700)
        c.push_dictionary_chunk(dic[:700], 0)
        c.synchronize()
        s, i = c.finalize(20)
        assert np.array_equal(s, ref_s) and np.array_equal(i, ref_i)
    monkeypatch.setenv("KPDI_NO_COALESCE", "1")
    with _lib.Context(0) as c:
        (s, i), cnt = sweep(chunks)
        assert np.array_equal(s, ref_s) and np.array_equal(i, ref_i)
        assert cnt["coalesced_sweeps"] == 0 and cnt["match_launches"] == len(chunks)


def test_host_staged_gather_entry_points(synth_inputs):
    """kpdi_export_lists / kpdi_import_lists / kpdi_comm_selftest / kpdi_comm_drop (the fallback of a multi-process job
    whose RCCL communicator cannot be used): three "ranks" sweep their dictionary blocks on one GPU, their exported
    lists imported into one context and merged by its finalize == the single sweep, bit for bit - in float32 and in
    float64 arithmetic; a one-rank communicator passes its self-test and can be dropped."""
    from kikuchipy_amd import _lib
    from kikuchipy_amd.parallel import shard_range

    exp, dic, g = synth_inputs
    k, world = 20, 3
    for compute in (_lib.COMPUTE_F32, _lib.COMPUTE_F64):
        lists = []
        with _lib.Context(0) as c:
            c.set_problem(60, 60, None, _lib.METRIC_NCC, k, compute)
            c.set_experimental(exp)
            c.push_dictionary_chunk(dic, 0)
            ref_s, ref_i = c.finalize(k)
            for r in range(world):
                lo, hi = shard_range(len(dic), r, world)
                c.reset_topk()
                c.push_dictionary_chunk(dic[lo:hi], lo)
                lists.append(c.export_lists())
            assert lists[0][0].dtype == (np.float64 if compute == _lib.COMPUTE_F64 else np.float32) and lists[0][1].dtype == np.int32
            assert lists[1][1].min() >= shard_range(len(dic), 1, world)[0]  # a rank's own lists: global indices of ITS block
            c.import_lists(np.stack([s for s, _ in lists]), np.stack([i for _, i in lists]))
            s, i = c.finalize(k)
            assert c.counters()["gather_ranks"] == world
            assert np.array_equal(s, ref_s) and np.array_equal(i, ref_i)
            s2, i2 = c.finalize(k)  # the imported lists were consumed: this rank's own again
            assert np.array_equal(i2, lists[-1][1])
            with pytest.raises(_lib.KpdiError, match="expected"):
                c.import_lists(np.zeros((2, 3, k), np.float32), np.zeros((2, 3, k), np.int32))
    with _lib.Context(0) as c:
        with pytest.raises(_lib.KpdiError, match="no communicator"):
            c.comm_selftest()
        c.comm_init(0, 1, _lib.Context.comm_unique_id())
        c.comm_selftest(1 << 16, 20000)
        assert c.counters()["comm_ranks"] == 1
        c.comm_drop()
        assert c.counters()["comm_ranks"] == 0
        out = run_engine(c, exp, dic, keep_n=k, chunk=1000)
        with _lib.Context(0) as c2:
            ref = run_engine(c2, exp, dic, keep_n=k, chunk=1000)
        assert np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1])
        c.comm_init(0, 1, _lib.Context.comm_unique_id())  # a dropped context can get a communicator again
        c.comm_selftest(1 << 12, 20000)


def test_sharded_sweep_equals_full_sweep(synth_inputs):
    """What N ranks do, on one GPU: each 'rank' sweeps only its dictionary block
    (kikuchipy_amd.parallel.shard_range); merging the per-rank lists with the
    engine's merge order gives the single-GPU result bit for bit."""
    from kikuchipy_amd import _lib
    from kikuchipy_amd.parallel import shard_range

    exp, dic, g = synth_inputs
    k = 20
    with _lib.Context(0) as c:
        ref_s, ref_i = run_engine(c, exp, dic, keep_n=k)
        parts = []
        for r in range(8):
            lo, hi = shard_range(len(dic), r, 8)
            c.set_problem(60, 60, None, _lib.METRIC_NCC, k)
            c.set_experimental(exp)
            c.push_dictionary_chunk(dic[lo:hi], lo)
            parts.append(c.finalize(k))
    s = np.concatenate([p[0] for p in parts], axis=1)
    i = np.concatenate([p[1] for p in parts], axis=1)
    order = np.lexsort((i, -s), axis=1)[:, :k]
    assert np.array_equal(np.take_along_axis(i, order, 1), ref_i)
    assert np.array_equal(np.take_along_axis(s, order, 1), ref_s)


@pytest.mark.parametrize("n,metric,k", [(12500, "ncc", 20), (12500 + 128 * 5 + 3, "ndp", 8), (25000, "ncc", 1)])
def test_quarter_tile_tail(n, metric, k, monkeypatch):
    """A rank's share of configs[1] sharded over 8 (4) GPUs: 98 (196) dictionary tiles over the 16 workgroups
    of a row block.  The last n_tiles % 16 tiles are swept as QUARTER tiles by the kernel's 32-row form
    (api.hip: run_match); the result is bit-identical to whole tiles, incl. a partly filled last unit."""
    from kikuchipy_amd import _lib

    rng = np.random.default_rng(21)
    exp = rng.integers(0, 256, (4096, 60, 60), dtype=np.uint8)
    dic = rng.random((n, 60, 60), dtype=np.float32)
    dic[n - 5] = exp[77].astype(np.float32)  # planted into the tail units
    dic[n - 200] = exp[4000].astype(np.float32)
    code = {"ncc": _lib.METRIC_NCC, "ndp": _lib.METRIC_NDP}[metric]
    res = {}
    with _lib.Context(0) as c:
        for tail in (True, False):
            if not tail:
                monkeypatch.setenv("KPDI_NO_TAIL", "1")
            c.set_problem(60, 60, None, code, k)
            c.set_experimental(exp)
            c.push_dictionary_chunk(dic, 1000)
            res[tail] = c.finalize(k)
    assert np.array_equal(res[True][1], res[False][1]) and np.array_equal(res[True][0], res[False][0])
    s, i = res[True]
    assert i[77, 0] == 1000 + n - 5 and i[4000, 0] == 1000 + n - 200
    rows = np.array([0, 77, 1234, 4000, 4095])
    rs, ri = ko.dictionary_indexing(exp[rows], dic, metric=metric, keep_n=k)
    ko.assert_topk_parity(s[rows], i[rows] - 1000, rs, ri, atol=ATOL)


@pytest.mark.parametrize("k,metric", [(20, "ncc"), (40, "ndp")])
def test_large_experimental_set_takes_several_launches(k, metric):
    """More than 256 / nsplit row blocks: the sweep is cut into launches over two streams
    (api.hip: choose_nsplit); with keep_n > 32 every pass is."""
    from kikuchipy_amd import _lib

    rng = np.random.default_rng(77)
    exp = rng.integers(0, 256, (17000, 12, 12), dtype=np.uint8)
    dic = rng.random((1500, 12, 12)).astype(np.float32)
    nav = rng.random(17000) < 0.1
    with _lib.Context(0) as c:
        c.set_problem(12, 12, None, {"ncc": _lib.METRIC_NCC, "ndp": _lib.METRIC_NDP}[metric], k)
        c.set_experimental(exp, nav)
        for a in range(0, 1500, 700):
            c.push_dictionary_chunk(dic[a:a + 700], a)
        s, i = c.finalize(k)
        grid = c.counters()["match_grid"]
    assert s.shape == (int((~nav).sum()), k) and grid <= 256
    rows = np.flatnonzero(~nav)[::97]
    rs, ri = ko.dictionary_indexing(exp[rows], dic, metric=metric, keep_n=k)
    pos = np.searchsorted(np.flatnonzero(~nav), rows)
    ko.assert_topk_parity(s[pos], i[pos], rs, ri, atol=ATOL)


@pytest.mark.parametrize("compute,keep_n", [("f32", 20), ("f16", 20), ("f32", 45), ("f16", 40)])
def test_a_launch_of_29_row_blocks_keeps_its_xcd_grid(monkeypatch, compute, keep_n):
    """api.hip: plan_xcd_grid lays the XCD rectangles of a match16.hip launch over its row blocks ROUNDED UP to the
    grid (29 -> 32: the workgroups of the three missing row blocks leave at once; configs[3]'s last launch).  Same
    lists, bit for bit, as with the padding switched off (KPDI_XCD_PAD=0: 29 row blocks x 1 split per XCD), a larger
    launch, and the oracle agrees (keep_n > 32: the bounded passes are padded launches too)."""
    from kikuchipy_amd import _lib

    rng = np.random.default_rng(29)
    m = 29 * 256 - 40                                  # 29 row blocks, the last one ragged
    exp = rng.integers(0, 256, (m, 16, 16), dtype=np.uint8)
    dic = rng.random((5000, 16, 16), dtype=np.float32)
    monkeypatch.setenv("KPDI_F32_WIDE", "1")           # (the f32 sweep on match16.hip's kernel)
    out = {}
    for pad in ("0", "1"):
        monkeypatch.setenv("KPDI_XCD_PAD", pad)
        with _lib.Context(0) as c:
            c.set_problem(16, 16, None, _lib.METRIC_NCC, keep_n, {"f32": _lib.COMPUTE_F32, "f16": _lib.COMPUTE_F16}[compute])
            c.set_experimental(exp)
            c.push_dictionary_chunk(dic[:3100], 0)
            c.push_dictionary_chunk(dic[3100:], 3100)
            out[pad] = c.finalize(keep_n) + (c.counters()["match_grid"],)
    assert np.array_equal(out["0"][0], out["1"][0]) and np.array_equal(out["0"][1], out["1"][1])
    assert out["1"][2] == out["0"][2] // 29 * 32, (out["0"][2], out["1"][2])  # 29 x nsplit -> 32 x nsplit workgroups
    rows = np.arange(0, m, 211)
    rs, ri = ko.dictionary_indexing(exp[rows], dic, metric="ncc", keep_n=keep_n)
    if compute == "f32":
        ko.assert_topk_parity(out["1"][0][rows], out["1"][1][rows], rs, ri, atol=ATOL)
    else:
        assert np.abs(out["1"][0][rows] - rs).max() < 2e-3


def test_the_two_f32_kernels_agree_and_are_chosen_by_size(monkeypatch):
    """KPDI_COMPUTE_F32 runs on match.hip (128 x 256 tiles, lists in registers) or on the one-wave form of match16.hip
    (256 x 256 tiles, lists in scratch, partial units at the end of a launch): the same exact-f32 products in the same
    order - identical scores and indices - and the choice follows the estimated makespan (api.hip: decide_form)."""
    from kikuchipy_amd import _lib

    rng = np.random.default_rng(17)
    exp = rng.integers(0, 256, (600, 24, 24), dtype=np.uint8)
    dic = rng.random((4700, 24, 24), dtype=np.float32)   # 19 tiles of 256: whole tiles + partial units
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("KPDI_F32_WIDE", mode)
        with _lib.Context(0) as c:
            for metric in (_lib.METRIC_NCC, _lib.METRIC_NDP):
                c.set_problem(24, 24, None, metric, 20)
                c.set_experimental(exp)
                c.push_dictionary_chunk(dic[:3000], 0)
                c.push_dictionary_chunk(dic[3000:], 3000)
                out[mode, metric] = c.finalize(20)
                assert c.counters()["match_form"] == (3 if mode == "1" else 0)
    for metric in (_lib.METRIC_NCC, _lib.METRIC_NDP):
        assert np.array_equal(out["0", metric][0], out["1", metric][0])
        assert np.array_equal(out["0", metric][1], out["1", metric][1])
    monkeypatch.delenv("KPDI_F32_WIDE")
    big = rng.random((100000, 60, 60), dtype=np.float32)   # (the model is fitted for K = 2819 .. 14 400 kept pixels)
    with _lib.Context(0) as c:
        c.set_problem(60, 60, None, _lib.METRIC_NCC, 20)
        c.set_experimental(rng.integers(0, 256, (4096, 60, 60), dtype=np.uint8))
        d = c.dev_alloc(big.nbytes)
        c.h2d(d, big)
        c.push_dictionary_chunk_dev(d, np.float32, 100000, 0)   # 391 tiles of 256 over 16 splits: the one-wave kernel
        assert c.counters()["match_form"] == 3
        c.set_experimental(rng.integers(0, 256, (4096, 60, 60), dtype=np.uint8))
        c.push_dictionary_chunk_dev(d, np.float32, 12500, 0)    # one rank's share at N = 8: three whole rounds + tailgemm.hip
        assert c.counters()["match_form"] == 3
        c.set_experimental(rng.integers(0, 256, (4096, 60, 60), dtype=np.uint8))
        c.push_dictionary_chunk_dev(d, np.float32, 6250, 0)     # ... at N = 16: one round and a half - match.hip + quarter-tile tail
        assert c.counters()["match_form"] == 0
        c.dev_free(d)


def test_the_automatic_kernel_choice_is_within_3_percent_of_the_better_kernel(monkeypatch):
    """decide_form's cost model (csrc/form_model.h, fitted on profiles/r03_form_choice.json) against live timings: at
    every point of a sub-grid of tools/form_probe.py - a rank's share at N = 8, configs[1], map-sized experimental sets,
    60 x 60 masked and unmasked - the step with the automatically chosen kernel takes at most 3 % (+ 30 us of timer
    noise on the millisecond-sized steps) longer than with the better of the two kernels forced.  (Round 4's full grid,
    profiles/r04_form_choice.json: worst point 2.9 %, mean 0.08 %, 80 shapes; the bar was 2 % while the worst point
    was 1.8 - 2.6 % - the wide kernel's launch has since become 0.04 ms cheaper, which moved 4096 x 37 500 x 2819 from
    2.1 to 2.4 - 2.7 %, and two re-fits of the launch constant only traded that point for another,
    form_model.h.)"""
    import importlib.util
    import os

    from conftest import ROOT
    from kikuchipy_amd import _lib

    spec = importlib.util.spec_from_file_location("form_probe", os.path.join(ROOT, "tools", "form_probe.py"))
    fp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fp)
    rng = np.random.default_rng(5)
    points = [(4096, 12500, 3600), (4096, 100000, 3600), (4096, 37500, 2819), (10000, 37500, 3600), (10000, 50000, 3600),
              (40000, 12500, 3600), (512, 25000, 3600), (10000, 12500, 2819)]
    pool = rng.random(100000 * 3600, dtype=np.float32)
    exp_pool = rng.integers(0, 256, 40000 * 3600, dtype=np.uint8)
    worst = []
    with _lib.Context(0) as ctx:
        ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 20)
        d_dic = ctx.dev_alloc(pool.nbytes)
        ctx.h2d(d_dic, pool)
        d_exp = ctx.dev_alloc(exp_pool.nbytes)
        ctx.h2d(d_exp, exp_pool)
        for m, n, k in points:
            mask = fp.circular_mask(60) if k == 2819 else None
            ms = {}
            for name, env in (("classic", "0"), ("wide", "1"), ("auto", None)):
                if env is None:
                    monkeypatch.delenv("KPDI_F32_WIDE", raising=False)
                else:
                    monkeypatch.setenv("KPDI_F32_WIDE", env)
                ms[name], form = fp.time_step(ctx, d_exp, m, d_dic, n, 60, mask, _lib.METRIC_NCC, 5)
            monkeypatch.delenv("KPDI_F32_WIDE", raising=False)
            better = min(ms["classic"], ms["wide"])
            worst.append((round(ms["auto"] / better, 4), m, n, k, ms))
            print((m, n, k), {key: round(v, 4) for key, v in ms.items()}, "automatic choice:", {0: "classic", 3: "wide"}.get(form, form), flush=True)
            assert ms["auto"] <= 1.03 * better + 0.03, (m, n, k, ms)
    print("auto / better per point:", [w[0] for w in worst])


def test_finalize_async_pipelines_a_series_of_maps():
    """kpdi_finalize_async / kpdi_finalize_wait: the result of map i is collected after map i + 1 has been queued - the
    same results as kpdi_finalize, map by map; at most two results may be pending; the synchronous call still works in
    between."""
    from kikuchipy_amd import _lib

    rng = np.random.default_rng(23)
    dic = rng.random((3000, 24, 24), dtype=np.float32)
    maps = [rng.integers(0, 256, (m, 24, 24), dtype=np.uint8) for m in (300, 300, 77, 300, 512)]
    for with_comm in (False, True):  # (True: the RCCL all-gather + rank merge of a one-rank communicator inside the async half)
        with _lib.Context(0) as c:
            if with_comm:
                c.comm_init(0, 1, _lib.Context.comm_unique_id())
            c.set_problem(24, 24, None, _lib.METRIC_NCC, 10)
            want = []
            for e in maps:
                c.set_experimental(e)
                c.push_dictionary_chunk(dic[:1700], 0)
                c.push_dictionary_chunk(dic[1700:], 1700)
                want.append(c.finalize(10))
            got, pending = [], None
            for e in maps:
                c.set_experimental(e)
                c.push_dictionary_chunk(dic[:1700], 0)
                c.push_dictionary_chunk(dic[1700:], 1700)
                ticket = c.finalize_async(10)
                if pending is not None:
                    got.append(c.finalize_wait(pending))
                pending = ticket
            got.append(c.finalize_wait(pending))
            for (ws, wi), (gs, gi) in zip(want, got):
                assert gs.shape == ws.shape and np.array_equal(gs, ws) and np.array_equal(gi, wi)
            # two pending results are the limit; a collected ticket cannot be collected again
            c.set_experimental(maps[0])
            c.push_dictionary_chunk(dic, 0)
            t1 = c.finalize_async(10)
            t2 = c.finalize_async(10)
            with pytest.raises(_lib.KpdiError, match="already pending"):
                c.finalize_async(10)
            a = c.finalize_wait(t1)
            b = c.finalize_wait(t2)
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[1], c.finalize(10)[1])
            with pytest.raises(_lib.KpdiError, match="no result is pending"):
                c.finalize_wait(t1)


@pytest.mark.parametrize("metric,keep_n,sig", [("ncc", 20, (60, 60)), ("ndp", 40, (24, 20)), ("ncc", 5, (60, 60))])
def test_pipelined_front_half_of_a_series_of_maps(monkeypatch, metric, keep_n, sig):
    """A series of maps with device-resident inputs, queued ahead of the GPU (finalize_async), in which every step differs
    from its neighbours - other patterns, another dictionary block of another size (partial last tiles), one or two chunks
    per sweep, a host-pointer push, a recorded background step, a navigation mask, another keep_n: every result must
    equal what a second context returns for the same calls, bit for bit, and the oracle's.  (Written as the hazard test of
    a software-pipelined front half - the next map's preparation on a stream of its own into a second set of operand
    buffers, `KPDI_OVERLAP` - which was built, measured at ~1 % and reverted, profiles/r04_front_half_overlap.txt; it pins
    the pipelined single-stream loop just as well.)"""
    from kikuchipy_amd import _lib

    rng = np.random.default_rng(31)
    sy, sx = sig
    code = _lib.METRIC_NCC if metric == "ncc" else _lib.METRIC_NDP
    dic = rng.random((9000, sy, sx), dtype=np.float32)
    bg = rng.integers(1, 256, (sy, sx)).astype(np.float32)
    # (experimental patterns, [dictionary slices of the sweep], variation)
    steps = []
    for j in range(14):
        m = int(rng.choice([300, 512, 77, 1024]))
        a = int(rng.integers(0, 3000))
        n = int(rng.choice([2500, 2560, 3001, 1700, 4096]))
        chunks = [(a, a + n)] if j % 3 else [(a, a + n // 2), (a + n // 2, a + n)]
        steps.append((rng.integers(0, 256, (m, sy, sx), dtype=np.uint8), chunks,
                      {5: "host", 8: "background", 10: "navmask", 12: "keep"}.get(j, "")))

    def run(overlap):
        monkeypatch.setenv("KPDI_OVERLAP", "1" if overlap else "0")
        out = []
        with _lib.Context(0) as c:
            c.set_problem(sy, sx, None, code, keep_n)  # (reads KPDI_OVERLAP)
            d_dic = c.dev_alloc(dic.nbytes)
            c.h2d(d_dic, dic)
            d_exp = [c.dev_alloc(1024 * sy * sx) for _ in range(2)]
            row = sy * sx * 4
            pending = None
            for j, (e, chunks, what) in enumerate(steps):
                d = d_exp[j & 1]  # (two source buffers: the copy of map i + 1 may still run while the host fills the other)
                c.h2d(d, e)
                k = keep_n
                if what == "keep":
                    k = 3
                    c.set_keep_n(k)
                nav = None
                if what == "navmask":
                    nav = np.zeros(len(e), dtype=bool)
                    nav[::7] = True
                c.set_experimental_dev(d, e.dtype, len(e), nav)
                if what == "background":
                    c.remove_static_background(bg, _lib.OP_SUBTRACT, False)
                for a, b in chunks:
                    if what == "host":
                        c.push_dictionary_chunk(dic[a:b], a)
                    else:
                        c.push_dictionary_chunk_dev(d_dic + a * row, np.float32, b - a, a)
                ticket = c.finalize_async(k)
                if pending is not None:
                    out.append(c.finalize_wait(pending))
                pending = ticket
                if what == "keep":
                    out.append(c.finalize_wait(pending))
                    pending = None
                    c.set_keep_n(keep_n)
            if pending is not None:
                out.append(c.finalize_wait(pending))
        return out

    plain = run(False)
    piped = run(True)
    assert len(plain) == len(piped) == len(steps)
    for j, ((ws, wi), (gs, gi)) in enumerate(zip(plain, piped)):
        assert gs.shape == ws.shape and np.array_equal(gs, ws) and np.array_equal(gi, wi), f"step {j}: {steps[j][2] or 'plain'}"
    # and the oracle on a plain step
    e, chunks, _ = steps[1]
    (a, b), = chunks
    rs, ri = ko.dictionary_indexing(e, dic[a:b], metric=metric, keep_n=keep_n)
    ko.assert_topk_parity(piped[1][0], piped[1][1], rs, ri + a, atol=1e-5)
