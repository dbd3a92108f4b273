"""A dictionary prepared once and kept in HBM (kpdi_hold_* / kpdi_sweep_held,
`kikuchipy_amd.ResidentDictionary`): indexing against it must give exactly what
pushing the same patterns again gives, map after map, and both must agree with
the oracle (the reference's loop, indexing/_dictionary_indexing.py:94-128)."""

import numpy as np
import pytest

from conftest import load_golden
from oracle import kpdi_oracle as ko

pytestmark = pytest.mark.gpu


def patterns(seed, n, shape=(12, 10), dtype=np.float32):
    rng = np.random.default_rng(seed)
    base = rng.random((n,) + shape)
    if np.issubdtype(dtype, np.integer):
        return (base * 255).astype(dtype)
    return base.astype(dtype)


@pytest.mark.parametrize("metric,keep_n,compute,masked", [
    ("ncc", 10, "f32", False), ("ndp", 1, "f32", True), ("ncc", 50, "f32", True), ("ncc", 20, "f16x2", False),
])
def test_series_of_maps_equals_pushing_again(metric, keep_n, compute, masked):
    import kikuchipy_amd as ka

    dic = patterns(1, 3000)
    signal_mask = None
    if masked:
        signal_mask = np.zeros((12, 10), dtype=bool)
        signal_mask[:2] = True
        signal_mask[5, 3:7] = True
    resident = ka.ResidentDictionary(dic, metric, signal_mask, n_per_iteration=1100, compute=compute)
    assert resident.held[0] == 3000 and resident.held[1] > 0
    for seed, nav, dtype in ((2, (7, 9), np.uint8), (3, (40,), np.float32), (4, (5, 5), np.uint8)):
        exp = patterns(seed, int(np.prod(nav)), dtype=dtype).reshape(nav + (12, 10))
        nav_mask = None
        if seed == 4:
            nav_mask = np.zeros(nav, dtype=bool)
            nav_mask[1, 2] = nav_mask[4, 4] = True
        got = ka.dictionary_indexing(exp, resident, metric, keep_n, navigation_mask=nav_mask, verbose=False)
        again = ka.dictionary_indexing(exp, dic, metric, keep_n, n_per_iteration=1100, navigation_mask=nav_mask,
                                       signal_mask=signal_mask, compute=compute, verbose=False)
        assert np.array_equal(got.scores, again.scores)
        assert np.array_equal(got.simulation_indices, again.simulation_indices)
        if compute == "f32":
            rs, ri = ko.dictionary_indexing(exp, dic, metric=metric, keep_n=keep_n, signal_mask=signal_mask,
                                            navigation_mask=nav_mask)
            sel = slice(None) if nav_mask is None else ~nav_mask.ravel()
            ko.assert_topk_parity(np.asarray(got.scores).reshape(-1, keep_n)[sel],
                                  np.asarray(got.simulation_indices).reshape(-1, keep_n)[sel], rs, ri, atol=1e-5)
    resident.release()
    assert resident.held == (0, 0)
    with pytest.raises(ValueError, match="released"):
        ka.dictionary_indexing(exp, resident, metric, keep_n, verbose=False)


def test_small_held_chunks_are_prepared_together():
    """Chunks handed over to be HELD that are smaller than two tile rounds (a lazy dictionary's Dask chunks usually are:
    3044 patterns in the reference's tutorial) wait for each other and become ONE resident chunk - one launch set per
    map instead of one per chunk - whose rows the merge maps back to dictionary indices.  Same result, bit for bit, as
    pushing the chunks, for one-pass and multi-pass keep_n, host / device / generated chunks, any order."""
    from kikuchipy_amd import _lib

    rng = np.random.default_rng(9)
    exp = rng.integers(0, 256, (300, 24, 20), dtype=np.uint8)
    dic = rng.random((6000, 24, 20), dtype=np.float32)
    dic[4100] = dic[33]
    chunks = [(a, min(a + 450, 6000)) for a in range(0, 6000, 450)]
    with _lib.Context(0) as c:
        for keep_n, compute in ((20, _lib.COMPUTE_F32), (40, _lib.COMPUTE_F32), (8, _lib.COMPUTE_F16)):
            c.set_problem(24, 20, None, _lib.METRIC_NCC, keep_n, compute)
            c.set_experimental(exp)
            c.push_dictionary_chunk(dic, 0)
            ref = c.finalize(keep_n)
            for order, dev in ((chunks, False), (chunks[::2] + chunks[1::2], True)):
                c.release_held()
                d = None
                if dev:
                    d = c.dev_alloc(dic.nbytes)
                    c.h2d(d, dic)
                for a, b in order:  # (held BEFORE the experimental set of the next map is known: set-up order is free)
                    if dev:
                        c.hold_dictionary_chunk_dev(d + a * dic[0].nbytes, dic.dtype, b - a, a)
                    else:
                        c.hold_dictionary_chunk(dic[a:b], a)
                assert c.held_size()[0] == 6000
                for _ in range(2):  # map after map
                    c.set_experimental(exp)
                    c.reset_counters()
                    c.sweep_held()
                    got = c.finalize(keep_n)
                    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
                held_chunks = 1 if order is chunks else 2  # the step back in the dictionary closes the first group
                if keep_n <= 32:  # (one match launch set per held chunk and map; bounded passes launch more)
                    assert c.counters()["match_launches"] == held_chunks
                if dev:
                    c.dev_free(d)
        c.release_held()
        assert c.held_size() == (0, 0)


def test_chunk_larger_than_one_upload_piece():
    """A held chunk is uploaded in pieces of 192 tiles and prepared piece by piece into one buffer."""
    from kikuchipy_amd import _lib

    n = 192 * 128 + 5000
    dic = patterns(5, n, shape=(8, 8))
    exp = patterns(6, 300, shape=(8, 8))
    ctx = _lib.Context(0)
    ctx.set_problem(8, 8, None, _lib.METRIC_NCC, 10)
    ctx.set_experimental(exp, None)
    ctx.hold_dictionary_chunk(dic, 100)
    ctx.sweep_held()
    s1, i1 = ctx.finalize(10)
    ctx.reset_topk()
    ctx.push_dictionary_chunk(dic, 100)
    s2, i2 = ctx.finalize(10)
    assert np.array_equal(s1, s2) and np.array_equal(i1, i2)
    rs, ri = ko.dictionary_indexing(exp, dic, keep_n=10)
    ko.assert_topk_parity(s1, i1, rs, ri + 100, atol=1e-5)


def test_hold_from_device_memory_and_problem_changes():
    from kikuchipy_amd import _lib

    dic = patterns(7, 700)
    exp = patterns(8, 50)
    ctx = _lib.Context(0)
    with pytest.raises(_lib.KpdiError, match="kpdi_set_problem"):
        ctx.hold_dictionary_chunk(dic, 0)
    ctx.set_problem(12, 10, None, _lib.METRIC_NDP, 5)
    with pytest.raises(_lib.KpdiError, match="no resident dictionary"):
        ctx.set_experimental(exp, None)
        ctx.sweep_held()
    d = ctx.dev_alloc(dic.nbytes)
    ctx.h2d(d, dic)
    ctx.hold_dictionary_chunk_dev(d, np.float32, 400, 0)
    ctx.hold_dictionary_chunk_dev(d + 400 * 120 * 4, np.float32, 300, 400)
    ctx.dev_free(d)  # the held chunks are prepared copies
    assert ctx.held_size()[0] == 700
    ctx.sweep_held()
    s, i = ctx.finalize(5)
    rs, ri = ko.dictionary_indexing(exp, dic, metric="ndp", keep_n=5)
    ko.assert_topk_parity(s, i, rs, ri, atol=1e-5)
    # the same problem again (what every prepare_experimental() does) keeps the chunks ...
    ctx.set_problem(12, 10, None, _lib.METRIC_NDP, 1)
    assert ctx.held_size()[0] == 700
    # ... another metric, mask or arithmetic does not: the prepared layout depends on them
    ctx.set_problem(12, 10, None, _lib.METRIC_NCC, 1)
    assert ctx.held_size() == (0, 0)
    ctx.hold_dictionary_chunk(dic, 0)
    mask = np.zeros((12, 10), dtype=bool)
    mask[0] = True
    ctx.set_problem(12, 10, mask, _lib.METRIC_NCC, 1)
    assert ctx.held_size() == (0, 0)
    ctx.hold_dictionary_chunk(dic, 0)
    ctx.set_problem(12, 10, mask, _lib.METRIC_NCC, 1)
    assert ctx.held_size()[0] == 700
    ctx.set_problem(12, 10, mask, _lib.METRIC_NCC, 1, _lib.COMPUTE_F16X2)
    assert ctx.held_size() == (0, 0)


def test_simulated_resident_dictionary():
    """Master pattern + rotations -> patterns simulated in device memory and held."""
    import kikuchipy_amd as ka

    p = load_golden("projection.npz")
    mp = ka.EBSDMasterPattern(np.stack([p["mp_upper"], p["mp_lower"]]), phase_name="ni")
    det = ka.EBSDDetector(shape=(60, 60), pc=(0.4210, 0.7794, 0.5049), sample_tilt=70)
    rng = np.random.default_rng(11)
    q = rng.standard_normal((1500, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    sim = mp.get_patterns(q, det, chunk_shape=600)
    signal_mask = ~ka.filters.Window("circular", (60, 60)).astype(bool)
    resident = ka.ResidentDictionary.from_signal(sim, "ncc", signal_mask)
    assert resident.phase_name == "ni" and resident.held[0] == 1500
    g1 = load_golden("config1_ni.npz")
    for exp in (g1["exp"].reshape(3, 3, 60, 60), g1["exp"].reshape(9, 60, 60)[::-1][:4]):
        s = ka.EBSD(np.ascontiguousarray(exp))
        got = s.dictionary_indexing(resident, keep_n=5, verbose=False)
        again = s.dictionary_indexing(sim, keep_n=5, signal_mask=signal_mask, verbose=False)
        assert np.array_equal(got.scores, again.scores)
        assert np.array_equal(got.simulation_indices, again.simulation_indices)
        assert np.array_equal(got.rotations, again.rotations) and got.phase_name == "ni"


def test_call_must_match_what_was_prepared():
    import kikuchipy_amd as ka

    dic = patterns(9, 200)
    exp = patterns(10, 10)
    mask = np.zeros((12, 10), dtype=bool)
    mask[0] = True
    resident = ka.ResidentDictionary(dic, "ndp", mask)
    with pytest.raises(ValueError, match="prepared for NormalizedDotProductMetric"):
        ka.dictionary_indexing(exp, resident, "ncc", verbose=False)
    with pytest.raises(ValueError, match="signal mask differs"):
        ka.dictionary_indexing(exp, resident, "ndp", signal_mask=~mask, verbose=False)
    with pytest.raises(ValueError, match="signal shapes must be identical"):
        ka.dictionary_indexing(patterns(10, 10, shape=(10, 12)), resident, "ndp", verbose=False)
    ok = ka.dictionary_indexing(exp, resident, resident.metric, 3, signal_mask=mask, verbose=False)
    assert ok.scores.shape == (10, 3)
