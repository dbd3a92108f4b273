"""One interpreter, several GPUs: `kpdi_group` (include/kpdi.h) / `kikuchipy_amd._lib.Group` /
`dictionary_indexing(..., devices=...)`.

The reference's `EBSD.dictionary_indexing` is ONE call in ONE process (signals/ebsd.py:1827-1984; the
chunk loop of indexing/_dictionary_indexing.py:100-128).  A group shards every dictionary chunk over
its members and hands back one merged result; because the merge is a total order (score desc, index
asc) that result must equal the single-context result BIT FOR BIT - which is what these tests assert.
Members may share a device (peer-copy gather), so the whole multi-device code path - threads, chunk
assignment and queues, gather, merge, pipelined hand-over - runs on a 1-GPU box; the in-process RCCL
communicator is exercised with one member here and with one member per GPU in test_gpu_multigpu.py.
"""

import numpy as np
import pytest

from conftest import synth
from oracle import kpdi_oracle as ko

pytestmark = pytest.mark.gpu


def patterns(seed, n, shape=(12, 10), dtype=np.float32):
    rng = np.random.default_rng(seed)
    base = rng.random((n,) + shape)
    if np.issubdtype(dtype, np.integer):
        return (base * 255).astype(dtype)
    return base.astype(dtype)


def single(exp, dic, metric, keep_n, compute, signal_mask=None, nav_mask=None, start=0):
    from kikuchipy_amd import _lib

    with _lib.Context(0) as c:
        c.set_problem(exp.shape[-2], exp.shape[-1], signal_mask, metric, keep_n, compute)
        c.set_experimental(exp, nav_mask)
        c.push_dictionary_chunk(dic, start)
        return c.finalize(keep_n)


def test_eight_in_process_shards_equal_the_single_sweep_at_full_size():
    """configs[1] at full size (4096 x 100 000 x 60 x 60, ncc, keep_n 20): eight members on device 0,
    each sweeping its eighth of the dictionary, peer-copy gather + merge == one context's sweep."""
    from kikuchipy_amd import _lib

    exp, dic = synth(2024, 4096, 100000)
    s1, i1 = single(exp, dic, _lib.METRIC_NCC, 20, _lib.COMPUTE_F32)
    with _lib.Group([0] * 8) as g:
        assert len(g) == 8 and g.gather == "p2p" and "p2p" in g.describe()
        g.set_problem(60, 60, None, _lib.METRIC_NCC, 20, _lib.COMPUTE_F32)
        g.set_experimental(exp, None)
        g.set_profiling(True)
        g.push_dictionary_chunk(dic, 0)
        s8, i8 = g.finalize(20)
        cnt = g.counters()
    assert np.array_equal(s1, s8) and np.array_equal(i1, i8)
    # (a single pass nobody announced is still cut eight ways: pieces of three tile rounds each)
    # every member swept exactly its block of the chunk, member 0 merged eight lists
    assert cnt["gather_ranks"] == 8 and cnt["gather"] == "p2p"
    flops = [m["match_flops"] for m in cnt["members"]]
    assert flops == [2.0 * 4096 * 12500 * 3600] * 8, flops
    # and the oracle agrees on a sample of rows (float64 C oracle over the whole dictionary)
    from oracle import c_oracle

    rows = np.arange(0, 4096, 128)
    rs, ri = c_oracle.rows_topk_f64(exp, [(0, dic)], rows, "ncc", 20, None)
    ko.assert_topk_parity(s8[rows], i8[rows], rs, ri, atol=1e-5)


def test_members_on_different_kernel_forms_still_merge_to_the_single_sweep():
    """The "bit for bit" claim of include/kpdi.h rests on every f32 kernel form producing the same bits.  Here the members
    of one group sweep shares whose sizes make the planner pick DIFFERENT forms (match16.hip's 256 x 256 tiles with
    partial units for 50 000 patterns, match.hip's 128 x 256 tiles + quarter-tile tail for 6250, match16.hip + tailgemm.hip
    for 12 500, whatever it likes for the odd rest) - the counters say so - and the merged result is the single sweep's,
    bit for bit."""
    from kikuchipy_amd import _lib

    exp, dic = synth(2024, 4096, 100000)
    s1, i1 = single(exp, dic, _lib.METRIC_NCC, 20, _lib.COMPUTE_F32)
    sizes = [50000, 6250, 12500, 31147, 103]
    assert sum(sizes) == 100000 and _lib.plan_describe(4096, 12500).tail_gemm_rows == 212
    starts = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    with _lib.Group([0] * 5) as g:
        g.set_problem(60, 60, None, _lib.METRIC_NCC, 20, _lib.COMPUTE_F32)
        g.set_experimental(exp, None)
        d = []
        for mem, a, n in zip(g.members, starts, sizes):
            p = mem.dev_alloc(dic[a:a + n].nbytes)
            mem.h2d(p, dic[a:a + n])
            d.append(p)
        g.push_dictionary_chunk_dev(d, np.float32, sizes, [int(a) for a in starts])
        s, i = g.finalize(20)
        forms = [m["match_form"] for m in g.counters()["members"]]
    assert forms[0] == 3 and forms[1] == 0 and forms[2] == 3, forms  # (wide, classic, wide + tail kernel; the others whatever the planner likes)
    assert np.array_equal(s, s1) and np.array_equal(i, i1)


class Lazy:
    """A Dask-like dictionary: sliced along axis 0, chunks materialised by `.compute()` inside the loop."""

    def __init__(self, a, chunk, log=None):
        self._a, self.shape, self.ndim, self.chunksize = a, a.shape, a.ndim, (chunk,) + a.shape[1:]
        self.dtype, self.log = a.dtype, log if log is not None else []

    def __getitem__(self, sl):
        return Lazy(self._a[sl], self.chunksize[0], self.log)

    def compute(self):
        self.log.append(len(self._a))
        return np.array(self._a)  # a fresh temporary per chunk, like Dask's


def test_the_tutorial_call_on_eight_members_equals_the_single_sweep_at_full_size():
    """configs[1] through the reference tutorial's call shape - `n_per_iteration` = 3044 patterns per iteration
    (doc/tutorials/pattern_matching.ipynb:582; the loop of indexing/_dictionary_indexing.py:100-128) - on eight
    members: whole chunks go round the members (csrc/group_assign.h), and the merged result is the single sweep's,
    BIT FOR BIT, for an in-memory dictionary and for a lazy one whose chunks are temporaries."""
    import kikuchipy_amd as ka
    from kikuchipy_amd import _lib

    exp, dic = synth(2024, 4096, 100000)
    s1, i1 = single(exp, dic, _lib.METRIC_NCC, 20, _lib.COMPUTE_F32)
    made, stats = [], []
    real, real_release = _lib.make_engine, _lib.release_engine

    def make(*a, **k):
        made.append(real(*a, **k))
        return made[-1]

    def release(eng):  # (the call hands the engine it made back when it is done: look at its counters first)
        stats.append(eng.counters())
        real_release(eng)

    _lib.make_engine, _lib.release_engine = make, release
    try:
        calls = []
        got = ka.dictionary_indexing(exp, dic, keep_n=20, n_per_iteration=3044, devices=[0] * 8, verbose=False,
                                     progress=lambda done, total: calls.append((done, total)))
        assert calls == [(j, 33) for j in range(1, 34)]
        grp, cnt = made[-1], stats[-1]
        lazy = Lazy(dic, 3044)
        got_lazy = ka.dictionary_indexing(exp, lazy, keep_n=20, devices=[0] * 8, verbose=False)
    finally:
        _lib.make_engine, _lib.release_engine = real, real_release
    assert isinstance(grp, _lib.Group) and len(grp) == 8 and len(made) == 1  # (the second call took the first one's engine)
    assert np.array_equal(got.scores, s1) and np.array_equal(got.simulation_indices, i1)
    assert np.array_equal(got_lazy.scores, s1) and np.array_equal(got_lazy.simulation_indices, i1)
    assert lazy.log == [3044] * 32 + [100000 - 32 * 3044]
    # every member took its quota as 4 whole chunks + its 324 patterns of the last one (not 33 pieces of 380) - and swept
    # them TOGETHER (csrc/sweep.hip: small chunks wait for company): one launch set per member, as in a single-pass call
    per = [(m["match_flops"], m["match_launches"], m["coalesced_sweeps"]) for m in cnt["members"]]
    assert per == [(2.0 * 4096 * 12500 * 3600, 1, 1)] * 8, per


def test_generated_chunks_on_a_group_equal_the_single_device():
    """`get_patterns(..., chunk_shape=3044)`: a lazy dictionary whose chunks are SIMULATED on the member that
    sweeps them (only the rotations are queued)."""
    import kikuchipy_amd as ka

    rng = np.random.default_rng(11)
    mp = ka.EBSDMasterPattern(rng.random((2, 201, 201)).astype(np.float32), hemisphere="both")
    det = ka.EBSDDetector(shape=(60, 60), pc=(0.42, 0.78, 0.5))
    q = rng.standard_normal((25000, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    sim = mp.get_patterns(q, det, compute=False, chunk_shape=3044)
    assert sim.data.chunksize[0] == 3044
    exp = rng.integers(0, 256, (32, 32, 60, 60), dtype=np.uint8)
    s = ka.EBSD(exp)
    a = s.dictionary_indexing(sim, keep_n=20, devices=[0], verbose=False)
    b = s.dictionary_indexing(sim, keep_n=20, devices=[0] * 8, verbose=False)
    assert np.array_equal(a.scores, b.scores) and np.array_equal(a.simulation_indices, b.simulation_indices)


def test_queued_chunks_borrowed_buffers_and_late_errors():
    """The mechanics under the chunked call: a push returns at once and BORROWS the array (tickets), a synchronous
    push returns when its buffer has been consumed, and a queued chunk that fails is reported - with its member
    named - by the next joining call."""
    import ctypes as C

    from kikuchipy_amd import _lib

    dic = patterns(7, 4000)
    exp = patterns(8, 100, dtype=np.uint8)
    s1, i1 = single(exp, dic, _lib.METRIC_NCC, 10, _lib.COMPUTE_F32)
    lib = _lib.load()
    with _lib.Group([0] * 3) as g:
        g.set_problem(12, 10, None, _lib.METRIC_NCC, 10, _lib.COMPUTE_F32)
        g.set_experimental(exp, None)
        g.set_dictionary_size(4000)
        tickets = []
        for a in range(0, 4000, 500):
            chunk = np.array(dic[a:a + 500])  # a temporary: the Group keeps it alive while it is borrowed
            g.push_dictionary_chunk(chunk, a)
            tickets.append(g._borrowed[-1][0])
            del chunk
        assert tickets == list(range(1, 9))
        s, i = g.finalize(10)
        assert not g._borrowed
        t = C.c_int64(-1)
        _lib.check(lib.kpdi_group_chunks_consumed(g._h, C.byref(t)))
        assert t.value == 8
        assert np.array_equal(s, s1) and np.array_equal(i, i1)
        # the C ABI's synchronous form: the buffer may be overwritten as soon as the call returns
        g.reset_topk()
        buf = np.empty((500, 12, 10), dtype=np.float32)
        for a in range(0, 4000, 500):
            buf[:] = dic[a:a + 500]
            _lib.check(lib.kpdi_group_push_dictionary_chunk(g._h, buf.ctypes.data_as(C.c_void_p), 2, 500, a))
        s, i = g.finalize(10)
        assert np.array_equal(s, s1) and np.array_equal(i, i1)
        # a queued chunk that cannot be pushed (dictionary index beyond int32): the push itself returns, the failure
        # surfaces at the next joining call and names the member
        g.reset_topk()
        g.set_dictionary_size(0)
        g.push_dictionary_chunk(dic[:100], 2**31 - 50)
        with pytest.raises(_lib.KpdiError, match=r"group member \d of 3.*int32"):
            g.synchronize()
        g.reset_topk()  # the group stays usable
        g.set_dictionary_size(4000)
        g.push_dictionary_chunk(dic, 0)
        s, i = g.finalize(10)
        assert np.array_equal(s, s1) and np.array_equal(i, i1)


def test_a_refused_finalize_leaves_the_collective_in_step():
    """Argument and slot errors of a group finalize are found BEFORE any member queues its half of the gather
    (kpdi::finalize_precheck): the group is still usable and the next result is right."""
    import ctypes as C

    from kikuchipy_amd import _lib

    dic = patterns(3, 2500)
    exp = patterns(4, 200, dtype=np.uint8)
    s1, i1 = single(exp, dic, _lib.METRIC_NCC, 20, _lib.COMPUTE_F32)
    lib = _lib.load()
    for gather, ids in (("rccl", [0]), ("p2p", [0, 0, 0])):
        with _lib.Group(ids, gather=gather) as g:
            g.set_problem(12, 10, None, _lib.METRIC_NCC, 20, _lib.COMPUTE_F32)
            g.set_experimental(exp, None)
            g.push_dictionary_chunk(dic, 0)
            assert lib.kpdi_group_finalize(g._h, None, None) != 0 and "NULL" in _lib.last_error()
            t1, t2 = g.finalize_async(20), g.finalize_async(20)
            with pytest.raises(_lib.KpdiError, match="two results are already pending"):
                g.finalize_async(20)
            with pytest.raises(_lib.KpdiError, match="two results are already pending"):
                g.finalize(20)
            out = C.c_double()
            assert lib.kpdi_group_finalize_f64(g._h, C.byref(out), C.byref(out)) != 0 and "KPDI_COMPUTE_F64" in _lib.last_error()
            for t in (t1, t2):
                s, i = g.finalize_wait(t)
                assert np.array_equal(s, s1) and np.array_equal(i, i1)
            s, i = g.finalize(20)
            assert np.array_equal(s, s1) and np.array_equal(i, i1)


@pytest.mark.parametrize("members,metric,keep_n,compute,masked,chunk", [
    (3, "ncc", 10, "f32", False, 1000),
    (4, "ndp", 1, "f32", True, 700),
    (2, "ncc", 50, "f32", True, 3000),     # keep_n > 32: bounded passes on every member
    (5, "ncc", 20, "f16x2", False, 1300),
    (3, "ndp", 20, "f16", False, 3000),
    (4, "ncc", 12, "f64", True, 900),      # float64 lists gathered and merged in double
    (8, "ncc", 5, "f32", False, 5),        # chunks smaller than the group: some members get nothing
])
def test_group_equals_single_context(members, metric, keep_n, compute, masked, chunk):
    """Host-level API: `devices=[0, 0, ...]` against `device=0` - chunked, masked, every arithmetic."""
    import kikuchipy_amd as ka

    dic = patterns(1, 3000 if chunk > 5 else 23)
    exp = patterns(2, 63, dtype=np.uint8).reshape(7, 9, 12, 10)
    signal_mask = nav_mask = None
    if masked:
        signal_mask = np.zeros((12, 10), dtype=bool)
        signal_mask[:2] = True
        signal_mask[5, 3:7] = True
        nav_mask = np.zeros((7, 9), dtype=bool)
        nav_mask[1, 2] = nav_mask[6, 8] = True
    kw = dict(metric=metric, keep_n=keep_n, n_per_iteration=chunk, navigation_mask=nav_mask, signal_mask=signal_mask,
              verbose=False)
    if compute == "f64":
        kw["dtype"] = np.float64
    else:
        kw["compute"] = compute
    one = ka.dictionary_indexing(exp, dic, device=0, **kw)
    grp = ka.dictionary_indexing(exp, dic, devices=[0] * members, **kw)
    assert grp.scores.dtype == one.scores.dtype
    assert np.array_equal(one.scores, grp.scores)
    assert np.array_equal(one.simulation_indices, grp.simulation_indices)


def test_in_process_rccl_communicator_with_one_member():
    """gather="rccl" with ONE device: ncclCommInitAll + the all-gather of kpdi_finalize inside the process."""
    from kikuchipy_amd import _lib

    dic = patterns(3, 2500)
    exp = patterns(4, 200, dtype=np.uint8)
    s1, i1 = single(exp, dic, _lib.METRIC_NCC, 20, _lib.COMPUTE_F32, start=40)
    with _lib.Group([0], gather="rccl") as g:
        assert g.gather == "rccl"
        g.set_problem(12, 10, None, _lib.METRIC_NCC, 20, _lib.COMPUTE_F32)
        g.set_experimental(exp, None)
        g.push_dictionary_chunk(dic, 40)
        s, i = g.finalize(20)
        c = g.counters()
        assert c["comm_ranks"] == 1 and c["gather_ranks"] == 1
    assert np.array_equal(s, s1) and np.array_equal(i, i1)
    with pytest.raises(_lib.KpdiError, match="share a device"):
        _lib.Group([0, 0], gather="rccl")  # RCCL refuses duplicate devices: the message says what to use instead


def test_a_group_tries_its_rccl_communicator_before_it_relies_on_it(monkeypatch):
    """kpdi_group_create runs one small all-gather through a new communicator (every member on a thread of its own, under
    $KPDI_COMM_TIMEOUT): asked for by name, a failure is an error that names the member; chosen automatically (distinct
    devices - not reachable on a one-GPU box), the group takes the peer-copy gather instead and says why."""
    from kikuchipy_amd import _lib

    monkeypatch.setenv("KPDI_GROUP_SELFTEST_FAIL", "1")
    with pytest.raises(_lib.KpdiError, match=r"device 0 \(group member 0 of 1\): injected failure"):
        _lib.Group([0], gather="rccl")
    monkeypatch.delenv("KPDI_GROUP_SELFTEST_FAIL")
    monkeypatch.setenv("KPDI_COMM_TIMEOUT", "20")
    with _lib.Group([0], gather="rccl") as g:  # (the real self-test, one rank)
        assert g.gather == "rccl"


def test_pipelined_series_of_maps_on_a_group():
    """finalize_async / finalize_wait on a group: map i's merged result is collected after map i + 1 has been
    queued on every member (what bench.py --single-process does); device-resident inputs per member."""
    from kikuchipy_amd import _lib

    n_dev, n, m, k = 4, 6000, 300, 20
    dic = patterns(5, n)
    maps = [patterns(10 + j, m, dtype=np.uint8) for j in range(5)]
    want = [single(e, dic, _lib.METRIC_NCC, k, _lib.COMPUTE_F32) for e in maps]
    with _lib.Group([0] * n_dev) as g:
        g.set_problem(12, 10, None, _lib.METRIC_NCC, k, _lib.COMPUTE_F32)
        shares = [_lib.Group.chunk_share(n, i, n_dev) for i in range(n_dev)]
        d_dic, d_exp = [], []
        for mem, (a, b) in zip(g.members, shares):
            d = mem.dev_alloc(dic[a:b].nbytes)
            mem.h2d(d, dic[a:b])
            d_dic.append(d)
            d_exp.append(mem.dev_alloc(maps[0].nbytes))
        got, pending = [], None
        for e in maps:
            for mem, d in zip(g.members, d_exp):
                mem.synchronize()  # (the previous map's kernels have read this buffer)
                mem.h2d(d, e)
            g.set_experimental_dev(d_exp, e.dtype, m)
            g.push_dictionary_chunk_dev(d_dic, np.float32, [b - a for a, b in shares], [a for a, _ in shares])
            ticket = g.finalize_async(k)
            if pending is not None:
                got.append(g.finalize_wait(pending))
            pending = ticket
        got.append(g.finalize_wait(pending))
        # a synchronous finalize between two async ones must not disturb a pending ticket
        t = g.finalize_async(k)
        s_sync, i_sync = g.finalize(k)
        s_t, i_t = g.finalize_wait(t)
        assert np.array_equal(s_sync, s_t) and np.array_equal(i_sync, i_t)
    for (s, i), (ws, wi) in zip(got, want):
        assert np.array_equal(s, ws) and np.array_equal(i, wi)


def test_group_preprocessing_resident_and_generated_dictionaries():
    """The recorded background steps, `ResidentDictionary(devices=...)` and a `ProjectedDictionary` simulated on
    every member: each equals the single-device result."""
    import kikuchipy_amd as ka
    from kikuchipy_amd import _lib

    rng = np.random.default_rng(8)
    exp = rng.integers(0, 256, (40, 60, 60), dtype=np.uint8)
    dic = rng.random((2000, 60, 60), dtype=np.float32)
    bg = rng.integers(1, 256, (60, 60)).astype(np.float32)
    # static + dynamic background recorded on every member, fused with the preparation there
    res = []
    for engine in (_lib.Context(0), _lib.Group([0, 0, 0])):
        with engine as c:
            c.set_problem(60, 60, None, _lib.METRIC_NCC, 8, _lib.COMPUTE_F32)
            c.set_experimental(exp, None)
            c.remove_static_background(bg, _lib.OP_SUBTRACT, False)
            c.remove_dynamic_background(_lib.OP_SUBTRACT, _lib.DOMAIN_FREQUENCY, 0.0, 4.0)
            c.push_dictionary_chunk(dic, 0)
            res.append(c.finalize(8) + (c.get_experimental(),))
    for a, b in zip(*res):
        assert np.array_equal(a, b)
    # a dictionary prepared once, its chunks handed to the members whole (quota 667 / 667 / 666)
    r1 = ka.ResidentDictionary(dic, "ncc", n_per_iteration=900, device=0)
    r3 = ka.ResidentDictionary(dic, "ncc", n_per_iteration=900, devices=[0, 0, 0])
    assert r3.held[0] == 2000
    for seed in (1, 2):
        e = np.random.default_rng(seed).integers(0, 256, (5, 6, 60, 60), dtype=np.uint8)
        a = ka.dictionary_indexing(e, r1, "ncc", 10, verbose=False)
        b = ka.dictionary_indexing(e, r3, "ncc", 10, verbose=False)
        assert np.array_equal(a.scores, b.scores) and np.array_equal(a.simulation_indices, b.simulation_indices)
    # a lazy dictionary: every member simulates its share of the rotations in its own memory
    mp = ka.EBSDMasterPattern(rng.random((2, 101, 101)).astype(np.float32), hemisphere="both")
    det = ka.EBSDDetector(shape=(60, 60), pc=(0.42, 0.78, 0.5))
    q = rng.standard_normal((700, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    sim = mp.get_patterns(q, det, compute=False)
    s = ka.EBSD(exp.reshape(5, 8, 60, 60))
    a = s.dictionary_indexing(sim, keep_n=6, devices=[0], verbose=False)
    b = s.dictionary_indexing(sim, keep_n=6, devices=[0, 0, 0, 0], verbose=False)
    assert np.array_equal(a.scores, b.scores) and np.array_equal(a.simulation_indices, b.simulation_indices)
    assert list(s._groups) == [(0, 0, 0, 0)]  # the signal keeps its group (and the group its communicator)
    b2 = s.dictionary_indexing(sim, keep_n=6, devices=[0, 0, 0, 0], verbose=False)
    assert np.array_equal(b.scores, b2.scores) and len(s._groups) == 1


def test_default_is_every_visible_device(monkeypatch):
    """A call that names no device runs on `default_devices()`: $KPDI_DEVICES, else every visible GPU."""
    import kikuchipy_amd as ka
    from kikuchipy_amd import _lib
    from kikuchipy_amd.indexing import _dictionary_indexing as di

    assert _lib.default_devices() == list(range(_lib.device_count()))
    monkeypatch.setenv("KPDI_DEVICES", "0,0,0")
    assert _lib.default_devices() == [0, 0, 0]
    monkeypatch.setattr(di, "GROUP_MIN_COMPARISONS", 0)
    made = []
    real = _lib.make_engine
    monkeypatch.setattr(_lib, "make_engine", lambda *a, **k: made.append(real(*a, **k)) or made[-1])
    dic = patterns(1, 500)
    exp = patterns(2, 20, dtype=np.uint8)
    got = ka.dictionary_indexing(exp, dic, keep_n=5, verbose=False)
    assert isinstance(made[-1], _lib.Group) and len(made[-1]) == 3
    one = ka.dictionary_indexing(exp, dic, keep_n=5, device=0, verbose=False)
    assert isinstance(made[-1], _lib.Context) and not isinstance(made[-1], _lib.Group)
    assert np.array_equal(got.scores, one.scores) and np.array_equal(got.simulation_indices, one.simulation_indices)


def test_a_members_error_names_the_member():
    from kikuchipy_amd import _lib

    with _lib.Group([0, 0]) as g:
        with pytest.raises(_lib.KpdiError, match=r"group member \d of 2.*kpdi_set_problem"):
            g.set_experimental(patterns(1, 4, dtype=np.uint8), None)
        with pytest.raises(_lib.KpdiError, match="one entry per member"):
            g.push_dictionary_chunk_dev([0], np.float32, [1], [0])
    with pytest.raises(_lib.KpdiError, match="out of range"):
        _lib.Group([0, 99])


def test_background_removal_over_a_group():
    """`remove_static_background` / `remove_dynamic_background` with `devices=`: every GPU takes a contiguous block of the
    patterns (they are independent) - the output is the single-device output, byte for byte, for every dtype, with fewer
    patterns than members, and through the array-level functions."""
    import kikuchipy_amd as ka
    from kikuchipy_amd import _lib
    from kikuchipy_amd.pattern import _pattern

    rng = np.random.default_rng(21)
    for dtype, nav in ((np.uint8, (7, 9)), (np.uint16, (50,)), (np.float32, (3, 4))):
        hi = 255 if dtype == np.uint8 else (60000 if dtype == np.uint16 else 1)
        data = (rng.random(nav + (60, 60)) * hi).astype(dtype)
        bg = (rng.random((60, 60)) * hi * 0.5 + hi * 0.25).astype(dtype)
        one = ka.EBSD(data.copy(), static_background=bg, device=0)
        grp = ka.EBSD(data.copy(), static_background=bg, devices=[0, 0, 0])
        for s in (one, grp):
            s.remove_static_background(scale_bg=(dtype == np.uint16))
            s.remove_dynamic_background(operation="divide" if dtype == np.float32 else "subtract")
        assert grp.data.dtype == one.data.dtype and grp.data.shape == data.shape
        assert np.array_equal(one.data, grp.data)
        assert list(grp._groups) == [(0, 0, 0)] and not one._groups
    few = (rng.random((2, 60, 60)) * 255).astype(np.uint8)  # fewer patterns than members: one context does it
    with _lib.Group([0] * 4) as g:
        a = _pattern.remove_dynamic_background(few, contexts=g.members)
    assert np.array_equal(a, _pattern.remove_dynamic_background(few, device=0))


def test_a_held_chunk_larger_than_the_announced_dictionary_keeps_every_row():
    """with a stale / too small announced size the assignment names a member TWICE for one chunk (the
    overflow goes whole to the least-loaded member: csrc/group_assign.h) - `Group.assign_chunk(2, 4, [0, 0], 6)` =
    [(0, 0, 2), (1, 2, 2), (0, 4, 2)] - and the hold path used to keep only the member's last piece: rows silently left
    the resident dictionary.  Every piece must be held."""
    from kikuchipy_amd import _lib

    assert _lib.Group.assign_chunk(2, 4, [0, 0], 6) == [(0, 0, 2), (1, 2, 2), (0, 4, 2)]
    dic = patterns(3, 900)
    exp = patterns(4, 40, dtype=np.uint8)
    with _lib.Context(0) as c:
        c.set_problem(12, 10, None, _lib.METRIC_NCC, 10)
        c.set_experimental(exp)
        c.push_dictionary_chunk(dic, 0)
        want = c.finalize(10)
    with _lib.Group([0, 0, 0]) as g:
        g.set_problem(12, 10, None, _lib.METRIC_NCC, 10)
        g.set_dictionary_size(500)             # announced: 500; held below: 900 (in two chunks, the second beyond it)
        g.set_experimental(exp)
        g.hold_dictionary_chunk(dic[:700], 0)  # 500 by quota + 200 overflow: a member holds two non-adjacent pieces
        g.hold_dictionary_chunk(dic[700:], 700)
        assert g.held_size()[0] == 900
        g.sweep_held()
        got = g.finalize(10)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
