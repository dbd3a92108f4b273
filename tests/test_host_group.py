"""`devices=` of the host layer on CPU: which engine a call gets (`pick_devices`, $KPDI_DEVICES, the size
threshold), the chunk loop feeding a group, the block assignment of the library (`kpdi_group_chunk_share`, host
code), and the merged result - with tests/_standin_engine.StandInGroup (the oracle behind the `_lib.Group`
interface, its members driven from threads) in place of the GPUs.  The GPU counterpart is tests/test_gpu_group.py."""

import numpy as np
import pytest

from _standin_engine import StandInContext, StandInGroup
from oracle import kpdi_oracle as ko


def test_chunk_share_is_contiguous_and_even():
    from kikuchipy_amd import _lib
    from kikuchipy_amd.parallel import shard_range

    for n, n_dev in ((100000, 8), (7, 8), (0, 3), (12345, 5), (1, 1)):
        parts = [_lib.Group.chunk_share(n, i, n_dev) for i in range(n_dev)]
        assert parts == [shard_range(n, i, n_dev) for i in range(n_dev)]  # the split of the multi-process form
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        sizes = [b - a for a, b in parts]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(_lib.KpdiError):
        _lib.Group.chunk_share(10, 3, 3)


def test_resolve_devices(monkeypatch):
    from kikuchipy_amd import _lib

    assert _lib.resolve_devices(None) is None
    assert _lib.resolve_devices("all") == [0]  # no GPU here: the one device a call would fail on, loudly
    assert _lib.resolve_devices("0, 2 3") == [0, 2, 3]
    assert _lib.resolve_devices(3) == [0, 1, 2]
    assert _lib.resolve_devices([1, 1]) == [1, 1]
    for bad in ("gpu0", 0, []):
        with pytest.raises(_lib.KpdiError):
            _lib.resolve_devices(bad)
    monkeypatch.setenv("KPDI_DEVICES", "4")
    assert _lib.default_devices() == [4]  # (a string names ids; a COUNT is the int form, devices=4)


def test_default_devices_under_a_launcher(monkeypatch, caplog):
    """A process started by a launcher (one process per GPU) that names no device stays on ONE GPU - its local rank's -
    instead of opening a context on every GPU like its siblings would; $KPDI_DEVICES still overrides; the choice is
    logged once."""
    import logging

    from kikuchipy_amd import _lib

    for v in _lib.LAUNCHER_VARIABLES + ("KPDI_DEVICES",):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setattr(_lib, "device_count", lambda: 8)
    _lib._logged_defaults.clear()
    with caplog.at_level(logging.INFO, logger="kikuchipy_amd"):
        assert _lib.default_devices() == list(range(8)) and not _lib.under_a_launcher()
        assert _lib.default_devices() == list(range(8))
        monkeypatch.setenv("WORLD_SIZE", "8")
        monkeypatch.setenv("LOCAL_RANK", "5")
        assert _lib.under_a_launcher() and _lib.default_devices() == [5]
        monkeypatch.setenv("LOCAL_RANK", "11")  # more ranks than GPUs: they share
        assert _lib.default_devices() == [3]
        monkeypatch.delenv("LOCAL_RANK")
        monkeypatch.delenv("WORLD_SIZE")
        monkeypatch.setenv("SLURM_PROCID", "2")
        monkeypatch.setenv("SLURM_LOCALID", "2")
        assert _lib.default_devices() == [2]
        monkeypatch.setenv("KPDI_DEVICES", "0,1")
        assert _lib.default_devices() == [0, 1]
    said = [r.getMessage() for r in caplog.records]
    assert len(said) == 5 and "every visible GPU" in said[0] and "launcher" in said[1] and "KPDI_DEVICES" in said[-1]


def test_pick_devices(monkeypatch):
    from kikuchipy_amd.indexing._dictionary_indexing import GROUP_MIN_COMPARISONS, pick_devices

    big = GROUP_MIN_COMPARISONS
    monkeypatch.setenv("KPDI_DEVICES", "0,1,2,3")
    assert pick_devices(None, None, None, 4096, big) == (0, [0, 1, 2, 3])     # nothing said: every device
    assert pick_devices(None, None, None, 9, 1000) == (0, None)               # ... unless the job is tiny
    assert pick_devices(2, None, None, 4096, big) == (2, None)                # device=: that one
    assert pick_devices(None, [1, 3], None, 9, 1000) == (0, [1, 3])           # devices=: those, whatever the size
    assert pick_devices(None, "all", object(), 4096, big) == (0, None)        # comm: one process per GPU
    assert pick_devices(5, None, object(), 4096, big) == (5, None)


@pytest.mark.parametrize("metric,keep_n,chunk,masked", [("ncc", 7, 250, False), ("ndp", 3, 999, True), ("ncc", 5, 3, False)])
def test_dictionary_indexing_over_a_group(monkeypatch, metric, keep_n, chunk, masked):
    import kikuchipy_amd as ka
    from kikuchipy_amd import _lib

    made = []

    def make_engine(device=0, devices=None, gather=None):
        ids = _lib.resolve_devices(devices)
        made.append(StandInContext(device) if ids is None else StandInGroup(ids))
        return made[-1]

    monkeypatch.setattr(_lib, "make_engine", make_engine)
    rng = np.random.default_rng(3)
    exp = rng.integers(0, 256, (4, 5, 12, 10), dtype=np.uint8)
    dic = rng.random((1000 if chunk > 3 else 17, 12, 10), dtype=np.float32)
    nav = sig = None
    if masked:
        nav = np.zeros((4, 5), dtype=bool)
        nav[2, 2] = True
        sig = np.zeros((12, 10), dtype=bool)
        sig[0] = True
    got = ka.dictionary_indexing(exp, dic, metric, keep_n, n_per_iteration=chunk, navigation_mask=nav, signal_mask=sig,
                                 devices=[0, 0, 0], verbose=False)
    grp = made[-1]
    assert isinstance(grp, StandInGroup) and len(grp) == 3 and len(grp.threads_seen) > 1
    # the call made this engine itself: it handed it back - kept idle for the next call on the same devices - and
    # clear_engine_cache() closes it (threads, contexts, communicator)
    assert not grp._pool._shutdown and _lib._ENGINE_POOL[((0, 0, 0), None)] is grp
    again = ka.dictionary_indexing(exp, dic, metric, keep_n, n_per_iteration=chunk, navigation_mask=nav, signal_mask=sig,
                                   devices=[0, 0, 0], verbose=False)
    assert made[-1] is grp and np.array_equal(again.simulation_indices, got.simulation_indices)  # ... which it served
    for m in grp.members:
        del m.pushed[len(m.pushed) // 2:]  # (the second call pushed the same spans again)
    ka.clear_engine_cache()
    assert grp._pool._shutdown and not _lib._ENGINE_POOL
    # every member swept its block of every chunk: contiguous, disjoint, covering the dictionary
    spans = sorted(sp for m in grp.members for sp in m.pushed)
    assert spans[0][0] == 0 and sum(n for _, n in spans) == len(dic)
    assert all(a + n == b for (a, n), (b, _) in zip(spans, spans[1:]))
    rs, ri = ko.dictionary_indexing(exp, dic, metric=metric, keep_n=keep_n, n_per_iteration=chunk, navigation_mask=nav,
                                    signal_mask=sig)
    sel = slice(None) if nav is None else ~nav.ravel()
    ko.assert_topk_parity(np.asarray(got.scores).reshape(-1, keep_n)[sel],
                          np.asarray(got.simulation_indices).reshape(-1, keep_n)[sel], rs, ri, atol=1e-5)
    one = ka.dictionary_indexing(exp, dic, metric, keep_n, n_per_iteration=chunk, navigation_mask=nav, signal_mask=sig,
                                 device=0, verbose=False)
    assert isinstance(made[-1], StandInContext)
    # (the stand-in's BLAS rounds differently for differently shaped chunks; the ENGINE's group result is bit-identical
    # to its single-device result: tests/test_gpu_group.py)
    assert np.allclose(one.scores, got.scores, atol=1e-6) and np.array_equal(one.simulation_indices, got.simulation_indices)


def test_ebsd_keeps_its_group(monkeypatch):
    import kikuchipy_amd as ka
    from kikuchipy_amd import _lib

    made = []
    monkeypatch.setattr(_lib, "make_engine", lambda device=0, devices=None, gather=None: made.append(StandInGroup(devices)) or made[-1])
    rng = np.random.default_rng(4)
    s = ka.EBSD(rng.integers(0, 256, (2, 3, 12, 10), dtype=np.uint8), devices=[0, 0])
    d = ka.EBSD(rng.random((300, 12, 10), dtype=np.float32), xmap=ka.signals.DictionaryXmap.empty(300))
    a = s.dictionary_indexing(d, keep_n=4, verbose=False)
    b = s.dictionary_indexing(d, keep_n=4, verbose=False)
    assert len(made) == 1 and np.array_equal(a.simulation_indices, b.simulation_indices)
    rs, ri = ko.dictionary_indexing(s.data, d.data, keep_n=4)
    ko.assert_topk_parity(a.scores, a.simulation_indices, rs, ri, atol=1e-5)


def test_a_one_shot_call_hands_its_engine_back_or_closes_it(monkeypatch):
    """The stand-alone driver's own engine (metric given by name): a call that SUCCEEDS hands it to the engine pool for the
    next call on the same device; a call that FAILS half-way closes it at once (not when the garbage collector
    finds the metric); the progress callback fires after each chunk has been handed over, in order; `KPDI_ENGINE_CACHE=0`
    closes every engine with its call."""
    import kikuchipy_amd as ka
    from kikuchipy_amd import _lib

    class Engine(StandInContext):
        closed = 0
        fail_at = None

        def push_dictionary_chunk(self, patterns, global_start):
            if Engine.fail_at is not None and global_start >= Engine.fail_at:
                raise _lib.KpdiError("injected upload failure")
            events.append(("push", global_start))
            return super().push_dictionary_chunk(patterns, global_start)

        def close(self):
            Engine.closed += 1
            super().close()

    made, events = [], []
    monkeypatch.setattr(_lib, "make_engine", lambda device=0, devices=None, gather=None: made.append(Engine(device)) or made[-1])
    rng = np.random.default_rng(9)
    exp = rng.integers(0, 256, (6, 12, 10), dtype=np.uint8)
    dic = rng.random((90, 12, 10), dtype=np.float32)
    a = ka.dictionary_indexing(exp, dic, keep_n=4, n_per_iteration=30, device=0, verbose=False,
                               progress=lambda done, total: events.append(("progress", done, total)))
    assert events == [("push", 0), ("progress", 1, 3), ("push", 30), ("progress", 2, 3), ("push", 60), ("progress", 3, 3)]
    assert len(made) == 1 and Engine.closed == 0 and _lib._ENGINE_POOL[((0,), None)] is made[0]
    b = ka.dictionary_indexing(exp, dic, keep_n=4, device=0, verbose=False)      # the idle engine serves the next call
    assert len(made) == 1 and np.array_equal(a.simulation_indices, b.simulation_indices)
    Engine.fail_at = 30
    with pytest.raises(_lib.KpdiError, match="injected upload failure"):
        ka.dictionary_indexing(exp, dic, keep_n=4, n_per_iteration=30, device=0, verbose=False)
    assert Engine.closed == 1 and not _lib._ENGINE_POOL                           # closed by the failing call, not kept
    Engine.fail_at = None
    monkeypatch.setenv("KPDI_ENGINE_CACHE", "0")
    ka.dictionary_indexing(exp, dic, keep_n=4, device=0, verbose=False)
    assert len(made) == 2 and Engine.closed == 2 and not _lib._ENGINE_POOL
