"""A stand-in for `kikuchipy_amd._lib.Context` on machines without a GPU (test infrastructure).

It lets the REAL multi-rank host code - `kikuchipy_amd.dictionary_indexing(..., comm=)`, `bench.py`'s
main() with its spawner, barriers, max-over-ranks timing, result check and JSON line - run under
two ranks on CPU.  The oracle computes every pushed chunk; `finalize` does what kpdi_finalize does
with RCCL - all-gather of the per-rank best-k lists + the (score desc, index asc) merge - over the
communicator's control plane.  Like the real context it only gathers when `comm_init` reached it:
a context that missed it returns its own shard's lists and every comparison with the global result
fails (the id()-reuse bug of round 1).  "Device memory" is a dict of byte buffers addressed by
integers, so pointer arithmetic on the host side (`d + offset`) behaves as with HBM pointers."""
import time

import numpy as np

from oracle import kpdi_oracle as ko

DTYPES = {0: np.uint8, 1: np.uint16, 2: np.float32, 3: np.float64, 8: np.float16}


class StandInContext:
    live = 0
    _next_base = 1 << 40

    def __init__(self, device=0):
        self.device = device
        self.comm = None
        self._comm = None  # set by Communicator.attach, as on the real context
        self._host_gather = None  # ... when the ranks gather over the control plane (no usable "RCCL")
        self._imported = None
        self.pushed = []
        self.mem = {}
        self.scores = None
        self.pre = []
        self.t = dict(match_ms=0.0, match_launches=0, match_flops=0.0, prep_ms=0.0, merge_ms=0.0, comm_ms=0.0,
                      fixed_ms=0.0, preproc_ms=0.0, preproc_launches=0)
        StandInContext.live += 1

    def __del__(self):
        StandInContext.live -= 1

    # -- problem / patterns
    def set_problem(self, sy, sx, signal_mask, metric, keep_n, compute=0):
        self.sig, self.metric, self.keep_n = (sy, sx), {0: "ncc", 1: "ndp"}[metric], keep_n
        self.mask = None if signal_mask is None else np.asarray(signal_mask, dtype=bool).reshape(sy, sx)
        self.scores = None

    def set_keep_n(self, keep_n):
        self.keep_n = keep_n
        self.scores = None

    def set_dictionary_size(self, n_total):
        pass

    def set_experimental(self, patterns, navigation_mask=None):
        nav = None if navigation_mask is None else np.asarray(navigation_mask, dtype=bool).ravel()
        self.exp = patterns if nav is None else patterns[~nav]
        self.scores = None
        self.pre = []

    def set_experimental_dev(self, d_ptr, dtype, m_all, navigation_mask=None):
        sy, sx = self.sig
        self.set_experimental(self._read(d_ptr, dtype, (m_all, sy, sx)), navigation_mask)

    @property
    def n_experimental(self):
        return len(self.exp)

    def remove_static_background(self, bg, operation=0, scale_bg=False):
        self.pre.append(lambda e: ko.remove_static_background(e, np.asarray(bg).astype(e.dtype), "subtract" if operation == 0 else "divide",
                                                              bool(scale_bg)))

    def remove_dynamic_background(self, operation=0, filter_domain=0, std=0.0, truncate=4.0):
        self.pre.append(lambda e: ko.remove_dynamic_background(e, "subtract" if operation == 0 else "divide",
                                                               "frequency" if filter_domain == 0 else "spatial",
                                                               std or None, truncate))

    # -- sweep
    def push_dictionary_chunk(self, patterns, global_start):
        if self.pre:
            for f in self.pre:
                self.exp = f(self.exp)
            self.pre = []
        self.pushed.append((global_start, len(patterns)))
        k = min(self.keep_n, len(patterns))
        t0 = time.perf_counter()
        s, i = ko.dictionary_indexing(self.exp, patterns, metric=self.metric, keep_n=k, signal_mask=self.mask)
        self.t["match_ms"] += (time.perf_counter() - t0) * 1e3
        self.t["match_launches"] += 1
        kept = int(np.prod(self.sig)) if self.mask is None else int(np.count_nonzero(~self.mask))
        self.t["match_flops"] += 2.0 * len(self.exp) * len(patterns) * kept
        if self.scores is None:
            self.scores = np.full((len(self.exp), self.keep_n), -np.inf, dtype=np.float32)
            self.idx = np.full((len(self.exp), self.keep_n), np.iinfo(np.int64).max, dtype=np.int64)
        self.scores, self.idx = ko.merge_topk(self.scores, self.idx, s, i + global_start, self.keep_n)

    def push_dictionary_chunk_dev(self, d_ptr, dtype, n_chunk, global_start):
        sy, sx = self.sig
        self.push_dictionary_chunk(self._read(d_ptr, dtype, (n_chunk, sy, sx)), global_start)

    def finalize(self, keep_n=None):
        assert keep_n is None or keep_n == self.keep_n
        if self.scores is None:  # a rank that pushed nothing contributes empty lists
            self.scores = np.full((len(self.exp), self.keep_n), -np.inf, dtype=np.float32)
            self.idx = np.full((len(self.exp), self.keep_n), np.iinfo(np.int64).max, dtype=np.int64)
        if self._host_gather is not None:  # as _lib.Context.finalize: export -> control plane -> import -> merge
            self._host_gather.gather_lists(self)
        if self._imported is not None:
            box, self._imported = self._imported, None
        elif self.comm is None:
            return self.scores, self.idx
        else:
            t0 = time.perf_counter()
            box = self._comm.all_gather((self.scores, self.idx))
            self.t["comm_ms"] += (time.perf_counter() - t0) * 1e3
        s = np.full_like(self.scores, -np.inf)
        i = np.full_like(self.idx, np.iinfo(np.int64).max)
        for s_r, i_r in box:
            s, i = ko.merge_topk(s, i, np.asarray(s_r), np.asarray(i_r), self.keep_n)
        return s, i

    def finalize_async(self, keep_n=None):
        self._pending = getattr(self, "_pending", {})
        assert len(self._pending) < 2, "two results are already pending (as the engine: two result slots)"
        ticket = self._tickets = getattr(self, "_tickets", 0) + 1
        self._pending[ticket] = self.finalize(keep_n)  # (the stand-in has nothing to overlap: it finishes here)
        return ticket

    def finalize_wait(self, ticket):
        return self._pending.pop(ticket)

    # -- multi-rank
    @staticmethod
    def comm_unique_id():
        return bytes(range(128))

    # $KPDI_TEST_COMM_FAULT = "<what>:<rank>" injects what a broken fabric does on the GPU box (tests/test_comm_fallback.py):
    # init_error - kpdi_comm_init fails at once on that rank; init_hang - it never returns there (the other ranks then
    # hang in their own bootstrap, as ncclCommInitRank does); collective_hang - the communicator comes up everywhere but
    # that rank's first all-gather never completes
    @staticmethod
    def _fault(what, rank):
        import os

        spec = os.environ.get("KPDI_TEST_COMM_FAULT", "")
        return spec == f"{what}:{rank}"

    def comm_init(self, rank, nranks, uid):
        from kikuchipy_amd import _lib

        assert uid == bytes(range(128)), "the unique id did not travel from rank 0"
        if self._fault("init_error", rank):
            raise _lib.KpdiError("libkpdi error -4: ncclCommInitRank(rank %d of %d, device 0): unhandled system error" % (rank, nranks))
        if self._fault("init_hang", rank) or any(self._fault("init_hang", r) or self._fault("init_error", r) for r in range(nranks)):
            time.sleep(3600)  # a bootstrap that waits for a rank that will never arrive
        self.comm = (rank, nranks)

    def comm_selftest(self, n_bytes=1 << 20, timeout_ms=60000):
        from kikuchipy_amd import _lib

        rank, nranks = self.comm
        if any(self._fault("collective_hang", r) for r in range(nranks)):  # one rank missing: nobody's all-gather completes
            time.sleep(timeout_ms / 1e3)
            raise _lib.KpdiError(f"libkpdi error -6: the first RCCL all-gather ({n_bytes} bytes per rank, {nranks} ranks) did not "
                                 f"complete within {timeout_ms} ms")

    def comm_drop(self):
        self.comm = None

    def export_lists(self):
        if self.scores is None:
            self.scores = np.full((len(self.exp), self.keep_n), -np.inf, dtype=np.float32)
            self.idx = np.full((len(self.exp), self.keep_n), np.iinfo(np.int64).max, dtype=np.int64)
        return self.scores, self.idx

    def import_lists(self, scores_all, indices_all):
        self._imported = list(zip(np.asarray(scores_all), np.asarray(indices_all)))

    # -- "device memory"
    def dev_alloc(self, nbytes):
        base = StandInContext._next_base
        StandInContext._next_base += (int(nbytes) + (1 << 20)) & ~0xFFF
        self.mem[base] = bytearray(int(nbytes))
        return base

    def _find(self, ptr):
        for base, buf in self.mem.items():
            if base <= ptr < base + max(len(buf), 1):
                return buf, ptr - base
        raise KeyError(f"pointer {ptr:#x} is not inside an allocation")

    def h2d(self, d_ptr, array):
        a = np.ascontiguousarray(array)
        buf, off = self._find(d_ptr)
        assert off + a.nbytes <= len(buf), "h2d beyond the allocation"
        buf[off:off + a.nbytes] = a.tobytes()

    def _read(self, d_ptr, dtype, shape):
        buf, off = self._find(d_ptr)
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        assert off + n <= len(buf), "read beyond the allocation"
        return np.frombuffer(bytes(buf[off:off + n]), dtype=dtype).reshape(shape)

    # -- measurement
    def synchronize(self):
        pass

    def set_profiling(self, on=True):
        pass

    def reset_counters(self):
        for k in self.t:
            self.t[k] = 0 if isinstance(self.t[k], int) else 0.0

    def counters(self):
        kept = int(np.prod(self.sig)) if self.mask is None else int(np.count_nonzero(~self.mask))
        return dict(self.t, k_kept=kept, kpad=kept, match_grid=0, match_nsplit=0, match_form=-1,
                    comm_ranks=self.comm[1] if self.comm else 0,
                    gather_ranks=self._host_gather.world_size if self._host_gather is not None else (self.comm[1] if self.comm else 0))

    def close(self):
        self.mem = {}


class StandInGroup:
    """Stand-in for `kikuchipy_amd._lib.Group`: N `StandInContext` members driven by one thread each, every chunk
    handed out by the library's own `kpdi_group_assign_chunk` (host code: loads without a GPU), the members' lists
    merged like the group's peer-copy gather does.  Lets the host layer's `devices=` logic run on CPU."""

    def __init__(self, devices, gather=None):
        from concurrent.futures import ThreadPoolExecutor

        self.devices = list(devices)
        self.device = self.devices[0]
        self.members = [StandInContext(d) for d in self.devices]
        self.root = self.members[0]
        self.gather = "p2p"
        self._pool = ThreadPoolExecutor(len(self.members))
        self.threads_seen = set()
        self.n_total = 0
        self.loads = [0] * len(self.members)
        self.pieces = []  # (member, global start, rows) of every piece handed out

    def __len__(self):
        return len(self.members)

    def _all(self, fn):
        import threading

        def run(im):
            self.threads_seen.add(threading.get_ident())
            return fn(*im)
        return list(self._pool.map(run, enumerate(self.members)))

    def set_problem(self, *a, **k):
        self._all(lambda i, m: m.set_problem(*a, **k))
        self.keep_n = self.root.keep_n

    def set_keep_n(self, keep_n):
        self._all(lambda i, m: m.set_keep_n(keep_n))
        self.keep_n = keep_n
        self.loads = [0] * len(self.members)

    def set_dictionary_size(self, n_total):
        self.n_total = int(n_total)
        self.loads = [0] * len(self.members)

    def set_experimental(self, patterns, navigation_mask=None):
        self._all(lambda i, m: m.set_experimental(patterns, navigation_mask))
        self.loads = [0] * len(self.members)

    @property
    def n_experimental(self):
        return self.root.n_experimental

    def push_dictionary_chunk(self, patterns, global_start):
        from kikuchipy_amd import _lib

        mine = {}
        for member, row0, rows in _lib.Group.assign_chunk(len(self.members), self.n_total, self.loads, len(patterns), 4096):
            mine[member] = (row0, rows)
            self.pieces.append((member, global_start + row0, rows))

        def push(i, m):
            if i in mine:
                a, n = mine[i]
                m.push_dictionary_chunk(patterns[a:a + n], global_start + a)
        self._all(push)

    def finalize(self, keep_n=None):
        lists = self._all(lambda i, m: m.finalize(keep_n))
        s = np.full_like(lists[0][0], -np.inf)
        i = np.full_like(lists[0][1], np.iinfo(np.int64).max)
        for s_r, i_r in lists:
            s, i = ko.merge_topk(s, i, s_r, i_r, self.keep_n)
        return s, i

    # -- what bench.py --single-process calls
    def set_experimental_dev(self, d_ptrs, dtype, m_all, navigation_mask=None):
        self._all(lambda i, m: m.set_experimental_dev(d_ptrs[i], dtype, m_all, navigation_mask))

    def push_dictionary_chunk_dev(self, d_ptrs, dtype, n_chunk, global_start):
        self._all(lambda i, m: n_chunk[i] > 0 and m.push_dictionary_chunk_dev(d_ptrs[i], dtype, n_chunk[i], global_start[i]))

    def remove_static_background(self, *a, **k):
        self._all(lambda i, m: m.remove_static_background(*a, **k))

    def remove_dynamic_background(self, *a, **k):
        self._all(lambda i, m: m.remove_dynamic_background(*a, **k))

    def finalize_async(self, keep_n=None):
        self._pending = getattr(self, "_pending", {})
        ticket = len(self._pending)
        self._pending[ticket] = self.finalize(keep_n)
        return ticket

    def finalize_wait(self, ticket):
        return self._pending.pop(ticket)

    def set_profiling(self, on=True):
        pass

    def reset_counters(self):
        self._all(lambda i, m: m.reset_counters())

    def counters(self):
        per = [m.counters() for m in self.members]
        return dict(per[0], members=per, gather=self.gather, gather_ranks=len(self.members))

    def synchronize(self):
        pass

    def close(self):
        self._pool.shutdown()
