"""Worker of tests/test_distributed_gloo.py: 2 ranks on CPU, once over torch.distributed (gloo,
launched by torch.distributed.run - the optional transport) and once over the product's own TCP
control plane (KPDI_TEST_TRANSPORT=socket, launched as two plain processes).  Exercises the N>1 host path: rendezvous, unique-id
exchange, dictionary sharding, and that merging per-shard best-k lists with
the (score desc, index asc) rule reproduces the global result.  The GPU data
path (RCCL all-gather + merge kernel) implements the same merge; here the CPU
oracle stands in for the per-shard engine because there is no GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from kikuchipy_amd.parallel import Communicator, shard_range  # noqa: E402
from oracle import kpdi_oracle as ko  # noqa: E402

TRANSPORT = os.environ.get("KPDI_TEST_TRANSPORT", "gloo")
if TRANSPORT == "gloo":
    # the optional transport: torch.distributed (gloo) handed to Communicator as three callables
    import torch.distributed as dist  # (test only: the product's control plane is torch-free)

    dist.init_process_group(backend="gloo")

    def _bcast(payload, src):
        box = [payload]
        dist.broadcast_object_list(box, src=src)
        return box[0]

    def _gather(obj):
        box = [None] * dist.get_world_size()
        dist.all_gather_object(box, obj)
        return box

    comm = Communicator(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), broadcast_bytes=_bcast,
                        barrier=dist.barrier, all_gather=_gather)
    assert comm.group is None, "callables were given: no socket rendezvous must be attempted"
    assert comm.world_size == dist.get_world_size() == 2 and comm.rank == dist.get_rank()
else:
    # the default transport: kikuchipy_amd.parallel.SocketGroup (plain TCP), no torch in the process
    comm = Communicator.from_env()
    assert comm.group is not None and comm.world_size == 2
    assert "torch" not in sys.modules

# 1. unique-id exchange: rank 0's payload reaches everybody
uid = comm.exchange_unique_id(lambda: bytes(range(128)))
assert uid == bytes(range(128))
comm.barrier()

# 2. sharded sweep == global sweep
rng = np.random.default_rng(11)
exp = rng.integers(0, 256, (21, 12, 12)).astype(np.uint8)
dic = rng.random((401, 12, 12)).astype(np.float32)
dic[300] = dic[7]  # a tie across the shard boundary
k = 6
lo, hi = shard_range(len(dic), comm.rank, comm.world_size)
s_loc, i_loc = ko.dictionary_indexing(exp, dic[lo:hi], metric="ncc", keep_n=k, n_per_iteration=97)
i_loc = i_loc + lo
gathered = comm.all_gather((s_loc, i_loc))
scores = np.full((len(exp), k), -np.inf, dtype=np.float32)
idx = np.full((len(exp), k), np.iinfo(np.int64).max, dtype=np.int64)
for s_r, i_r in gathered:
    scores, idx = ko.merge_topk(scores, idx, s_r, i_r, k)
s_ref, i_ref = ko.dictionary_indexing(exp, dic, metric="ncc", keep_n=k)
assert np.array_equal(idx, i_ref), (idx[:2], i_ref[:2])
assert np.allclose(scores, s_ref, atol=1e-6)
comm.barrier()

# 2b. the REAL host path of a sharded run: kikuchipy_amd.dictionary_indexing(..., comm=comm) with its
# shard / chunk intersection, Communicator.attach and the finalize ordering, run under 2 ranks.  The
# engine behind the `_lib.Context` interface is a stand-in (no GPU here): the oracle computes every
# pushed chunk, and `finalize` does what kpdi_finalize does with RCCL - all-gather of the per-rank
# best-k lists + the (score desc, index asc) merge - over gloo.  Like the real context it only
# gathers when a communicator was attached: a context that missed `comm_init` returns its own
# shard's lists and the comparison below fails (the id()-reuse bug of round 1).
import gc  # noqa: E402

import kikuchipy_amd as ka  # noqa: E402
from kikuchipy_amd.indexing.similarity_metrics import NormalizedCrossCorrelationMetric, NormalizedDotProductMetric  # noqa: E402


sys.path.insert(0, os.path.join(ROOT, "tests"))
from _standin_engine import StandInContext as OracleEngineContext  # noqa: E402

exp4 = exp.reshape(3, 7, 12, 12)
nav = np.zeros((3, 7), dtype=bool)
nav[1, 2:5] = True
sig = np.zeros((12, 12), dtype=bool)
sig[:2] = True
for call, (Metric, name, kw) in enumerate([
        (NormalizedCrossCorrelationMetric, "ncc", dict(n_per_iteration=97)),            # chunks straddle the shard boundary
        (NormalizedCrossCorrelationMetric, "ncc", dict(n_per_iteration=97)),            # same again: a NEW context each call
        (NormalizedDotProductMetric, "ndp", dict(n_per_iteration=None, navigation_mask=nav, signal_mask=sig)),
        (NormalizedCrossCorrelationMetric, "ncc", dict(n_per_iteration=250)),
]):
    fake = OracleEngineContext()
    res = ka.dictionary_indexing(exp4, dic, metric=Metric(context=fake), keep_n=k, comm=comm, verbose=False, **kw)
    assert fake.comm == (comm.rank, 2), "Communicator.attach did not reach the new context"
    # this rank pushed exactly its shard, cut at the reference's chunk boundaries
    assert sum(n for _, n in fake.pushed) == hi - lo and min(a for a, _ in fake.pushed) == lo
    assert max(a + n for a, n in fake.pushed) == hi
    per = kw["n_per_iteration"] or len(dic)
    assert all(a // per == (a + n - 1) // per for a, n in fake.pushed)
    s_one, i_one = ko.dictionary_indexing(exp4, dic, metric=name, keep_n=k, navigation_mask=kw.get("navigation_mask"),
                                          signal_mask=kw.get("signal_mask"))
    if kw.get("navigation_mask") is not None:
        s_one, i_one, _ = ko.scatter_navigation_mask(s_one, i_one, kw["navigation_mask"], k)
    assert np.array_equal(res.simulation_indices, i_one), (call, res.simulation_indices[:2], i_one[:2])
    assert np.allclose(res.scores, s_one, atol=1e-6)
    del fake, res
    gc.collect()  # the next context may now get this one's id()
assert OracleEngineContext.live == 0
comm.barrier()

# 3. refinement sharded over the map's patterns: every rank solves its block, the rows are
# gathered over the control plane.  A stand-in context (the oracle's objective + SciPy, i.e. the
# reference's own solver) replaces the GPU engine, which is absent here.
from kikuchipy_amd.indexing._refinement import refine, rotation_from_euler  # noqa: E402


class OracleContext:
    calls = 0

    def set_master_pattern(self, up, lo):
        self.mp = (up, lo)

    def refine_set_patterns(self, pats, signal_mask, rescale, om):
        self.pats, self.rescale, self.om = pats, rescale, om

    def refine_solve(self, mode, x0, fixed, lower, upper, xatol, fatol, maxiter, maxfev):
        OracleContext.calls += 1
        out = np.zeros((x0.shape[0], 1, 6))
        for i in range(x0.shape[0]):
            dc = ko.direction_cosines_fixed_pc(ko.gnomonic_bounds((20, 20), fixed[i, 0]), fixed[i, 0, 2], 20, 20, self.om)
            r = ko.refine_solver(self.pats[i].ravel(), "ori", x0[i, 0], *self.mp, self.rescale, direction_cosines=dc,
                                 method_kwargs=dict(options=dict(maxfev=25)))
            out[i, 0] = [1 - r[0], r[1], 0, *r[2:5]]
        return out

    def close(self):
        pass


rng = np.random.default_rng(5)
mpd = rng.random((41, 41)).astype(np.float32)
det = ka.EBSDDetector(shape=(20, 20), pc=(0.45, 0.6, 0.5))
eu = np.column_stack([rng.uniform(0.3, 6, 5), rng.uniform(0.3, 2.8, 5), rng.uniform(0.3, 6, 5)])
dc = ko.detector_direction_cosines((20, 20), (0.45, 0.6, 0.5))
sim = ko.project_patterns(rotation_from_euler(eu), dc, mpd, mpd)
pats = (sim * 255).astype(np.uint8).reshape(5, 20, 20)
rot0 = rotation_from_euler(eu + 0.01)
mp = ka.EBSDMasterPattern(mpd)
res, _ = refine("ori", pats, rot0, det, mp, context=OracleContext(), comm=comm, verbose=False)
one, _ = refine("ori", pats, rot0, det, mp, context=OracleContext(), verbose=False)
assert res.scores.shape == (5,) and np.array_equal(res.scores, one.scores) and np.array_equal(res.euler, one.euler)
assert np.array_equal(res.num_evals, one.num_evals)
lo5, hi5 = shard_range(5, comm.rank, 2)
assert (lo5, hi5) == ((0, 3) if comm.rank == 0 else (3, 5))
comm.barrier()
assert comm.all_reduce_max(1.5 + comm.rank) == 2.5
if TRANSPORT != "gloo":
    assert "torch" not in sys.modules, "the product pulled torch in"
if comm.rank == 0:
    print("GLOO_WORKER_OK" if TRANSPORT == "gloo" else "SOCKET_WORKER_OK")
comm.close()
if TRANSPORT == "gloo":
    dist.destroy_process_group()
