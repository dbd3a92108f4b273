"""Worker of tests/test_distributed_gloo.py: launched by torch.distributed.run
with 2 ranks on CPU (gloo).  Exercises the N>1 host path: rendezvous, unique-id
exchange, dictionary sharding, and that merging per-shard best-k lists with
the (score desc, index asc) rule reproduces the global result.  The GPU data
path (RCCL all-gather + merge kernel) implements the same merge; here the CPU
oracle stands in for the per-shard engine because there is no GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from kikuchipy_amd.parallel import Communicator, init_process_group, shard_range  # noqa: E402
from oracle import kpdi_oracle as ko  # noqa: E402

dist = init_process_group("gloo")
comm = Communicator.from_env()
assert comm.world_size == dist.get_world_size() == 2 and comm.rank == dist.get_rank()

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
gathered = [None, None]
dist.all_gather_object(gathered, (s_loc, i_loc))
scores = np.full((len(exp), k), -np.inf, dtype=np.float32)
idx = np.full((len(exp), k), np.iinfo(np.int64).max, dtype=np.int64)
for s_r, i_r in gathered:
    scores, idx = ko.merge_topk(scores, idx, s_r, i_r, k)
s_ref, i_ref = ko.dictionary_indexing(exp, dic, metric="ncc", keep_n=k)
assert np.array_equal(idx, i_ref), (idx[:2], i_ref[:2])
assert np.allclose(scores, s_ref, atol=1e-6)
comm.barrier()

# 3. refinement sharded over the map's patterns: every rank solves its block, the rows are
# gathered over the control plane.  A stand-in context (the oracle's objective + SciPy, i.e. the
# reference's own solver) replaces the GPU engine, which is absent here.
import kikuchipy_amd as ka  # noqa: E402
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
if comm.rank == 0:
    print("GLOO_WORKER_OK")
dist.destroy_process_group()
