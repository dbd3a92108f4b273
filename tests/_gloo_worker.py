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
if comm.rank == 0:
    print("GLOO_WORKER_OK")
dist.destroy_process_group()
