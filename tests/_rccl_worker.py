"""Worker of tests/test_gpu_multigpu.py: launched with one rank per GPU (RANK / WORLD_SIZE / MASTER_* in the environment).
The REAL N > 1 path: every rank matches its dictionary shard on its own MI355X,
`kpdi_finalize` all-gathers the per-rank best-k lists with RCCL over xGMI and merges them; the
result of every rank must equal the oracle's single-process result.  Called twice (a new engine
context per call) and once with a navigation + signal mask and `ndp`."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import kikuchipy_amd as ka  # noqa: E402
from kikuchipy_amd import _lib  # noqa: E402
from kikuchipy_amd.parallel import Communicator  # noqa: E402
from oracle import kpdi_oracle as ko  # noqa: E402

comm = Communicator.from_env()  # TCP rendezvous on MASTER_ADDR:MASTER_PORT (no torch)
device = int(os.environ.get("LOCAL_RANK", "0"))
if os.environ.get("KPDI_BENCH_SHARE_GPU"):  # more ranks than GPUs (only to see what RCCL says about it)
    device %= max(_lib.device_count(), 1)
rng = np.random.default_rng(11)
exp = rng.integers(0, 256, (6, 50, 60, 60)).astype(np.uint8)
dic = rng.random((5003, 60, 60)).astype(np.float32)
dic[4000] = dic[7]  # a tie across shard boundaries
nav = rng.random((6, 50)) < 0.2
sig = ~ko.circular_window((60, 60)).astype(bool)
for kw in (dict(metric="ncc", n_per_iteration=1300), dict(metric="ncc", n_per_iteration=1300),
           dict(metric="ndp", navigation_mask=nav, signal_mask=sig)):
    res = ka.dictionary_indexing(exp, dic, keep_n=20, comm=comm, device=device, verbose=False, **kw)
    rs, ri = ko.dictionary_indexing(exp, dic, metric=kw["metric"], keep_n=20, navigation_mask=kw.get("navigation_mask"),
                                    signal_mask=kw.get("signal_mask"))
    s, i = res.scores, res.simulation_indices
    if "navigation_mask" in kw:
        s, i = s[~nav.ravel()], i[~nav.ravel()]
    ko.assert_topk_parity(s, i, rs, ri, atol=1e-5)
    # every rank holds the identical result
    box = comm.all_gather((res.scores, res.simulation_indices))
    assert all(np.array_equal(b[0], box[0][0]) and np.array_equal(b[1], box[0][1]) for b in box)
comm.barrier()
expect = os.environ.get("KPDI_TEST_EXPECT_GATHER")
if expect:
    assert comm.gather == expect, (comm.gather, comm.gather_reason)
if comm.rank == 0:
    print("RCCL_WORKER_OK", comm.world_size, "gather", comm.gather, "|", comm.gather_reason)
comm.close()
