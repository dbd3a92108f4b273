"""Worker of tests/test_comm_fallback.py: `kikuchipy_amd.dictionary_indexing(..., comm=)` under 2 ranks on CPU with a
fault injected into the stand-in engine's "RCCL" ($KPDI_TEST_COMM_FAULT, tests/_standin_engine.py) - the ranks must
agree on the host-staged gather over the TCP control plane and every rank must still end with the global result."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import kikuchipy_amd as ka  # noqa: E402
from _standin_engine import StandInContext  # noqa: E402
from kikuchipy_amd.indexing.similarity_metrics import NormalizedCrossCorrelationMetric  # noqa: E402
from kikuchipy_amd.parallel import Communicator  # noqa: E402
from oracle import kpdi_oracle as ko  # noqa: E402

comm = Communicator.from_env()
rng = np.random.default_rng(11)
exp = rng.integers(0, 256, (3, 7, 12, 12)).astype(np.uint8)
dic = rng.random((401, 12, 12)).astype(np.float32)
dic[300] = dic[7]  # a tie across the shard boundary
want_s, want_i = ko.dictionary_indexing(exp, dic, metric="ncc", keep_n=6)
for call in range(2):  # a NEW context each call: the negotiation runs again, with the same outcome
    fake = StandInContext()
    res = ka.dictionary_indexing(exp, dic, metric=NormalizedCrossCorrelationMetric(context=fake), keep_n=6, n_per_iteration=97,
                                 comm=comm, verbose=False)
    assert np.array_equal(res.simulation_indices, want_i) and np.allclose(res.scores, want_s, atol=1e-6)
    expect = os.environ.get("KPDI_TEST_EXPECT_GATHER", "host")
    assert comm.gather == expect, (comm.gather, comm.gather_reason)
    if expect == "host":
        assert fake.comm is None and fake._host_gather is comm, "a rank kept its communicator: the next finalize would hang"
comm.barrier()
if comm.rank == 0:
    print(f"FALLBACK_WORKER_OK gather={comm.gather} reason={comm.gather_reason}")
comm.close()
sys.stdout.flush()
os._exit(0)  # (a "hung" bootstrap sleeps on a daemon thread: do not wait for it)
