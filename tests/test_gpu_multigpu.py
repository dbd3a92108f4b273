"""The dictionary-sharded run on REAL GPUs: one process per GPU, RCCL all-gather of the per-rank
best-k lists inside kpdi_finalize.  Needs >= 2 visible MI355X; on the 1-GPU boxes
the builder can reach it skips (the same host path runs under gloo in test_distributed_gloo.py, the
RCCL call itself with a one-rank communicator in test_gpu_engine.py::test_rccl_path_single_rank)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from test_distributed_gloo import launch_plain

pytestmark = pytest.mark.gpu


def test_sharded_dictionary_indexing_over_rccl():
    from kikuchipy_amd import _lib

    n = _lib.device_count()
    if n < 2:
        pytest.skip(f"{n} GPU visible: the multi-rank RCCL run needs at least 2")
    ranks = min(n, 8)
    out = launch_plain(os.path.join(ROOT, "tests", "_rccl_worker.py"), ranks, {"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert f"RCCL_WORKER_OK {ranks}" in out
    # (which gather ran is printed, not asserted: on a node whose RCCL cannot start the fallback is the right outcome -
    # tests/test_gpu_multigpu.py::test_bench_runs_sharded_over_rccl reports it in the line)
    print(out)


def test_two_ranks_sharing_one_gpu_fall_back_to_the_host_staged_gather():
    """Runs on ANY box: two ranks on GPU 0.  RCCL refuses a device that appears twice (ncclCommInitRank: invalid usage) -
    a real communicator failure on real contexts - so `Communicator.attach` must agree on the host-staged gather
    (kpdi_export_lists -> TCP control plane -> kpdi_import_lists -> the merge kernel) and every rank must still end
    with the oracle's global result, three calls in a row."""
    out = launch_plain(os.path.join(ROOT, "tests", "_rccl_worker.py"), 2,
                       {"KPDI_BENCH_SHARE_GPU": "1", "KPDI_TEST_EXPECT_GATHER": "host", "KPDI_COMM_TIMEOUT": "30"})
    assert "RCCL_WORKER_OK 2 gather host" in out and "kpdi_comm_init failed on rank" in out, out


def test_bench_two_ranks_sharing_one_gpu():
    """`python bench.py --gpus 2` with both ranks on GPU 0: the verified line of the host-staged gather (timings of two
    processes contending for one device mean nothing; the control flow and the merged result are what is checked)."""
    import json

    env = dict(os.environ, KPDI_BENCH_SHARE_GPU="1", KPDI_COMM_TIMEOUT="30")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    mg = out["multi_gpu"]
    assert out["n_gpus"] == 2 and out["check"]["rows"] == 64 and mg["processes"] == 2
    assert mg["gather"].startswith("host-staged") and mg["rccl_ranks"] == 0 and mg["identical_result_on_every_rank"]


def test_host_staged_gather_on_every_visible_gpu():
    """The fallback on distinct devices (>= 2 GPUs), forced with $KPDI_GATHER=host: the same worker as the RCCL run."""
    from kikuchipy_amd import _lib

    n = _lib.device_count()
    if n < 2:
        pytest.skip(f"{n} GPU visible: needs at least 2 (two ranks on ONE GPU run in the test above)")
    ranks = min(n, 8)
    out = launch_plain(os.path.join(ROOT, "tests", "_rccl_worker.py"), ranks,
                       {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "KPDI_GATHER": "host", "KPDI_TEST_EXPECT_GATHER": "host"})
    assert f"RCCL_WORKER_OK {ranks} gather host" in out


def test_bench_runs_sharded_over_rccl():
    """`python bench.py --gpus N` as the driver types it, on every visible GPU (>= 2)."""
    import json

    from kikuchipy_amd import _lib

    n = min(_lib.device_count(), 8)
    if n < 2:
        pytest.skip(f"{n} GPU visible: the multi-rank bench needs at least 2")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == n and out["check"]["rows"] == 64
    mg = out["multi_gpu"]
    assert mg["rccl_ranks"] == n or mg["gather"].startswith("host-staged"), mg


@pytest.mark.timeout(900)  # (a collective that does not complete must fail the test, not hang the suite)
@pytest.mark.parametrize("gather", ["rccl", "p2p", None])
def test_group_over_every_visible_gpu(gather):
    """ONE process, one group member per GPU: the in-process RCCL communicator (ncclCommInitAll) and the peer-copy gather
    over xGMI, each against the single-GPU result, bit for bit (the 1-GPU boxes run this with members sharing a device:
    tests/test_gpu_group.py)."""
    import numpy as np

    import kikuchipy_amd as ka
    from kikuchipy_amd import _lib

    n = min(_lib.device_count(), 8)
    if n < 2:
        pytest.skip(f"{n} GPU visible: a group over distinct devices needs at least 2")
    rng = np.random.default_rng(5)
    exp = rng.integers(0, 256, (600, 60, 60), dtype=np.uint8)
    dic = rng.random((20000, 60, 60), dtype=np.float32)
    one = ka.dictionary_indexing(exp, dic, keep_n=20, device=0, verbose=False)
    with _lib.Group(list(range(n)), gather=gather) as g:
        assert g.gather == (gather or "rccl"), g.describe()
        for chunk in (len(dic), 3000):  # one chunk, then streamed chunks: every GPU takes its part of each
            g.set_problem(60, 60, None, _lib.METRIC_NCC, 20, _lib.COMPUTE_F32)
            g.set_experimental(exp, None)
            for a in range(0, len(dic), chunk):
                g.push_dictionary_chunk(dic[a:a + chunk], a)
            s, i = g.finalize(20)
            assert np.array_equal(s, one.scores) and np.array_equal(i, one.simulation_indices)
        c = g.counters()
        assert c["gather_ranks"] == n and (c["comm_ranks"] == n) == (g.gather == "rccl")
    # the call that names no device uses all of them
    res = ka.dictionary_indexing(exp, dic, keep_n=20, verbose=False)
    assert np.array_equal(res.scores, one.scores) and np.array_equal(res.simulation_indices, one.simulation_indices)


@pytest.mark.timeout(1200)
def test_bench_single_process_over_every_visible_gpu():
    """`python bench.py --gpus N --single-process`: the sharded benchmark driven from one interpreter."""
    import json

    from kikuchipy_amd import _lib

    n = min(_lib.device_count(), 8)
    if n < 2:
        pytest.skip(f"{n} GPU visible: needs at least 2")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--single-process", "--steps", "3",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == n and out["multi_gpu"]["processes"] == 1 and out["multi_gpu"]["lists_merged"] == n
    assert out["multi_gpu"]["rccl_ranks"] == n and out["check"]["rows"] == 64
