"""The fallback chain of the multi-GPU exchange step, rehearsed on CPU with faults injected (world size 2).

One process per GPU gathers the per-rank best-k lists with an RCCL all-gather inside `kpdi_finalize`
(the exchange step that replaces the host merge of indexing/_dictionary_indexing.py:120-128 across dictionary shards).
That path has never met two GPUs (no multi-GPU box in any round), so it must not be a single point of failure:
`Communicator.attach` falls back to a HOST-STAGED gather over the TCP control plane when a rank cannot create its
communicator or the first all-gather does not complete, and `bench.py` retries as ONE process over all GPUs when the
multi-process run fails altogether.  The engine is tests/_standin_engine.py (faults: $KPDI_TEST_COMM_FAULT); the GPU
counterpart of the host-staged gather is tests/test_gpu_multigpu.py."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from test_distributed_gloo import free_port, launch_plain

WORKER = os.path.join(ROOT, "tests", "_bench_worker.py")
FAULTS = [("init_error:1", "kpdi_comm_init failed on rank 1"), ("init_error:0", "kpdi_comm_init failed on rank 0"),
          ("init_hang:1", "kpdi_comm_init failed on rank"), ("collective_hang:1", "first all-gather failed on rank")]


@pytest.mark.parametrize("fault,why", FAULTS + [(None, "KPDI_GATHER=host")])
def test_dictionary_indexing_agrees_on_the_host_staged_gather(fault, why):
    env = {"KPDI_COMM_TIMEOUT": "2"}
    if fault:
        env["KPDI_TEST_COMM_FAULT"] = fault
    else:
        env["KPDI_GATHER"] = "host"
    out = launch_plain(os.path.join(ROOT, "tests", "_fallback_worker.py"), 2, env)
    assert "FALLBACK_WORKER_OK gather=host" in out and why in out, out


def test_no_fault_keeps_the_collective():
    out = launch_plain(os.path.join(ROOT, "tests", "_fallback_worker.py"), 2, {"KPDI_TEST_EXPECT_GATHER": "rccl"})
    assert "FALLBACK_WORKER_OK gather=rccl" in out


def run_bench(cmd, extra_env):
    env = dict(os.environ, OMP_NUM_THREADS="2", KPDI_COMM_TIMEOUT="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout  # the contract: ONE JSON line on stdout
    return json.loads(lines[0]), p.stderr


BENCH_ARGS = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--check-rows", "8", "--no-cpu-baseline"]


@pytest.mark.parametrize("fault,why", [FAULTS[0], FAULTS[3]])
def test_bench_line_survives_a_broken_collective(fault, why):
    """`python bench.py --gpus 2` with the communicator failing on rank 1 / its first all-gather hanging: still the
    verified line (rank 0 checks the MERGED result against the C oracle), and it says which gather ran and why."""
    out, _ = run_bench([sys.executable, WORKER] + BENCH_ARGS, {"KPDI_TEST_COMM_FAULT": fault})
    mg = out["multi_gpu"]
    assert out["n_gpus"] == 2 and out["check"]["rows"] == 8 and mg["processes"] == 2
    assert mg["gather"].startswith("host-staged") and why in mg["gather_fallback_reason"]
    assert mg["rccl_ranks"] == 0 and mg["lists_merged"] == 2 and mg["identical_result_on_every_rank"]
    assert "host-staged" in out["config"]["parallelism"]
    shards = [p["shard"] for p in mg["per_rank"]]
    assert shards[0][0] == 0 and shards[0][1] == shards[1][0] and shards[1][1] == out["config"]["dictionary_patterns"]


def test_bench_under_the_launcher_with_a_hung_collective():
    """The driver's form (`python -m torch.distributed.run ... bench.py --gpus 2`)."""
    out, _ = run_bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(free_port()), WORKER] + BENCH_ARGS + ["--workload", "config3"],
                       {"KPDI_TEST_COMM_FAULT": "collective_hang:0"})
    assert out["multi_gpu"]["gather"].startswith("host-staged") and out["check"]["rows"] == 8


def test_bench_retries_as_one_process_when_a_rank_dies():
    """Last step of the chain: rank 1 exits before it has a context - the spawner runs the job again as ONE process over
    both GPUs (a kpdi_group; stand-ins here) and the line says so."""
    out, err = run_bench([sys.executable, WORKER] + BENCH_ARGS, {"KPDI_BENCH_FAIL_RANK": "1"})
    mg = out["multi_gpu"]
    assert out["n_gpus"] == 2 and mg["processes"] == 1 and mg["lists_merged"] == 2 and out["check"]["rows"] == 8
    assert "rank 1 exited with code 3" in mg["gather_fallback_reason"] and "retrying as ONE process" in err


def test_bench_under_the_launcher_retries_when_a_rank_raises():
    """Under a launcher a rank that fails with an exception steps aside (exit code 0) and rank 0 runs the single-process
    form itself."""
    out, err = run_bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(free_port()), WORKER] + BENCH_ARGS,
                         {"KPDI_BENCH_RAISE_RANK": "1"})
    mg = out["multi_gpu"]
    assert mg["processes"] == 1 and mg["lists_merged"] == 2 and out["check"]["rows"] == 8
    assert "one process per GPU failed on rank" in mg["gather_fallback_reason"]


def test_no_fallback_switch_keeps_the_failure():
    env = dict(os.environ, KPDI_BENCH_FAIL_RANK="1", KPDI_BENCH_NO_FALLBACK="1")
    p = subprocess.run([sys.executable, WORKER] + BENCH_ARGS, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and not p.stdout.strip()
