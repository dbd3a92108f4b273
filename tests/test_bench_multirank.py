"""`bench.py --gpus N` under two ranks on CPU: the driver's command shapes, rehearsed.

The engine is the stand-in of tests/_standin_engine.py (no GPU here); everything else is bench.py's
own code: its spawner (`python bench.py --gpus 2` as typed), the launcher form (`python -m
torch.distributed.run ... bench.py --gpus 2`), the TCP control plane of kikuchipy_amd.parallel, the
dictionary shards, the check of the merged result against the C oracle, cpu_baseline, the JSON line."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from test_distributed_gloo import free_port

WORKER = os.path.join(ROOT, "tests", "_bench_worker.py")


def run(cmd, extra_env=None):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout  # the contract: ONE JSON line on stdout
    return json.loads(lines[0])


def check_line(out, workload_key, n_gpus=2):
    assert out["n_gpus"] == n_gpus and out["value"] > 0 and out["unit"] == "patterns/s"
    assert out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "strong"
    assert workload_key in out["config"]["workload"]
    assert out["check"]["rows"] == 8  # rank 0 checked the MERGED result against the C oracle
    mg = out["multi_gpu"]
    assert mg["rccl_ranks"] == n_gpus and mg["identical_result_on_every_rank"]
    shards = [p["shard"] for p in mg["per_rank"]]
    assert shards[0][0] == 0 and shards[-1][1] == out["config"]["dictionary_patterns"]
    assert all(a[1] == b[0] for a, b in zip(shards, shards[1:]))
    assert out["roofline"]["max_over_ranks"]["avg_launch_ms"] >= out["roofline"]["avg_launch_ms"] - 1e-9


@pytest.mark.parametrize("workload,key,extra", [
    ("config2", "configs[1]", ["--cpu-sample", "200"]),
    ("config4", "configs[3]", ["--no-cpu-baseline"]),
    ("config5", "configs[4]", ["--no-cpu-baseline", "--compute", "f16"]),
])
def test_bench_spawns_its_own_ranks(workload, key, extra):
    out = run([sys.executable, WORKER, "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", workload,
               "--check-rows", "8"] + extra)
    check_line(out, key)
    if workload == "config2":
        cb = out["cpu_baseline"]
        assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port"
    else:
        assert out["check"]["planted_found"] == 16


@pytest.mark.parametrize("workload,key,extra", [
    ("config2", "configs[1]", []), ("config3", "configs[2]", []), ("config4", "configs[3]", ["--no-pipeline"]),
])
def test_bench_single_process_over_a_group(workload, key, extra):
    """`python bench.py --gpus 3 --single-process`: ONE process, a group of engine contexts (stand-ins here), member i on
    dictionary block i, one merged and oracle-checked result, the same JSON line."""
    out = run([sys.executable, WORKER, "--gpus", "3", "--single-process", "--steps", "2", "--warmup", "1", "--workload",
               workload, "--check-rows", "8", "--no-cpu-baseline"] + extra)
    assert out["n_gpus"] == 3 and out["value"] > 0 and key in out["config"]["workload"]
    assert "ONE process" in out["config"]["parallelism"] and out["check"]["rows"] == 8
    mg = out["multi_gpu"]
    assert mg["processes"] == 1 and mg["lists_merged"] == 3 and mg["gather"].startswith("p2p") and mg["rccl_ranks"] == 0
    shards = [p["shard"] for p in mg["per_rank"]]
    assert len(shards) == 3 and shards[0][0] == 0 and shards[-1][1] == out["config"]["dictionary_patterns"]
    assert all(a[1] == b[0] for a, b in zip(shards, shards[1:]))


def test_bench_under_the_launcher():
    """The driver's form.  Under torch.distributed.run MASTER_PORT belongs to the launcher's own
    store: the control plane must find its own port next to it."""
    out = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(free_port()), WORKER, "--gpus", "2", "--steps", "2", "--warmup", "1",
               "--workload", "config3", "--check-rows", "8", "--no-cpu-baseline"])
    check_line(out, "configs[2]")


def test_bench_with_eight_ranks_under_the_launcher():
    """The driver's 8-GPU command, word for word, on CPU (stand-in engines): rank 0's ONE line carries what the scaling
    record is judged on - `multi_gpu.rccl_ranks == 8` (or a named fallback), every rank's shard, the identical result on
    every rank, `roofline.max_over_ranks` - and the shards are the eight contiguous blocks of the dictionary."""
    out = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
               "127.0.0.1", "--master-port", str(free_port()), WORKER, "--gpus", "8", "--steps", "2", "--warmup", "1",
               "--check-rows", "8", "--no-cpu-baseline"], {"OMP_NUM_THREADS": "1"})
    check_line(out, "configs[1]", n_gpus=8)
    mg = out["multi_gpu"]
    assert mg["processes"] == 8 and len(mg["per_rank"]) == 8 and [p["rank"] for p in mg["per_rank"]] == list(range(8))
    assert mg["rccl_ranks"] == 8 or mg["gather_fallback_reason"]
    sizes = [p["shard"][1] - p["shard"][0] for p in mg["per_rank"]]
    assert max(sizes) - min(sizes) <= 1 and sum(sizes) == out["config"]["dictionary_patterns"]
    mo = out["roofline"]["max_over_ranks"]
    assert 0 <= mo["rank"] < 8 and mo["avg_launch_ms"] > 0
    assert "dictionary sharded over 8 GPU(s)" in out["config"]["parallelism"] and out["vs_baseline"] is None


def test_bench_imports_no_torch():
    text = open(os.path.join(ROOT, "bench.py")).read()
    assert "import torch" not in text and "from torch" not in text


@pytest.mark.parametrize("resident_prefix", [True, False])
def test_rank_share_leg_of_the_default_run(monkeypatch, resident_prefix):
    """bench.py's informational legs for the 8-GPU configurations (one rank's share on one GPU), on a toy
    workload with the stand-in engine: the shard is rank 0's block, the result is checked against the C oracle
    over that shard (a prefix of a resident dictionary, or generated block + pixel-rotated copies)."""
    import numpy as np

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench
    from _standin_engine import StandInContext
    from kikuchipy_amd import _lib
    from kikuchipy_amd.parallel import shard_range

    monkeypatch.setitem(bench.WORKLOADS, "toy", dict(m=96, n=1000, sy=6, sx=5, metric="ndp", keep_n=7, mask=False,
                                                     preprocess=False, name="toy"))
    dic = np.random.default_rng(3).random((400, 6, 5), dtype=np.float32)
    made = []

    def make(device):
        made.append(StandInContext(device))
        return made[-1]

    d_dic = None
    if resident_prefix:
        holder = StandInContext(0)
        d_dic = holder.dev_alloc(dic.nbytes)
        holder.h2d(d_dic, dic)
        # (the stand-in's device pointers are process-wide handles: the leg's own context reads the holder's buffer)
        make = lambda device: holder  # noqa: E731
    rec = bench.rank_share_leg(_lib, make, 0, "toy", 8, d_dic, dic, 2, 12, shard_range, per=50)
    assert rec["shard_patterns"] == 125 and rec["check"]["rows"] == 12 and rec["check"]["index_agreement"] == 1.0
    assert rec["check"]["max_abs_score_diff"] < 1e-5 and rec["ms_per_step"] > 0
