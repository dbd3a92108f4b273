"""CPU rehearsal of `bench.py`'s multi-rank control flow (tests/test_bench_multirank.py).

Runs bench.main() itself - spawner, rendezvous, sharding, barriers, max-over-ranks timing, result
check against the C oracle, cpu_baseline, per-rank report, the ONE JSON line - with the workloads
shrunk to sizes the oracle finishes in seconds and `tests/_standin_engine.StandInContext` in place
of the GPU engine, so that the first real execution of that code is not on the driver's 8-GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench  # noqa: E402
from _standin_engine import StandInContext, StandInGroup  # noqa: E402

for name, small in (("config2", dict(m=40, n=701)), ("config3", dict(m=40, n=701)),
                    ("config4", dict(m=48, n=1001)), ("config5", dict(m=24, n=601))):
    bench.WORKLOADS[name].update(small)
bench.BLOCK = 256
os.environ["KPDI_BENCH_SCRIPT"] = os.path.abspath(__file__)  # the spawner starts THIS script per rank

if __name__ == "__main__":
    sys.exit(bench.main(sys.argv[1:], context_factory=StandInContext, group_factory=StandInGroup))
