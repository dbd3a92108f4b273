#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06j; rm -rf $O; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_bench_line.py > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
timeout 600 python tools/stress.py 150 > $O/stress.log 2>&1; echo "stress rc=$?"; tail -1 $O/stress.log
timeout 300 python tools/rank_share_probe.py $O/auto.json --no-whole-tiles --ranks 1,2,4,8 --pipeline > $O/auto.log 2>&1
python - <<PY
import json
d = json.load(open("$O/auto.json"))
print({r: (v["kernel"][:9], v["ms_per_step"], v["match_ms"], v["merge_ms"], v.get("step_over_even_share")) for r, v in d["ranks"].items()})
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/tools/rank_share_probe.py --no-whole-tiles --ranks 1,8 --reps 10 --pipeline > /dev/null 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
