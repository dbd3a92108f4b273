#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06i; rm -rf $O; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_bench_line.py > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
timeout 300 python tools/tile_ramp_probe.py wide > $O/ramp_wide.txt 2>&1; tail -4 $O/ramp_wide.txt
timeout 300 python tools/rank_share_probe.py $O/auto.json --no-whole-tiles --ranks 1,2,4,8 --pipeline > $O/auto.log 2>&1
python - <<PY
import json
d = json.load(open("$O/auto.json"))
print({r: (v["kernel"][:9], v["ms_per_step"], v["match_ms"], v.get("step_over_even_share")) for r, v in d["ranks"].items()})
PY
