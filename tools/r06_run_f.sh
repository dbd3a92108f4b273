#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06f; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_tail.py -m gpu -q -x > $O/pytest_tail.log 2>&1; echo "tail pytest rc=$?"; tail -12 $O/pytest_tail.log | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
KPDI_F32_WIDE=1 KPDI_TAIL_GEMM=1 timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_api.py tests/test_gpu_degenerate.py tests/test_gpu_resident.py tests/test_gpu_group.py -m gpu -q -x -k "not automatic_kernel and not chosen_by_size" > $O/pytest_forced.log 2>&1; echo "forced pytest rc=$?"; tail -3 $O/pytest_forced.log | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
KPDI_F32_WIDE=1 KPDI_TAIL_GEMM=1 timeout 300 python tools/rank_share_probe.py $O/wide_gemm.json --no-whole-tiles --ranks 1,2,4,8 --pipeline > $O/wide_gemm.log 2>&1
timeout 300 python tools/rank_share_probe.py $O/auto.json --no-whole-tiles --ranks 1,2,4,8 --pipeline > $O/auto.log 2>&1
python - <<PY
import json
for f in ("auto", "wide_gemm"):
    d = json.load(open("$O/%s.json" % f))
    print(f, {r: (v["kernel"][:9], v["ms_per_step"], v["match_ms"], v.get("step_over_even_share")) for r, v in d["ranks"].items()})
PY
cd /tmp && export TMPDIR=/tmp
KPDI_F32_WIDE=1 KPDI_TAIL_GEMM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/tools/rank_share_probe.py --no-whole-tiles --ranks 8 --reps 10 --pipeline > /dev/null 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
python $R/tools/trace_gaps.py $O/prof/p_kernel_trace.csv "prep_wave_lines_kernel<unsigned char" 12
