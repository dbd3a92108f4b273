#!/bin/bash
# round-6 check e: tailgemm.hip - correctness with the wide kernel forced + the tail kernel forced, then timings
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06e; rm -rf $O; mkdir -p $O; cd $R
KPDI_F32_WIDE=1 KPDI_TAIL_GEMM=1 timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_api.py tests/test_gpu_degenerate.py tests/test_gpu_resident.py -m gpu -q -x -k "not automatic_kernel and not chosen_by_size" > $O/pytest_forced.log 2>&1; echo "forced pytest rc=$?"; tail -6 $O/pytest_forced.log
KPDI_F32_WIDE=1 KPDI_TAIL_GEMM=1 timeout 600 python tools/stress.py 120 > $O/stress_forced.log 2>&1; echo "stress rc=$?"; tail -3 $O/stress_forced.log
echo "automatic:"; timeout 300 python tools/rank_share_probe.py $O/auto.json --no-whole-tiles --ranks 1,2,4,8 --pipeline > $O/auto.log 2>&1
echo "wide + tail gemm:"; KPDI_F32_WIDE=1 KPDI_TAIL_GEMM=1 timeout 300 python tools/rank_share_probe.py $O/wide_gemm.json --no-whole-tiles --ranks 1,2,4,8 --pipeline > $O/wide_gemm.log 2>&1
echo "wide, no tail gemm:"; KPDI_F32_WIDE=1 KPDI_TAIL_GEMM=0 timeout 300 python tools/rank_share_probe.py $O/wide_nogemm.json --no-whole-tiles --ranks 1,2,4,8 --pipeline > $O/wide_nogemm.log 2>&1
python - <<PY
import json
for f in ("auto", "wide_gemm", "wide_nogemm"):
    d = json.load(open("$O/%s.json" % f))
    print(f, {r: (v["kernel"][:9], v["ms_per_step"], v["match_ms"], v.get("step_over_even_share")) for r, v in d["ranks"].items()})
PY
cd /tmp && export TMPDIR=/tmp
KPDI_F32_WIDE=1 KPDI_TAIL_GEMM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/tools/rank_share_probe.py --no-whole-tiles --ranks 4,8 --reps 10 > /dev/null 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
