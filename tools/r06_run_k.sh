#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06k; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in shipped nocompact; do
  [ $v == shipped ] && unset KPDI_LIB_PATH || export KPDI_LIB_PATH=$R/build/variants/libkpdi_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o p -- python $R/tools/rank_share_probe.py --no-whole-tiles --ranks 1,8 --reps 10 --pipeline > /dev/null 2>&1
  python - <<PY
import csv, glob
for f in glob.glob("$O/prof_$v/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "merge" in r["Name"] or "match16" in r["Name"]: print("$v", r["Name"][:50], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
done
