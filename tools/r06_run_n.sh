#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06n; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do for t in r05 head; do
  [ $t == r05 ] && T=$R/build/r05tree || T=$R
  GRAFT_REPO_ROOT=$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/$t$i -o p -- python $T/tools/perf_probe.py --half --reps 4 --n 62500 --s 120 > /dev/null 2>&1
  python - <<PY
import csv, glob
for f in glob.glob("$O/$t$i/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "match16" in r["Name"] or "prep" in r["Name"]: print("$t$i", r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
done; done
