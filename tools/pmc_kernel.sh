#!/bin/bash
# counters of one kernel (substring match) through tools/perf_probe.py:  bash tools/pmc_kernel.sh <kernel substring> "<counters>" [perf_probe args]
set -u
kern=$1; ctrs=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/pmck
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --output-format csv -d $out -o p -- python $R/tools/perf_probe.py --reps 2 "$@" > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "$kern" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k:28s} n={len(v):3d} last={v[-1]:.4g}")
PY
