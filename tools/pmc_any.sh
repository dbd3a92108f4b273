#!/bin/bash
# counters of the kernels matching a substring, for any command:
#   bash tools/pmc_any.sh <kernel substring> "<counters>" -- <command ...>
set -u
kern=$1; ctrs=$2; shift 3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/pmc_any_$$
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --output-format csv -d $out -o p -- "$@" > $out/cmd.log 2>&1
python - <<PY
import csv, glob, collections
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "$kern" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k:28s} n={len(v):3d} last={v[-1]:.5g} mean={sum(v)/len(v):.5g}")
PY
rm -rf $out
