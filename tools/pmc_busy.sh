#!/bin/bash
# matrix-pipe busy fraction and clock of the float16 match kernel (one PMC pass through tools/perf_probe.py):
#   bash tools/pmc_busy.sh <tag> [lib] [perf_probe args]
set -u
tag=${1:-x}; lib=${2:-}; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -n "$lib" ] && [ "$lib" != "-" ] && export KPDI_LIB_PATH=$R/$lib
out=$R/gpurun_out/pmcbusy_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $out -o p -- python $R/tools/perf_probe.py --half --reps 3 "$@" 2>&1 | grep "rep 3" | cut -c1-60
python - <<PY
import csv, glob, collections
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "match16" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    busy = acc["SQ_VALU_MFMA_BUSY_CYCLES"][-1] / 1024 / (acc["GRBM_GUI_ACTIVE"][-1] / 8)
    print(f"$tag: MFMA busy / 1024 = {acc['SQ_VALU_MFMA_BUSY_CYCLES'][-1]/1024:.4g}  GRBM/8 = {acc['GRBM_GUI_ACTIVE'][-1]/8:.4g} cycles  pipe busy {busy:.3f}")
PY
