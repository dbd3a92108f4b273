#!/bin/bash
# PMC passes of the float16 match kernel through tools/perf_probe.py (on the GPU box):
#   bash tools/pmc_match16.sh <tag> [lib]     -> gpurun_out/pmc16_<tag>/*.csv + a summary on stdout
set -u
tag=${1:-x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -n "${2:-}" ] && export KPDI_LIB_PATH=$R/$2
cmd="python $R/tools/perf_probe.py --half --reps 2"
out=$R/gpurun_out/pmc16_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/a -o p -- $cmd > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $out/b -o p -- $cmd > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/c -o p -- $cmd > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/d -o p -- $cmd > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for d in "abcd":
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "match16" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"{k:28s} n={len(v):3d} last={v[-1]:.4g}")
PY
