#!/bin/bash
# Collect the rocprofv3 passes of bench.py that tools/summarize_pmc.py reads (run on the GPU box):
#   bash tools/collect_profiles.sh <tag> [bench.py arguments]  ->  gpurun_out/prof_<tag>/{stats,pmc_fetch,pmc_write,pmc_sq}
# Counter passes are separate runs with --pmc only (no trace domains), as the pool requires.
set -u
tag=${1:-r02}
shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
# the default step counts of bench.py (what the driver runs), without the informational legs
cmd="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pcie --no-generation --no-rank-shares --check-rows 0 $*"
D=$R/gpurun_out/prof_$tag
rm -rf "$D"; mkdir -p "$D"
echo "$cmd" > "$D/command.txt"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$D/stats" -o bench -- $cmd > "$D/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/pmc_fetch" -o bench -- $cmd > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$D/pmc_write" -o bench -- $cmd > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d "$D/pmc_sq" -o bench -- $cmd > /dev/null 2>&1
# the summariser wants flat files: rocprofv3 nests them under <hostname>/<pid>_
for d in stats pmc_fetch pmc_write pmc_sq; do
  for f in $(find "$D/$d" -name "*.csv"); do
    b=$(basename "$f"); b=${b#*_}
    cp "$f" "$D/$d/bench_${b#bench_}" 2>/dev/null
  done
done
ls "$D"/*
