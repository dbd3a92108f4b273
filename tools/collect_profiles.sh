#!/bin/bash
# Collect the rocprofv3 passes of bench.py that tools/summarize_pmc.py reads (run on the GPU box):
#   bash tools/collect_profiles.sh r01c   ->  gpurun_out/{prof_r01c,pmc_fetch,pmc_write,pmc_sq}
# Counter passes are separate runs with --pmc only (no trace domains), as the pool requires.
set -u
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cmd="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pcie --check-rows 0"
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/prof_$tag" "$R/gpurun_out/pmc_fetch" "$R/gpurun_out/pmc_write" "$R/gpurun_out/pmc_sq"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_$tag" -o bench -- $cmd > "$R/gpurun_out/prof_$tag.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/pmc_fetch" -o bench -- $cmd > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/pmc_write" -o bench -- $cmd > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d "$R/gpurun_out/pmc_sq" -o bench -- $cmd > /dev/null 2>&1
# the summariser wants flat files: rocprofv3 nests them under <hostname>/<pid>_
for d in prof_$tag pmc_fetch pmc_write pmc_sq; do
  for f in $(find "$R/gpurun_out/$d" -name "*.csv"); do
    b=$(basename "$f"); b=${b#*_}
    cp "$f" "$R/gpurun_out/$d/bench_${b#bench_}" 2>/dev/null
  done
  ls "$R/gpurun_out/$d" | head -5
done
tail -2 "$R/gpurun_out/prof_$tag.log" | cut -c1-300
