#!/bin/bash
# gpurun_out/r02 + gpurun_out/prof_r02* (merged back from the GPU box) -> profiles/r02_*  (run in the build container)
cd "$(dirname "$0")/.."
python tools/summarize_pmc.py gpurun_out r02 > /dev/null
python tools/summarize_pmc.py gpurun_out r02_config3 > /dev/null
O=gpurun_out/r02
for f in bench_n1 bench_config3 bench_config4_1gpu bench_config5_1gpu bench_config5_1gpu_f16 rank_share; do cp $O/$f.json profiles/r02_$f.json; done
cp $O/prekernel_probe.txt profiles/r02_prekernel_probe.txt
cp $O/match16_pmc.txt profiles/r02_match16_pmc.txt
cp $O/match16_busy.txt profiles/r02_match16_busy.txt
cp $O/f64_probe.txt profiles/r02_f64_probe.txt
{ cat $O/mfma_peak.txt; sed -n '/^# tools\/probes/,$p' profiles/r02_mfma_power_probe.txt; } > /tmp/mfma_probe.txt && cp /tmp/mfma_probe.txt profiles/r02_mfma_power_probe.txt
for d in config4 config5_f16; do f=$(find $O/prof_$d -name "*kernel_stats.csv" | head -1); cp $f profiles/r02_${d}_kernel_stats.csv; done
ls profiles | grep r02
