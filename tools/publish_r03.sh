#!/bin/bash
# gpurun_out/r03 + gpurun_out/prof_r03* (merged back from the GPU box) -> profiles/r03_*  (run in the build container)
cd "$(dirname "$0")/.."
python tools/summarize_pmc.py gpurun_out r03 > /dev/null
python tools/summarize_pmc.py gpurun_out r03_config3 > /dev/null
O=gpurun_out/r03
for f in bench_n1 bench_config3 bench_config4_1gpu bench_config5_1gpu bench_config5_1gpu_f16; do cp $O/$f.json profiles/r03_$f.json; done
python - <<PY
import json
out = {"what": "one rank's share of a dictionary-sharded job on ONE MI355X (tools/rank_share_probe.py): rank 0's shard of an "
               "N-rank job, inputs resident, whole step incl. preparation, merge and hand-over of the result; "
               "step_over_even_share = step / (t_1 / N) = what strong scaling can reach before the RCCL all-gather"}
for key in ("config2", "config2_pipeline", "config4", "config5", "config5_f16", "config5_f16_dict16"):
    out[key] = json.load(open("$O/rank_share_%s.json" % key))
json.dump(out, open("profiles/r03_rank_share.json", "w"), indent=1)
PY
cp $O/pmc_shares.json profiles/r03_rank_share_pmc.json
cp $O/prep_probe.txt profiles/r03_prep_probe.txt
cp $O/prekernel_probe.txt profiles/r03_prekernel_probe.txt
cp $O/prekernel_pmc.txt profiles/r03_prekernel_pmc.txt
cp $O/match16_busy.txt profiles/r03_match16_busy.txt
cp $O/mfma_peak.txt profiles/r03_mfma_power_probe.txt
cp $O/div_check.txt profiles/r03_div_check.txt
cp $O/form_choice.json profiles/r03_form_choice_final.json
cp $O/tile_ramp_probe.txt profiles/r03_tile_ramp_probe.txt
cp $O/f64_stream_probe.txt profiles/r03_f64_stream_probe.txt
cp $O/mfma_f64_overlap.txt profiles/r03_mfma_f64_overlap.txt
grep -E "Duplicate GPU|invalid usage|exit code|bench.py: rank" $O/bench_2ranks_1gpu.err | sed 's#/longer_pathname[^ ]*/##' | sort -u > profiles/r03_rccl_two_ranks_one_gpu.txt
for d in config4 config5_f16; do f=$(find $O/prof_$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/r03_${d}_kernel_stats.csv; done
ls profiles | grep r03
