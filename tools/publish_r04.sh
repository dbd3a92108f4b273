#!/bin/bash
# gpurun_out/r04c + gpurun_out/prof_r04* (merged back from the GPU box) -> profiles/r04_*  (run in the build container)
cd "$(dirname "$0")/.."
python tools/summarize_pmc.py gpurun_out r04 > /dev/null
python tools/summarize_pmc.py gpurun_out r04_config3 > /dev/null
O=gpurun_out/r04c
for f in bench_n1 bench_config3 bench_config4_1gpu bench_config5_1gpu bench_config5_1gpu_f16 bench_config5_1gpu_f16_dict32 bench_single_process_8_on_1gpu bench_single_process_config4_4_on_1gpu; do
  [ -s $O/$f.json ] && cp $O/$f.json profiles/r04_$f.json
done
python - <<PY
import json
out = {"what": "one rank's share of a dictionary-sharded job on ONE MI355X (tools/rank_share_probe.py): rank 0's shard of an "
               "N-rank job, inputs resident, whole step incl. preparation, merge and hand-over of the result; "
               "step_over_even_share = step / (t_1 / N) = what strong scaling can reach before the gather"}
for key in ("config2", "config2_pipeline", "config4", "config5", "config5_f16_dict16"):
    try:
        out[key] = json.load(open("$O/rank_share_%s.json" % key))
    except Exception as e:
        out[key] = {"error": str(e)}
json.dump(out, open("profiles/r04_rank_share.json", "w"), indent=1)
PY
cp $O/tile_ramp_probe.txt profiles/r04_tile_ramp_probe.txt
grep -a "passed\|failed" $O/pytest_gpu.log | tail -2 > profiles/r04_pytest_gpu.txt  # (the RCCL banner of the group tests prints after the summary line)
for d in config4 config5_f16; do f=$(find $O/prof_$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/r04_${d}_kernel_stats.csv; done
ls profiles | grep r04
