#!/bin/bash
# gpurun_out/r05c + gpurun_out/prof_r05* (merged back from the GPU box) -> profiles/r05_*  (run in the build container)
cd "$(dirname "$0")/.."
python tools/summarize_pmc.py gpurun_out r05 > /dev/null
python tools/summarize_pmc.py gpurun_out r05_config3 > /dev/null
O=gpurun_out/r05c
for f in bench_n1 bench_config3 bench_config4_1gpu bench_config5_1gpu_f16 bench_single_process_8_on_1gpu bench_two_ranks_host_gather_on_1gpu; do
  [ -s $O/$f.json ] && cp $O/$f.json profiles/r05_$f.json
done
python - <<PY
import json
out = {"what": "one rank's share of a dictionary-sharded job on ONE MI355X (tools/rank_share_probe.py): rank 0's shard of an "
               "N-rank job, inputs resident, whole step incl. preparation, merge and hand-over of the result; "
               "step_over_even_share = step / (t_1 / N) = what strong scaling can reach before the gather"}
for key in ("config2", "config2_pipeline", "config4", "config5_f16_dict16"):
    try:
        out[key] = json.load(open("$O/rank_share_%s.json" % key))
    except Exception as e:
        out[key] = {"error": str(e)}
json.dump(out, open("profiles/r05_rank_share.json", "w"), indent=1)
PY
{ echo "== with coalescing (the default, round 5)"; cat $O/group_chunks.txt; echo; echo "== KPDI_NO_COALESCE=1 (every chunk swept on arrival: rounds 1-4; 'new' here = the quota assignment alone)"; cat $O/group_chunks_nocoalesce.txt; } > profiles/r05_group_chunks.txt
cp $O/f64_bounds.txt profiles/r05_f64_bounds.txt
grep -a "passed\|failed" $O/pytest_gpu.log | tail -2 > profiles/r05_pytest_gpu.txt
f=$(find $O/prof_config5_f16 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/r05_config5_f16_kernel_stats.csv
ls profiles | grep r05
