#!/bin/bash
# gpurun_out/r06c + gpurun_out/prof_r06* (merged back from the GPU box) -> profiles/r06_*, then the generated blocks of
# README.md / DESIGN.md  (run in the build container)
cd "$(dirname "$0")/.."
python tools/summarize_pmc.py gpurun_out r06 > /dev/null
python tools/summarize_pmc.py gpurun_out r06_config3 > /dev/null
O=gpurun_out/r06c
for f in bench_n1 bench_config3 bench_config4_1gpu bench_config5_1gpu_f16 bench_single_process_8_on_1gpu bench_two_ranks_host_gather_on_1gpu structured structured_natural_order; do
  [ -s $O/$f.json ] && cp $O/$f.json profiles/r06_$f.json
done
python - <<PY
import json
out = {"what": "one rank's share of a dictionary-sharded job on ONE MI355X (tools/rank_share_probe.py --pipeline): rank 0's shard of an "
               "N-rank job, inputs resident, whole step incl. preparation, merge and hand-over of the result; "
               "step_over_even_share = step / (t_1 / N) = what strong scaling can reach before the gather.  config2 = the shipped "
               "choice (match16.hip + tailgemm.hip); config2_matchhip = KPDI_F32_WIDE=0 (round 5's choice for these shares); "
               "config2_partial_units = match16.hip with the last round as quarter tiles inside the kernel (KPDI_TAIL_GEMM=0) - "
               "their even share is taken from config2's N = 1"}
for key in ("config2", "config2_matchhip", "config2_partial_units", "config4", "config5_f16_dict16"):
    try:
        out[key] = json.load(open("$O/rank_share_%s.json" % key))
    except Exception as e:
        out[key] = {"error": str(e)}
try:
    t1 = out["config2"]["ranks"]["1"]["ms_per_step"]
    for key in ("config2_matchhip", "config2_partial_units"):
        for n, r in out[key].get("ranks", {}).items():
            r["step_over_even_share_of_config2"] = round(r["ms_per_step"] / (t1 / int(n)), 4)
except Exception:
    pass
json.dump(out, open("profiles/r06_rank_share.json", "w"), indent=1)
PY
f=$(find $O/prof_share8 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/r06_share8_kernel_stats.csv
[ -s $O/share8_timeline.txt ] && { echo "# one rank's share of configs[1] at N = 8 under rocprofv3 --kernel-trace (tools/collect_r06.sh): kernels of consecutive steps in start order, idle gap before each (us)"; cat $O/share8_timeline.txt; } > profiles/r06_share8_timeline.txt
if [ -s $O/standalone_call.txt ]; then
  { echo "# kikuchipy_amd.dictionary_indexing(exp, dictionary IN HOST MEMORY, metric=\"ncc\", keep_n=20, n_per_iteration=..., device=0) at configs[1] on one MI355X"
    echo "# (tools/standalone_call_probe.py): wall time of the whole call, best of 3 after a warm-up call; results bit-identical in every row."
    echo "# shipped: the engine of a finished call is kept for the next one (kikuchipy_amd._lib: engine pool)"; cat $O/standalone_call.txt
    echo "# KPDI_ENGINE_CACHE=0: one engine per call, created and destroyed by it (rounds 1-5)"; cat $O/standalone_call_nocache.txt; } > profiles/r06_standalone_call.txt
fi
[ -s $O/ramp_wide.txt ] && { echo "# per-launch cost of the wide f32 kernel = intercept of match ms against whole tiles per workgroup (tools/tile_ramp_probe.py wide; profiling level 1)"; cat $O/ramp_wide.txt; } > profiles/r06_tile_ramp.txt
grep -aq "^block" $O/launch_phases.txt 2>/dev/null && { echo "# one rank's share at N = 8 on a developer build of match16.hip (-DKPDI16_TIME_PHASES, tools/probes/share_step.py): shader cycles of a launch's phases, blocks 0 / 100 / 255, wave 0"; grep -a "^block\|^---" $O/launch_phases.txt; } > profiles/r06_launch_phases.txt
grep -a "passed\|failed" $O/pytest_gpu.log | tail -2 > profiles/r06_pytest_gpu.txt
tail -2 $O/stress.log > profiles/r06_stress.txt
python tools/make_measurements.py
ls profiles | grep r06
