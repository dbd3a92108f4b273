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
if [ -s $O/standalone_call.txt ]; then
  { cat <<'TXT'
# The stand-alone driver as a user calls it (tools/standalone_call_probe.py): kikuchipy_amd.dictionary_indexing(exp, dictionary IN HOST
# MEMORY, metric="ncc", keep_n=20, n_per_iteration=..., device=0) at configs[1] (4096 x 100 000 x 60 x 60) on one MI355X; wall time of
# the whole call - engine made and closed by the call - best of 3 after one warm-up call; results bit-identical in every row.
# "shipped": a small host chunk is uploaded straight into its pending rows on the copy stream (two pending buffers);
# KPDI_NO_DIRECT_UPLOAD=1: the path until late in round 5 - staging buffer -> device-to-device copy on the COMPUTE stream, behind the sweeps.
TXT
    cat $O/standalone_call.txt
    cat <<'TXT'
#
# Where a single-pass call's ~38 ms go (one engine per call): create 2.1, upload of the experimental set 0.35, push (upload-bound:
# 1.44 GB at 53 GB/s from pageable memory - 54 GB/s from page-locked memory: the link, not the source - the sweep beside it) 29.4,
# hand-over 1.1, close (hipFree of the engine's buffers) 6-7 ms; the same sweep on an engine that is kept (EBSD, a metric instance,
# ResidentDictionary): 27.6 ms = bench.py's pcie_inclusive leg.  The chunked call's remaining ~7 ms over the single pass: the sweep of
# the last pending rows has no upload left to hide behind (~3.3 ms), seven launch ramps, 33 host round trips.
TXT
  } > profiles/r05_standalone_call.txt
fi
grep -a "passed\|failed" $O/pytest_gpu.log | tail -2 > profiles/r05_pytest_gpu.txt
f=$(find $O/prof_config5_f16 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/r05_config5_f16_kernel_stats.csv
ls profiles | grep r05
