#!/bin/bash
# Counters of the float16 match kernel at configs[4]'s shape (one rank's share of 8: 4096 x 62 500 x 120^2, K = 14 400),
# through tools/perf_probe.py (on the GPU box):
#   bash tools/collect_f16_pmc.sh <tag> [n] [s]   -> gpurun_out/f16pmc_<tag>/{a..f}/ + gpurun_out/f16pmc_<tag>/summary.json
# Separate --pmc passes, nothing else in the command (the form the pool allows).  FETCH_SIZE is doubled (gfx950: 128-B
# requests tallied at 64 B, MI355X_MICROARCH.md "HBM"); *_SIZE are KiB.
set -u
tag=${1:-r06}
n=${2:-62500}
s=${3:-120}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cmd="python $R/tools/perf_probe.py --half --reps 2 --n $n --s $s"
out=$R/gpurun_out/f16pmc_$tag
rm -rf $out; mkdir -p $out
echo "$cmd" > $out/command.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o p -- $cmd > $out/stats.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/a -o p -- $cmd > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $out/b -o p -- $cmd > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/c -o p -- $cmd > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/d -o p -- $cmd > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/e -o p -- $cmd > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_BUSY_CYCLES --output-format csv -d $out/f -o p -- $cmd > /dev/null 2>&1
python - <<PY
import csv, glob, collections, json
out = "$out"
n, s, m = $n, $s, 4096
acc = collections.defaultdict(list)
for d in "abcdef":
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "match16" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                kname = r["Kernel_Name"]
dur = []
for f in glob.glob(f"{out}/stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "match16" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
last = {k: v[-1] for k, v in acc.items()}
res = {"command": open(f"{out}/command.txt").read().strip(), "kernel": kname if acc else None,
       "launches_ms": [round(x, 4) for x in dur], "counters_last_launch": last}
if dur:
    ms = dur[-1]
    k = s * s
    flops = 2.0 * m * n * k
    res["match_ms"] = round(ms, 4)
    res["tflops"] = round(flops / (ms * 1e-3) / 1e12, 1)
    res["frac_of_2500"] = round(flops / (ms * 1e-3) / 2.5e15, 4)
    algo = (n + m) * k * 2.0  # prepared float16 operands, each read once
    res["algorithmic_operand_bytes"] = algo
    if "FETCH_SIZE" in last:
        fetch = last["FETCH_SIZE"] * 1024 * 2
        res["fetch_bytes"] = fetch
        res["fetch_over_algorithmic"] = round(fetch / algo, 3)
        res["fetch_TBps"] = round(fetch / (ms * 1e-3) / 1e12, 3)
    if "WRITE_SIZE" in last:
        res["write_bytes"] = last["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in last:
        res["l2_hit_rate"] = round(last["TCC_HIT_sum"] / max(last["TCC_HIT_sum"] + last["TCC_MISS_sum"], 1), 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in last and "GRBM_GUI_ACTIVE" in last:
        # MFMA busy is summed over 1024 SIMDs, GRBM_GUI_ACTIVE over 8 XCDs (tools/summarize_pmc.py)
        res["mfma_busy"] = round((last["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (last["GRBM_GUI_ACTIVE"] / 8), 4)
        res["clock_GHz"] = round((last["GRBM_GUI_ACTIVE"] / 8) / (ms * 1e-3) / 1e9, 3)
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
