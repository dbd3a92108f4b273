#!/bin/bash
# Round 3, first GPU pass (run on the GPU box):  bash tools/collect_r03a.sh
# tests, default bench, what RCCL says to two ranks on one device, one rank's share of configs[1], [3], [4]
# with the traffic counters of the N = 8 shares.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err
KPDI_BENCH_SHARE_GPU=1 NCCL_DEBUG=WARN timeout 180 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_2ranks_1gpu.json 2> $O/bench_2ranks_1gpu.err; echo "rc=$?" >> $O/bench_2ranks_1gpu.err
timeout 300 python tools/rank_share_probe.py $O/rank_share_config2.json > $O/rank_share_config2.log 2>&1
timeout 600 python tools/rank_share_probe.py $O/rank_share_config4.json --workload config4 --no-whole-tiles > $O/rank_share_config4.log 2>&1
timeout 900 python tools/rank_share_probe.py $O/rank_share_config5.json --workload config5 --no-whole-tiles > $O/rank_share_config5.log 2>&1
timeout 900 python tools/rank_share_probe.py $O/rank_share_config5_f16.json --workload config5 --compute f16 > $O/rank_share_config5_f16.log 2>&1
cd /tmp && export TMPDIR=/tmp
for wl in "config4 f32" "config5 f32" "config5 f16"; do
  set -- $wl
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=$O/pmc_$1_$2_$ctr
    rm -rf $d
    timeout 600 rocprofv3 --pmc $ctr --output-format csv -d $d -o p -- python $R/tools/rank_share_probe.py --workload $1 --compute $2 --pmc-shard 8 --reps 2 > $d.log 2>&1
  done
done
python - <<PY > $O/pmc_shares.json
import csv, glob, json, collections
out = {}
for d in sorted(glob.glob("$O/pmc_*_*_*")):
    if d.endswith(".log"): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"].split("(")[0][:80], r["Counter_Name"])].append(float(r["Counter_Value"]))
    out[d.split("/")[-1]] = {f"{k[0]} | {k[1]}": {"launches": len(v), "mean": sum(v) / len(v), "last": v[-1]} for k, v in acc.items()}
print(json.dumps(out, indent=1))
PY
ls -la $O
