#!/bin/bash
# A/B of the padded XCD grid (plan_xcd_grid, KPDI_XCD_PAD): one rank's share of configs[3] at N = 8 (157 row blocks =
# 4 launches of 32 + one of 29), time by tools/rank_share_probe.py, fabric traffic by rocprofv3 --pmc FETCH_SIZE.
cd "$(dirname "$0")/.." && R=$PWD
O=gpurun_out/r04/xcd_pad; mkdir -p $O; export TMPDIR=/tmp
for pad in 1 0 1 0; do
  KPDI_XCD_PAD=$pad timeout 600 python tools/rank_share_probe.py $O/share_pad$pad.json --workload config4 --ranks 8 --no-whole-tiles --pipeline > $O/share_pad$pad.log 2>&1
  python - <<PY
import json
d=json.load(open("$O/share_pad$pad.json"))["ranks"]["8"]
print("KPDI_XCD_PAD=$pad  ms_per_step", d["ms_per_step"], " match_ms", d["match_ms"], " frac", d["match_frac_of_peak"])
PY
done
for pad in 1 0; do
  d=$O/pmc_pad$pad; rm -rf $d
  (cd /tmp && KPDI_XCD_PAD=$pad timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$d -o p -- python $R/tools/rank_share_probe.py --workload config4 --pmc-shard 8 --reps 2 > $R/$d.log 2>&1)
  python - <<PY
import csv, glob
tot=0; n=0
for f in glob.glob("$d/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "match" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE":
            tot+=float(r["Counter_Value"]); n+=1
print("KPDI_XCD_PAD=$pad  FETCH_SIZE x 2 x 1024 over", n, "match launches:", round(tot*2048/1e9,1), "GB =", round(tot*2048/1e9/(n/5),1), "GB per sweep (5 launches per sweep)")
PY
done
