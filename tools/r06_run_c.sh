#!/bin/bash
# round-6 check c: A/B of the tile order on the share legs (same box, alternating), then the rest of the GPU suite
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06c; rm -rf $O; mkdir -p $O; cd $R
for i in 1 2; do
  for ord in permuted natural; do
    KPDI_TILE_ORDER=$ord timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pcie --no-generation --no-config3 --no-traffic --check-rows 0 > $O/bench_${ord}_$i.json 2> $O/bench_${ord}_$i.err
    python - <<PY
import json
d = json.load(open("$O/bench_${ord}_$i.json"))
print("$ord $i", d["roofline"]["frac"], " ".join("%s %.3f/%.4f" % (k.replace("config", "c").replace("_share_of_", "s"), d["extra"][k]["match_ms"], d["extra"][k]["match_frac"]) for k in ("config2_share_of_4", "config2_share_of_8", "config4_share_of_8", "config5_share_of_8", "config5_share_of_8_f16") if k in d["extra"]))
PY
  done
done
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log
