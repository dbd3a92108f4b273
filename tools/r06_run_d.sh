#!/bin/bash
# round-6 check d: the wide kernel's tail units (shift 2 / 3 / 4) on one rank's share at N = 4, 8; correctness of the new units
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06d; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_group.py -m gpu -q -x > $O/pytest_a.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_a.log
for sh in 2 3 4; do
  KPDI_F32_WIDE=1 KPDI_WIDE_TAIL_SHIFT=$sh timeout 300 python -m pytest tests/test_gpu_engine.py -m gpu -q -x -k "golden or chunk or ties or kernels_agree or share" > $O/pytest_sh$sh.log 2>&1; echo "shift $sh pytest rc=$?"; tail -2 $O/pytest_sh$sh.log
done
echo "automatic:"; timeout 300 python tools/rank_share_probe.py $O/auto.json --no-whole-tiles --ranks 1,4,8 --pipeline 2>&1 | grep -v "^#" | tail -5
for sh in 0 2 3 4; do
  echo "wide, tail shift $sh:"
  KPDI_F32_WIDE=1 KPDI_WIDE_TAIL_SHIFT=$sh timeout 300 python tools/rank_share_probe.py $O/wide_sh$sh.json --no-whole-tiles --ranks 4,8 --pipeline 2>&1 | tail -3
done
python - <<PY
import json
for f in ("auto", "wide_sh0", "wide_sh2", "wide_sh3", "wide_sh4"):
    d = json.load(open("$O/%s.json" % f))
    print(f, {r: (v["kernel"][:9], v["ms_per_step"], v["match_ms"]) for r, v in d["ranks"].items()})
PY
