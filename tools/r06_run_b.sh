#!/bin/bash
# round-6 check b: GPU suite + structured probe, permuted vs natural tile order
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06b; rm -rf $O; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
timeout 300 python bench_structured.py 8 > $O/structured.json 2> $O/structured.err; echo "structured rc=$?"; tail -3 $O/structured.err
KPDI_TILE_ORDER=natural timeout 300 python bench_structured.py 8 > $O/structured_natural.json 2> $O/structured_natural.err
python - <<PY
import json
for f in ("structured", "structured_natural"):
    try:
        d = json.load(open("$O/%s.json" % f))
    except Exception as e:
        print(f, "unreadable", e); continue
    for k in (None, "dictionary_sorted_ascending", "dictionary_sorted_descending"):
        r = d if k is None else d[k]
        print(f, k or "sampler order", {x: r.get(x) for x in ("match_ms", "match_frac", "ms_per_step", "candidates_appended_per_lane_list", "buffer_overflows_per_launch", "direct_first_tiles_per_launch")})
PY
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pcie --no-generation --no-config3 --no-traffic --check-rows 16 > $O/bench_short.json 2> $O/bench_short.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_short.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
for k in ("config2_share_of_4", "config2_share_of_8", "config4_share_of_8", "config5_share_of_8", "config5_share_of_8_f16"):
    r = d["extra"].get(k, {})
    print(k, {x: r.get(x) for x in ("ms_per_step", "match_ms", "match_frac", "match_form", "step_over_even_share")})
PY
