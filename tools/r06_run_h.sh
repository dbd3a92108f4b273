#!/bin/bash
# round-6 check h: whole GPU suite + the default bench line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06h; rm -rf $O; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; tail -3 $O/bench_n1.err
python - <<PY
import json
d = json.load(open("$O/bench_n1.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"].get("traffic"))
e = d["extra"]
for k in ("config3", "structured_config2", "config2_share_of_4", "config2_share_of_8", "config4_share_of_8", "config5_share_of_8", "config5_share_of_8_f16"):
    r = e.get(k, {})
    print(k, {x: r.get(x) for x in ("ms_per_step", "match_ms", "match_frac", "match_form", "step_over_even_share")})
print("structured:", {k: e["structured_config2"][k]["match_frac"] for k in ("dictionary_sorted_ascending", "dictionary_sorted_descending")})
print("standalone:", e.get("standalone_call"))
print("seam:", {k: (v["patterns_per_s"], v["ms_per_call"]) for k, v in e.get("plugin_seam", {}).items() if isinstance(v, dict)})
print("errors:", {k: v for k, v in e.items() if k.endswith("_error")})
print("cpu:", d.get("cpu_baseline", {}).get("value"))
PY
