"""Summarise rocprofv3 outputs of `bench.py` into profiles/ (developer tool).

    python tools/summarize_pmc.py gpurun_out r01

Reads  <dir>/prof_<tag>/stats/bench_kernel_stats.csv          (--kernel-trace --stats)
       <dir>/prof_<tag>/pmc_fetch/bench_counter_collection.csv (--pmc FETCH_SIZE)
       <dir>/prof_<tag>/pmc_write/bench_counter_collection.csv (--pmc WRITE_SIZE)
       <dir>/prof_<tag>/pmc_sq/bench_counter_collection.csv    (--pmc SQ_* GRBM_GUI_ACTIVE)
Writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json and
profiles/<tag>_summary.md.

Corrections (/opt/skills/guides/MI355X_MICROARCH.md, section HBM): FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide
coalesced read, so it is doubled.  Both corrections are calibrated in the same
run on prep_kernel<float>, whose traffic is known exactly (it reads the raw
dictionary once and writes the K-padded prepared copy once).
"""
import collections
import csv
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)


def counters(name):
    path = os.path.join(src, f"prof_{tag}", name, "bench_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return agg
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def mean(v):
    return sum(v) / len(v) if v else None


stats_src = os.path.join(src, f"prof_{tag}", "stats", "bench_kernel_stats.csv")
command = open(os.path.join(src, f"prof_{tag}", "command.txt")).read().strip().replace(root + "/", "")
shutil.copy(stats_src, os.path.join(out, f"{tag}_kernel_stats.csv"))
stats = {r["Name"]: r for r in csv.DictReader(open(stats_src))}

fetch, write, sq = counters("pmc_fetch"), counters("pmc_write"), counters("pmc_sq")
summary = {"tag": tag, "command": command, "kernels": {}}
for name, row in stats.items():
    k = {"calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6,
         "percentage": float(row["Percentage"])}
    f = mean(fetch.get(name, {}).get("FETCH_SIZE", []))
    w = mean(write.get(name, {}).get("WRITE_SIZE", []))
    if f is not None:
        k["fetch_bytes_per_launch"] = f * 1024 * 2  # KiB, gfx950 half-count correction
    if w is not None:
        k["write_bytes_per_launch"] = w * 1024
    for c, v in sq.get(name, {}).items():
        k[c] = mean(v)
    if k.get("SQ_VALU_MFMA_BUSY_CYCLES") and k.get("GRBM_GUI_ACTIVE"):
        # MFMA busy is summed over 1024 SIMDs, GRBM_GUI_ACTIVE over 8 XCDs
        k["mfma_busy_frac"] = (k["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (k["GRBM_GUI_ACTIVE"] / 8)
        k["clock_ghz_profiled"] = (k["GRBM_GUI_ACTIVE"] / 8) / (k["avg_ms"] * 1e-3) / 1e9
    summary["kernels"][name] = k

def busiest(pattern):
    hits = [n for n in summary["kernels"] if pattern in n]
    return max(hits, key=lambda n: summary["kernels"][n]["percentage"]) if hits else None


match = busiest("match16_kernel<20, false, 4, true>") or busiest("match_topk_kernel<20, false, 0, 4>") or \
    busiest("match_topk_kernel") or busiest("match16_kernel")
if match:
    mk = summary["kernels"][match]
    summary["match_kernel"] = match
    summary["match_traffic_bytes_per_launch"] = mk.get("fetch_bytes_per_launch", 0) + mk.get("write_bytes_per_launch", 0)
# the match kernel launch by launch (kernel trace): the first launches of a process run at a clock that is still ramping
trace_src = os.path.join(src, f"prof_{tag}", "stats", "bench_kernel_trace.csv")
launch_ms = []
if match and os.path.exists(trace_src):
    rows = [r for r in csv.DictReader(open(trace_src)) if r["Kernel_Name"] == match]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    launch_ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    summary["match_launch_ms"] = [round(v, 4) for v in launch_ms]
    warm = 3 if len(launch_ms) > 6 else 2
    summary["match_steady_ms"] = mean(launch_ms[warm:])
    summary["match_steady_from_launch"] = warm
json.dump(summary, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)

lines = [f"# rocprofv3 summary {tag}: `{command}` (tools/collect_profiles.sh; "
         "one --kernel-trace --stats pass and three --pmc passes of the same command; fetch = FETCH_SIZE x 1024 x 2, the gfx950 correction)", "",
         "| kernel | calls | avg ms | % | fetch GB/launch | write GB/launch | MFMA busy | clock GHz |",
         "|---|---|---|---|---|---|---|---|"]
for name, k in summary["kernels"].items():
    lines.append("| `%s` | %d | %.4f | %.2f | %s | %s | %s | %s |" % (
        name[:70], k["calls"], k["avg_ms"], k["percentage"],
        "%.3f" % (k["fetch_bytes_per_launch"] / 1e9) if "fetch_bytes_per_launch" in k else "-",
        "%.3f" % (k["write_bytes_per_launch"] / 1e9) if "write_bytes_per_launch" in k else "-",
        "%.3f" % k["mfma_busy_frac"] if "mfma_busy_frac" in k else "-",
        "%.2f" % k["clock_ghz_profiled"] if "clock_ghz_profiled" in k else "-"))
if launch_ms:
    lines += ["", "Match kernel launch by launch (ms, kernel trace): " + ", ".join("%.2f" % v for v in launch_ms),
              "Mean of the launches after the first %d (the timed steps of the bench): **%.3f ms**; all %d launches: %.3f ms." % (
                  summary["match_steady_from_launch"], summary["match_steady_ms"], len(launch_ms), mean(launch_ms))]
open(os.path.join(out, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
