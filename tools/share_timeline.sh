#!/bin/bash
# Kernel timeline of one rank's share of configs[1] at N = 8 in the pipelined loop (GPU box):
#   bash tools/share_timeline.sh > gpurun_out/r04/share_timeline.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=/tmp/kpdi_timeline; rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $R/tools/rank_share_probe.py /tmp/kpdi_tl.json --ranks 8 --pipeline --no-whole-tiles --reps 20 > $D/log.txt 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$D/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]          # the steady half of the run
names, gaps, durs = [], collections.defaultdict(list), collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    na, nb = a["Kernel_Name"].split("(")[0][-44:], b["Kernel_Name"].split("(")[0][-44:]
    gaps[(na, nb)].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
    durs[na].append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
print("kernel durations (us, mean over the steady half):")
for k, v in durs.items():
    print(f"  {k:46s} n={len(v):3d}  {sum(v) / len(v):9.1f}")
print("gap between the end of one kernel and the start of the next (us): mean / min / max")
tot = 0.0
for (a, b), v in gaps.items():
    print(f"  {a:46s} -> {b:46s} n={len(v):3d}  {sum(v) / len(v):7.1f} / {min(v):6.1f} / {max(v):6.1f}")
steps = len(durs[max(durs, key=lambda k: sum(durs[k]))])
print(f"sum of all gaps per step: {sum(sum(v) for v in gaps.values()) / steps:.1f} us over {steps} steps")
PY
