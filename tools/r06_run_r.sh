#!/bin/bash
# what the final stage of a wide-kernel launch costs: shipped vs a timing-only build without it; phases of the shipped code
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for i in 1 2; do for v in shipped skipfinal; do
  [ $v == shipped ] && unset KPDI_LIB_PATH || export KPDI_LIB_PATH=$R/build/variants/libkpdi_$v.so
  KPDI_F32_WIDE=1 python tools/perf_probe.py --reps 6 --n 12500 2>&1 | grep "rep [456]" | cut -c1-90 | sed "s/^/$v $i n=12500: /"
done; done
KPDI_LIB_PATH=$R/build/variants/libkpdi_phases.so timeout 200 python tools/probes/share_step.py 2>&1 | grep -a "^block\|^---" | tail -7
