#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in 1 2; do for v in shipped noepi; do
  [ $v == shipped ] && unset KPDI_LIB_PATH || export KPDI_LIB_PATH=$R/build/variants/libkpdi_$v.so
  python $R/tools/perf_probe.py --half --reps 3 --n 62500 --s 120 2>&1 | grep "rep 3" | cut -c1-60 | sed "s/^/$v $i f16 K=14400: /"
  python $R/tools/perf_probe.py --reps 3 2>&1 | grep "rep 3" | cut -c1-60 | sed "s/^/$v $i f32 K=3600: /"
done; done
