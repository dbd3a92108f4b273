#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for t in r05 head; do
  [ $t == r05 ] && T=$R/build/r05tree || T=$R
  (cd $T && GRAFT_REPO_ROOT=$T KPDI_LIB_PATH=$T/build/variants/libkpdi_epi.so python tools/perf_probe.py --half --reps 1 2>&1 | grep "block 100 wave 0" | tail -1 | sed "s/^/$t K=3600: /")
  (cd $T && GRAFT_REPO_ROOT=$T KPDI_LIB_PATH=$T/build/variants/libkpdi_epi.so python tools/perf_probe.py --half --reps 1 --n 62500 --s 120 2>&1 | grep "block 100 wave 0" | tail -1 | sed "s/^/$t K=14400: /")
done
