#!/bin/bash
# A/B of the float16 kernel at configs[4]'s share: round 5's tree (build/r05tree) against HEAD, alternating on one box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in 1 2; do
  for t in r05 head; do
    [ $t == r05 ] && T=$R/build/r05tree || T=$R
    (cd $T && GRAFT_REPO_ROOT=$T python tools/perf_probe.py --half --reps 3 --n 62500 --s 120 2>&1 | grep "rep 3" | sed "s/^/$t $i f16 K=14400: /")
    (cd $T && GRAFT_REPO_ROOT=$T python tools/perf_probe.py --half --reps 3 2>&1 | grep "rep 3" | sed "s/^/$t $i f16 K=3600 : /")
  done
done
