#!/bin/bash
# Round 3, second pass: the reworked preparation kernels + the fused initialisations.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03b
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
{
echo "== masked dictionary preparation, 100k x 60x60 (f32 wide form / classic form / float16 form)"
for e in "" "KPDI_PREP_NO_GATHER=1"; do
  echo "-- $e"; env $e timeout 200 python tools/perf_probe.py --mask --reps 3 | tail -2
  echo "-- $e KPDI_F32_WIDE=0"; env $e KPDI_F32_WIDE=0 timeout 200 python tools/perf_probe.py --mask --reps 3 | tail -2
  echo "-- $e --half"; env $e timeout 200 python tools/perf_probe.py --mask --half --reps 3 | tail -2
done
echo "== float16 preparation of 62 500 x 120x120 (prep16_block4 / XCD-affine block kernel)"
for e in "" "KPDI_PREP16=block"; do
  echo "-- $e"; env $e timeout 300 python tools/perf_probe.py --s 120 --n 62500 --half --reps 3 | tail -2
done
} > $O/prep_probe.txt 2>&1
timeout 300 python tools/rank_share_probe.py $O/rank_share_config2.json --no-whole-tiles > $O/rank_share_config2.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err
ls -la $O
