#!/bin/bash
# Everything profiles/r02_* is made from (run on the GPU box):  bash tools/collect_r02.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
bash tools/collect_profiles.sh r02 --no-config3 > $O/collect.log 2>&1
bash tools/collect_profiles.sh r02_config3 --workload config3 >> $O/collect.log 2>&1
python bench.py --steps 20 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --workload config3 --steps 20 --warmup 3 --no-pcie --no-generation > $O/bench_config3.json 2> $O/bench_config3.err
python tools/rank_share_probe.py $O/rank_share.json > $O/rank_share.log 2>&1
python tools/prekernel_probe.py > $O/prekernel_probe.txt 2>&1
python tools/prekernel_probe.py 120 120 >> $O/prekernel_probe.txt 2>&1
bash tools/pmc_match16.sh r02 > $O/match16_pmc.txt 2>&1
python bench.py --workload config4 --steps 5 --warmup 1 > $O/bench_config4_1gpu.json 2> $O/bench_config4.err
python bench.py --workload config5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_config5_1gpu.json 2> $O/bench_config5.err
python bench.py --workload config5 --steps 3 --warmup 1 --no-cpu-baseline --compute f16 > $O/bench_config5_1gpu_f16.json 2> $O/bench_config5_f16.err
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_config5_f16 -o b -- python $R/bench.py --workload config5 --steps 2 --warmup 1 --no-cpu-baseline --compute f16 --check-rows 0 > /dev/null 2>&1)
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_config4 -o b -- python $R/bench.py --workload config4 --steps 2 --warmup 1 --no-cpu-baseline --check-rows 0 > /dev/null 2>&1)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o /tmp/mfma_peak > /dev/null 2>&1 && /tmp/mfma_peak > $O/mfma_peak.txt 2>&1
python tools/f64_probe.py > $O/f64_probe.txt 2>&1
for v in "" "-DKPDI16_NO_EPILOGUE"; do
  t=ship; [ -n "$v" ] && t=noepi && bash tools/build_variant.sh noepi match16.hip $v > /dev/null 2>&1
  lib=-; [ "$t" == "noepi" ] && lib=build/variants/libkpdi_noepi.so
  bash tools/pmc_busy.sh $t $lib >> $O/match16_busy.txt 2>&1
  bash tools/pmc_busy.sh ${t}_k14400 $lib --s 120 --n 62500 >> $O/match16_busy.txt 2>&1
done
ls -la $O
