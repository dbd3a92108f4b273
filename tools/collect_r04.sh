#!/bin/bash
# Everything profiles/r04_* is made from (run on the GPU box):  bash tools/collect_r04.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04c
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
# ---- the bench lines (the default one is what the driver runs)
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python bench.py --workload config3 --steps 20 --warmup 3 --no-pcie --no-generation > $O/bench_config3.json 2> $O/bench_config3.err
timeout 900 python bench.py --workload config4 --steps 5 --warmup 1 > $O/bench_config4_1gpu.json 2> $O/bench_config4.err
timeout 900 python bench.py --workload config5 --steps 3 --warmup 1 > $O/bench_config5_1gpu.json 2> $O/bench_config5.err
timeout 900 python bench.py --workload config5 --steps 3 --warmup 1 --compute f16 > $O/bench_config5_1gpu_f16.json 2> $O/bench_config5_f16.err
timeout 900 python bench.py --workload config5 --steps 3 --warmup 1 --compute f16 --dict-dtype f32 > $O/bench_config5_1gpu_f16_dict32.json 2> $O/bench_config5_f16_dict32.err
# ---- ONE process, a kpdi_group: 8 members sharing this box's one GPU (the whole multi-device code path; the timing is 8
# contexts contending for one device and means nothing), and 1 member over an in-process RCCL communicator
KPDI_BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 8 --single-process --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_single_process_8_on_1gpu.json 2> $O/bench_single_process_8_on_1gpu.err
KPDI_BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 4 --single-process --workload config4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_single_process_config4_4_on_1gpu.json 2> $O/bench_single_process_config4.err
# ---- rocprofv3 passes of the default command and of configs[2]
bash tools/collect_profiles.sh r04 --no-config3 --no-traffic > $O/collect.log 2>&1
bash tools/collect_profiles.sh r04_config3 --workload config3 --no-traffic >> $O/collect.log 2>&1
# ---- one rank's share of configs[1], [3], [4]
timeout 300 python tools/rank_share_probe.py $O/rank_share_config2.json > $O/rank_share_config2.log 2>&1
timeout 300 python tools/rank_share_probe.py $O/rank_share_config2_pipeline.json --pipeline --no-whole-tiles > $O/rank_share_config2_pipeline.log 2>&1
timeout 600 python tools/rank_share_probe.py $O/rank_share_config4.json --workload config4 --no-whole-tiles > $O/rank_share_config4.log 2>&1
timeout 900 python tools/rank_share_probe.py $O/rank_share_config5.json --workload config5 --no-whole-tiles > $O/rank_share_config5.log 2>&1
timeout 900 python tools/rank_share_probe.py $O/rank_share_config5_f16_dict16.json --workload config5 --compute f16 --dict-dtype f16 > $O/rank_share_config5_f16_dict16.log 2>&1
(cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_config5_f16 -o b -- python $R/bench.py --workload config5 --steps 2 --warmup 1 --no-cpu-baseline --compute f16 --check-rows 0 --no-traffic > /dev/null 2>&1)
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_config4 -o b -- python $R/bench.py --workload config4 --steps 2 --warmup 1 --no-cpu-baseline --check-rows 0 --no-traffic > /dev/null 2>&1)
)
{ echo "== match.hip (128 x 256 tiles, 16 workgroups per row block): python tools/tile_ramp_probe.py"; timeout 200 python tools/tile_ramp_probe.py
  echo; echo "== match16.hip f32 form (256 x 256 tiles): python tools/tile_ramp_probe.py wide"; timeout 200 python tools/tile_ramp_probe.py wide; } > $O/tile_ramp_probe.txt 2>&1
ls -la $O
