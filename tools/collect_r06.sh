#!/bin/bash
# Everything profiles/r06_* is made from (run on the GPU box):  bash tools/collect_r06.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06c
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
# ---- the bench lines (the default one is what the driver runs)
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python bench.py --workload config3 --steps 20 --warmup 3 --no-pcie --no-generation > $O/bench_config3.json 2> $O/bench_config3.err
timeout 900 python bench.py --workload config4 --steps 5 --warmup 1 > $O/bench_config4_1gpu.json 2> $O/bench_config4.err
timeout 900 python bench.py --workload config5 --steps 3 --warmup 1 --compute f16 > $O/bench_config5_1gpu_f16.json 2> $O/bench_config5_f16.err
# ---- the multi-GPU forms as far as a 1-GPU box shows them
KPDI_BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 8 --single-process --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_single_process_8_on_1gpu.json 2> $O/bench_single_process_8_on_1gpu.err
KPDI_BENCH_SHARE_GPU=1 KPDI_COMM_TIMEOUT=30 timeout 300 python bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_two_ranks_host_gather_on_1gpu.json 2> $O/bench_two_ranks_host_gather_on_1gpu.err
# ---- rocprofv3 passes of the default command and of configs[2]
bash tools/collect_profiles.sh r06 --no-config3 --no-traffic --no-structured > $O/collect.log 2>&1
bash tools/collect_profiles.sh r06_config3 --workload config3 --no-traffic >> $O/collect.log 2>&1
# ---- one rank's share of configs[1], [3], [4]: automatic, and the kernels of round 5 (match.hip / partial units) beside it
timeout 300 python tools/rank_share_probe.py $O/rank_share_config2.json --no-whole-tiles --pipeline > $O/rank_share_config2.log 2>&1
KPDI_F32_WIDE=0 timeout 300 python tools/rank_share_probe.py $O/rank_share_config2_matchhip.json --no-whole-tiles --pipeline --ranks 4,8 > /dev/null 2>&1
KPDI_F32_WIDE=1 KPDI_TAIL_GEMM=0 timeout 300 python tools/rank_share_probe.py $O/rank_share_config2_partial_units.json --no-whole-tiles --pipeline --ranks 4,8 > /dev/null 2>&1
timeout 600 python tools/rank_share_probe.py $O/rank_share_config4.json --workload config4 --no-whole-tiles > $O/rank_share_config4.log 2>&1
timeout 900 python tools/rank_share_probe.py $O/rank_share_config5_f16_dict16.json --workload config5 --compute f16 --dict-dtype f16 > $O/rank_share_config5_f16_dict16.log 2>&1
# ---- the N = 8 share under rocprofv3: kernel by kernel, and the timeline of a step
(cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_share8 -o p -- python $R/tools/rank_share_probe.py --no-whole-tiles --ranks 8 --reps 20 --pipeline > /dev/null 2>&1)
python tools/trace_gaps.py $(find $O/prof_share8 -name "*kernel_trace.csv" | head -1) "prep_wave_lines_kernel<unsigned char" 18 > $O/share8_timeline.txt 2>&1
# ---- structured workload: shipped order and the natural tile order beside it
timeout 300 python bench_structured.py 8 > $O/structured.json 2> $O/structured.err
KPDI_TILE_ORDER=natural timeout 300 python bench_structured.py 8 > $O/structured_natural_order.json 2> /dev/null
# ---- the stand-alone call, the tile ramp, the launch phases (developer build, if it travelled)
timeout 300 python tools/standalone_call_probe.py > $O/standalone_call.txt 2>/dev/null
KPDI_ENGINE_CACHE=0 timeout 300 python tools/standalone_call_probe.py > $O/standalone_call_nocache.txt 2>/dev/null
timeout 300 python tools/tile_ramp_probe.py wide > $O/ramp_wide.txt 2>&1
[ -f build/variants/libkpdi_phases.so ] && KPDI_LIB_PATH=$R/build/variants/libkpdi_phases.so timeout 200 python tools/probes/share_step.py > $O/launch_phases.txt 2>&1
timeout 600 python tools/stress.py 300 > $O/stress.log 2>&1
ls -la $O
