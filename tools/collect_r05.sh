#!/bin/bash
# Everything profiles/r05_* is made from (run on the GPU box):  bash tools/collect_r05.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05c
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
# ---- the bench lines (the default one is what the driver runs)
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python bench.py --workload config3 --steps 20 --warmup 3 --no-pcie --no-generation > $O/bench_config3.json 2> $O/bench_config3.err
timeout 900 python bench.py --workload config4 --steps 5 --warmup 1 > $O/bench_config4_1gpu.json 2> $O/bench_config4.err
timeout 900 python bench.py --workload config5 --steps 3 --warmup 1 --compute f16 > $O/bench_config5_1gpu_f16.json 2> $O/bench_config5_f16.err
# ---- the multi-GPU forms as far as a 1-GPU box shows them: ONE process over 8 members sharing the GPU (kpdi_group), and
# TWO processes on the one GPU - RCCL refuses the duplicate device, the ranks agree on the host-staged gather
KPDI_BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 8 --single-process --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_single_process_8_on_1gpu.json 2> $O/bench_single_process_8_on_1gpu.err
KPDI_BENCH_SHARE_GPU=1 KPDI_COMM_TIMEOUT=30 timeout 300 python bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_two_ranks_host_gather_on_1gpu.json 2> $O/bench_two_ranks_host_gather_on_1gpu.err
# ---- rocprofv3 passes of the default command and of configs[2]
bash tools/collect_profiles.sh r05 --no-config3 --no-traffic > $O/collect.log 2>&1
bash tools/collect_profiles.sh r05_config3 --workload config3 --no-traffic >> $O/collect.log 2>&1
# ---- one rank's share of configs[1], [3], [4]; the chunked call on a group
timeout 300 python tools/rank_share_probe.py $O/rank_share_config2.json --no-whole-tiles > $O/rank_share_config2.log 2>&1
timeout 300 python tools/rank_share_probe.py $O/rank_share_config2_pipeline.json --pipeline --no-whole-tiles > $O/rank_share_config2_pipeline.log 2>&1
timeout 600 python tools/rank_share_probe.py $O/rank_share_config4.json --workload config4 --no-whole-tiles > $O/rank_share_config4.log 2>&1
timeout 900 python tools/rank_share_probe.py $O/rank_share_config5_f16_dict16.json --workload config5 --compute f16 --dict-dtype f16 > $O/rank_share_config5_f16_dict16.log 2>&1
timeout 300 python tools/group_chunk_probe.py $O/group_chunks.txt > /dev/null 2>&1
KPDI_NO_COALESCE=1 timeout 300 python tools/group_chunk_probe.py $O/group_chunks_nocoalesce.txt > /dev/null 2>&1
timeout 300 python tools/f64_probe.py $O/f64_bounds.txt > /dev/null 2>&1
# ---- the stand-alone driver as a user calls it (host dictionary; one pass, a quarter, the tutorial's chunking), both upload paths
{ echo "# shipped"; timeout 300 python tools/standalone_call_probe.py 2>/dev/null | grep "n_per_iteration="; echo "# KPDI_NO_DIRECT_UPLOAD=1"; KPDI_NO_DIRECT_UPLOAD=1 timeout 300 python tools/standalone_call_probe.py 2>/dev/null | grep "n_per_iteration="; } > $O/standalone_call.txt
(cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_config5_f16 -o b -- python $R/bench.py --workload config5 --steps 2 --warmup 1 --no-cpu-baseline --compute f16 --check-rows 0 --no-traffic > /dev/null 2>&1)
)
ls -la $O
