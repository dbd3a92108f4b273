#!/bin/bash
# Everything profiles/r03_* is made from (run on the GPU box):  bash tools/collect_r03.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
# ---- the bench lines (the default one is what the driver runs)
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python bench.py --workload config3 --steps 20 --warmup 3 --no-pcie --no-generation > $O/bench_config3.json 2> $O/bench_config3.err
timeout 900 python bench.py --workload config4 --steps 5 --warmup 1 > $O/bench_config4_1gpu.json 2> $O/bench_config4.err
timeout 900 python bench.py --workload config5 --steps 3 --warmup 1 > $O/bench_config5_1gpu.json 2> $O/bench_config5.err
timeout 900 python bench.py --workload config5 --steps 3 --warmup 1 --compute f16 > $O/bench_config5_1gpu_f16.json 2> $O/bench_config5_f16.err
# ---- what RCCL says to two ranks on ONE device (the only multi-rank run a 1-GPU box allows)
KPDI_BENCH_SHARE_GPU=1 NCCL_DEBUG=WARN timeout 180 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_2ranks_1gpu.json 2> $O/bench_2ranks_1gpu.err; echo "exit code $?" >> $O/bench_2ranks_1gpu.err
# ---- rocprofv3 passes of the default command and of configs[2]
bash tools/collect_profiles.sh r03 --no-config3 --no-traffic > $O/collect.log 2>&1
bash tools/collect_profiles.sh r03_config3 --workload config3 --no-traffic >> $O/collect.log 2>&1
# ---- one rank's share of configs[1], [3], [4] (f32, float16, float16 from a float16-resident dictionary)
timeout 300 python tools/rank_share_probe.py $O/rank_share_config2.json > $O/rank_share_config2.log 2>&1
timeout 300 python tools/rank_share_probe.py $O/rank_share_config2_pipeline.json --pipeline --no-whole-tiles > $O/rank_share_config2_pipeline.log 2>&1
timeout 600 python tools/rank_share_probe.py $O/rank_share_config4.json --workload config4 --no-whole-tiles > $O/rank_share_config4.log 2>&1
timeout 900 python tools/rank_share_probe.py $O/rank_share_config5.json --workload config5 --no-whole-tiles > $O/rank_share_config5.log 2>&1
timeout 900 python tools/rank_share_probe.py $O/rank_share_config5_f16.json --workload config5 --compute f16 > $O/rank_share_config5_f16.log 2>&1
timeout 900 python tools/rank_share_probe.py $O/rank_share_config5_f16_dict16.json --workload config5 --compute f16 --dict-dtype f16 --ranks 1,8 > $O/rank_share_config5_f16_dict16.log 2>&1
(cd /tmp && export TMPDIR=/tmp
for wl in "config4 f32" "config5 f32" "config5 f16"; do
  set -- $wl
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=$O/pmc_$1_$2_$ctr
    timeout 600 rocprofv3 --pmc $ctr --output-format csv -d $d -o p -- python $R/tools/rank_share_probe.py --workload $1 --compute $2 --pmc-shard 8 --reps 2 > $d.log 2>&1
  done
done
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_config5_f16 -o b -- python $R/bench.py --workload config5 --steps 2 --warmup 1 --no-cpu-baseline --compute f16 --check-rows 0 --no-traffic > /dev/null 2>&1)
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_config4 -o b -- python $R/bench.py --workload config4 --steps 2 --warmup 1 --no-cpu-baseline --check-rows 0 --no-traffic > /dev/null 2>&1)
)
python - <<PY > $O/pmc_shares.json
import csv, glob, json, collections
out = {}
for d in sorted(glob.glob("$O/pmc_*_*_*")):
    if d.endswith(".log"): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"].split("(")[0][:80], r["Counter_Name"])].append(float(r["Counter_Value"]))
    ctr = d.rsplit("_", 2)[-2] + "_" + d.rsplit("_", 2)[-1]
    mult = 2048 if ctr == "FETCH_SIZE" else 1024   # KiB; FETCH_SIZE counts half of a wide coalesced read on gfx950
    out[d.split("/")[-1]] = {k[0]: {"launches": len(v), "launches_per_sweep": len(v) / 3, "GB_per_sweep": sum(v) / 3 * mult / 1e9}
                             for k, v in acc.items()}
print(json.dumps(out, indent=1))
PY
# ---- preparation and pre-processing kernels
{
echo "== dictionary preparation per 100 000 x 60x60 (ms in the 'prep' field include ~0.02 ms for the 4096 experimental patterns)"
echo "-- unmasked, f32 wide form"; timeout 200 python tools/perf_probe.py --reps 3 | tail -2 | head -1
echo "-- masked (K = 2819), gather kernel"; timeout 200 python tools/perf_probe.py --mask --reps 3 | tail -2 | head -1
echo "-- masked, round 2's LDS-staged kernel (KPDI_PREP_NO_GATHER=1)"; KPDI_PREP_NO_GATHER=1 timeout 200 python tools/perf_probe.py --mask --reps 3 | tail -2 | head -1
echo "-- masked, float16 form"; timeout 200 python tools/perf_probe.py --mask --half --reps 3 | tail -2 | head -1
echo "== float16 preparation of 62 500 x 120x120 (float32 raw rows)"
echo "-- pairs of rows, non-temporal loads (default)"; timeout 300 python tools/perf_probe.py --s 120 --n 62500 --half --reps 3 | tail -2 | head -1
echo "-- round 2's four rows per 1024-thread workgroup (KPDI_PREP16=block4)"; KPDI_PREP16=block4 timeout 300 python tools/perf_probe.py --s 120 --n 62500 --half --reps 3 | tail -2 | head -1
} > $O/prep_probe.txt 2>&1
{
timeout 300 python tools/prekernel_probe.py 60 60
timeout 300 python tools/prekernel_probe.py 120 120
echo "== the generic loop (round 2's correlation; KPDI_PRE_GENERIC=1)"
KPDI_PRE_GENERIC=1 timeout 300 python tools/prekernel_probe.py 60 60 | grep "M=4096"
KPDI_PRE_GENERIC=1 timeout 300 python tools/prekernel_probe.py 120 120 | grep "M=4096"
} > $O/prekernel_probe.txt 2>&1
{
for s in 60 120; do
echo "=== ${s} x ${s}, 16 384 patterns"
python tools/pk_probe.py $s 16384
bash tools/pmc_any.sh preproc_fused "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" -- python $R/tools/pk_probe.py $s 16384
bash tools/pmc_any.sh preproc_fused "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" -- python $R/tools/pk_probe.py $s 16384
bash tools/pmc_any.sh preproc_fused "FETCH_SIZE" -- python $R/tools/pk_probe.py $s 16384
bash tools/pmc_any.sh preproc_fused "WRITE_SIZE" -- python $R/tools/pk_probe.py $s 16384
done
} > $O/prekernel_pmc.txt 2>&1
# ---- float16 match kernel: pipe-busy fraction, clock; the matrix pipe alone
for v in "ship -" ; do set -- $v; bash tools/pmc_busy.sh $1 $2 >> $O/match16_busy.txt 2>&1; bash tools/pmc_busy.sh ${1}_k14400 $2 --s 120 --n 62500 >> $O/match16_busy.txt 2>&1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o /tmp/mfma_peak > /dev/null 2>&1 && /tmp/mfma_peak > $O/mfma_peak.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probes/div_check.hip -o /tmp/div_check > /dev/null 2>&1 && /tmp/div_check > $O/div_check.txt 2>&1
timeout 1200 python tools/form_probe.py $O/form_choice.json > $O/form_probe.log 2>&1
# ---- per-launch cost of the f32 match kernels; float64 host-chunk streaming; the float64 matrix pipe beside the vector ALU
{ echo "== match.hip (128 x 256 tiles, 16 workgroups per row block): python tools/tile_ramp_probe.py"; timeout 200 python tools/tile_ramp_probe.py
  echo; echo "== match16.hip f32 form (256 x 256 tiles): python tools/tile_ramp_probe.py wide"; timeout 200 python tools/tile_ramp_probe.py wide; } > $O/tile_ramp_probe.txt 2>&1
{ timeout 200 python tools/f64_stream_probe.py; echo "== KPDI_F64_SYNC=1 (round 2: the certification read back after every chunk)"; KPDI_F64_SYNC=1 timeout 200 python tools/f64_stream_probe.py; } > $O/f64_stream_probe.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/probes/mfma_f64_overlap.hip -o /tmp/overlap > /dev/null 2>&1 && timeout 120 /tmp/overlap > $O/mfma_f64_overlap.txt 2>&1
ls -la $O
