mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_f16.py tests/test_gpu_f16x2.py tests/test_gpu_f64.py tests/test_gpu_fullsize.py tests/test_gpu_degenerate.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
Q="--no-cpu-baseline --no-config3 --no-pcie --no-generation --no-traffic --check-rows 16"
for rep in 1 2 3; do
  for lib in new perreg; do
    if [ $lib = perreg ]; then export KPDI_LIB_PATH=$PWD/build/variants/libkpdi_perreg.so; else unset KPDI_LIB_PATH; fi
    python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib f32', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
    python bench.py $Q --compute f16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib f16', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
  done
done
unset KPDI_LIB_PATH
echo == ramp new; python tools/tile_ramp_probe.py wide | tail -3
export KPDI_LIB_PATH=$PWD/build/variants/libkpdi_perreg.so
echo == ramp perreg; python tools/tile_ramp_probe.py wide | tail -3
