# round 6, later session: A/B of an experimental counted path of the merge kernel (not shipped: profiles/r06_merge_counted_ab.txt) against a build without it
python -m pytest tests/test_gpu_engine.py tests/test_gpu_tail.py tests/test_gpu_group.py tests/test_gpu_api.py tests/test_gpu_config3.py tests/test_seam.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
for i in 1 2; do
  for v in "" build/variants/libkpdi_nocounted.so; do
    KPDI_LIB_PATH=$v python tools/rank_share_probe.py --ranks 1,8 --reps 20 2>&1 | grep -E "^(1|8) " | sed -E "s/.*'ms_per_step': ([0-9.]+).*'match_ms': ([0-9.]+).*'merge_ms': ([0-9.]+).*/step \1 match \2 merge \3/" | tr '\n' ';'; echo " <- ${v:-counted path}"
  done
done
