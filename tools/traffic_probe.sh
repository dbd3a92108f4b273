cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "A:KPDI_XCD_GRID=0 KPDI_FIXED_FRAC=0" "B:KPDI_XCD_GRID=0 KPDI_FIXED_FRAC=0.8" "C:KPDI_FIXED_FRAC=0.8" "D:KPDI_FIXED_FRAC=0.9" "E:KPDI_FIXED_FRAC=1.0" "F:KPDI_FIXED_FRAC=0.6"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  echo "== $tag $envs"
  env $envs python $R/tools/perf_probe.py --reps 4 2>&1 | grep "rep 4:"
  rm -rf $R/gpurun_out/tr_$tag
  env $envs rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/tr_$tag -o p -- python $R/tools/perf_probe.py --reps 2 > /dev/null 2>&1
  python - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/tr_$tag/**/*counter_collection.csv", recursive=True):
    v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "match_topk" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
    print("FETCH GB per launch (x2 corrected):", [round(x*1024*2/1e9,2) for x in v])
PY
done
