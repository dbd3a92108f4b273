cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5
for e in "" "KPDI_PRE_GENERIC=1"; do
  echo "=== $e"
  env $e timeout 300 python tools/prekernel_probe.py 60 60 | grep -E "M=4096|M=65536"
  env $e timeout 300 python tools/prekernel_probe.py 120 120 | grep -E "M=4096|M=16384"
done

build/div_check
