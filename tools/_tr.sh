cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tr8; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr8 -o t -- python $R/tools/rank_share_probe.py --pmc-shard 8 --reps 6 > /tmp/tr8.log 2>&1
f=$(find /tmp/tr8 -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_gaps.py $f "copyBuffer" 14
echo "== serial tail"
rm -rf /tmp/tr8; KPDI_TAIL_SERIAL=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr8 -o t -- python $R/tools/rank_share_probe.py --pmc-shard 8 --reps 6 > /tmp/tr8.log 2>&1
f=$(find /tmp/tr8 -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_gaps.py $f "copyBuffer" 14
