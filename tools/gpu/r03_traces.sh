R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r03tr; mkdir -p $O
for w in "cfg3" "cfg4" "cfg5"; do
  n=$(echo $w | tr -d ' -')
  a=k_ro; [ $w = cfg4 ] && a=k_shape_fill; [ $w = cfg5 ] && a=k_gru32_ro
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$n -- python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python $R/tools/trace_timeline.py $(find /tmp/kt_$n -name "*kernel_trace.csv" | head -1) $a 3 > $O/timeline_$n.txt 2>&1
done
cat $O/timeline_cfg3.txt; cat $O/timeline_cfg4.txt; head -30 $O/timeline_cfg5.txt
