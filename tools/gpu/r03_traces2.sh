R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r03tr; mkdir -p $O
for w in "cfg3 --envs 1024" "cfg3 --envs 2048" "cfg4 --envs 256"; do
  n=$(echo $w | tr -d ' -')
  a=k_ro; case "$w" in cfg4*) a=k_shape_fill;; esac
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$n -- python $R/bench.py --workload $w --steps 12 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python $R/tools/trace_timeline.py $(find /tmp/kt_$n -name "*kernel_trace.csv" | head -1) $a 3 > $O/timeline_$n.txt 2>&1
  echo "== $w"; cat $O/timeline_$n.txt
done
