R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03u
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in "cfg3 --envs 512" "cfg2" "cfg5"; do
  n=$(echo $w | tr -d ' -')
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$n -- python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$n.json 2>/dev/null
  python $R/tools/trace_timeline.py $(find /tmp/kt_$n -name "*kernel_trace.csv" | head -1) k_ro 3 > $O/timeline_$n.txt 2>&1
done
