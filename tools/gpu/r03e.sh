R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CM_CRITIC_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --workload cfg3 --envs 512 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench.json 2>/dev/null
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cp $(find /tmp/kt -name "*kernel_trace.csv" | head -1) $O/kernel_trace.csv
