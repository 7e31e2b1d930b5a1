R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03w
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in "cfg3" "cfg5" "cfg2"; do
  n=$(echo $w | tr -d ' -')
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k_$n -- python $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_$n.json 2>/dev/null
  cp $(find /tmp/k_$n -name "*kernel_stats.csv" | head -1) $O/${n}_kernel_stats.csv
  grep -E "k_reduce_step|k_gru2|k_mlp<1" $O/${n}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
done
