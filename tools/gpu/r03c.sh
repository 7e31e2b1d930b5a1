# kernel traces of the 512-env share with / without the fused optimiser step
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  CM_FUSED_STEP=$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$f -- python $R/bench.py --workload cfg3 --envs 512 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_f$f.json 2>/dev/null
  cp $(find /tmp/kt$f -name "*kernel_stats.csv" | head -1) $O/f${f}_kernel_stats.csv
  cp $(find /tmp/kt$f -name "*kernel_trace.csv" | head -1) $O/f${f}_kernel_trace.csv
done
ls -la $O
