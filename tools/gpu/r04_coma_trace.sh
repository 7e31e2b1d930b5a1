R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r04tr; mkdir -p $O
for h in 64 128; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/ktc_$h -- python $R/tools/bench_coma.py --steps 6 --warmup 2 --no-cpu-baseline --critic-hidden $h > $O/coma_$h.json 2>/dev/null
  python $R/tools/trace_timeline.py $(find /tmp/ktc_$h -name "*kernel_trace.csv" | head -1) k_ro 3 > $O/timeline_coma_$h.txt 2>&1
  cat $O/timeline_coma_$h.txt
done
