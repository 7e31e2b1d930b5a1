R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r04cfg5; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt5 -- python $R/bench.py --workload cfg5 --steps 6 --warmup 3 --no-cpu-baseline --no-extras > $O/bench.json 2>/dev/null
python $R/tools/trace_timeline.py $(find /tmp/kt5 -name "*kernel_trace.csv" | head -1) k_gru32_ro 3 > $O/timeline_cfg5.txt 2>&1
head -70 $O/timeline_cfg5.txt
