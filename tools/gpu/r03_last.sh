# the round's last commit: full GPU suite, the driver's bench command, the same command under rocprofv3 --kernel-trace --stats
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03last; mkdir -p $O
(cd $R && time timeout 1200 python -m pytest tests -m gpu -x -q) > $O/gputests.txt 2>&1; tail -3 $O/gputests.txt
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k5 -- python $R/bench.py --workload cfg3 --envs 512 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_cfg3envs512.json 2>/dev/null
cp $(find /tmp/k5 -name "*kernel_stats.csv" | head -1) $O/cfg3envs512_kernel_stats.csv
python $R/tools/trace_timeline.py $(find /tmp/k5 -name "*kernel_trace.csv" | head -1) k_ro 3 > $O/timeline_cfg3envs512.txt 2>&1
head -c 400 $O/bench.json
