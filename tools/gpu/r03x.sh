R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03x
mkdir -p $O
cd $R
python -m pytest tests/test_fused_step.py tests/test_hip_parity.py tests/test_dist_gpu.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
for w in "cfg3 --envs 512" "cfg3 --envs 256" "cfg3 --envs 1024" "cfg3" "cfg2" "cfg4 --envs 256" "cfg5"; do
  n=$(echo $w | tr -d ' -')
  python $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_$n.json 2>/dev/null
  python - <<P
import json
b=json.load(open("$O/bench_$n.json")); print("$n", round(b["ms_per_step"],4), {k:round(v,4) for k,v in b["phase_ms"].items()})
P
done
for w in "cfg3" "cfg5" "cfg2"; do
  n=$(echo $w | tr -d ' -')
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k_$n -- python $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>/dev/null
  cp $(find /tmp/k_$n -name "*kernel_stats.csv" | head -1) $O/${n}_kernel_stats.csv
done
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --workload cfg3 --envs 512 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
python $R/tools/trace_timeline.py $(find /tmp/kt -name "*kernel_trace.csv" | head -1) k_ro 3 > $O/timeline_cfg3envs512.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -- python $R/bench.py --workload cfg2 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
python $R/tools/trace_timeline.py $(find /tmp/kt2 -name "*kernel_trace.csv" | head -1) k_ro 3 > $O/timeline_cfg2.txt 2>&1
