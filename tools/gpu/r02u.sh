# round 2, step u: timeline of the 512-env share (kernel trace) with the row thresholds in place + share times
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02u
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for e in 4096 2048 1024 512; do python $R/bench.py --workload cfg3 --envs $e --steps 30 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', round(d['ms_per_step'],3), {k: round(v['ms'],3) for k,v in d['phase_roofline'].items()})" | tee -a $O/shares.txt; done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k512 -- python $R/bench.py --envs 512 --steps 12 --warmup 4 --no-cpu-baseline --no-extras > /dev/null 2>&1
cp $(find /tmp/k512 -name "*kernel_stats.csv" | head -1) $O/cfg3_512_kernel_stats.csv
cp $(find /tmp/k512 -name "*kernel_trace.csv" | head -1) $O/cfg3_512_kernel_trace.csv
