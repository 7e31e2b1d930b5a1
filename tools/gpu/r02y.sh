# round 2, step y: per-kernel times after the hand-ordered product forms (512-env share of config 3, config 4)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02y
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in "cfg3 --envs 512" "cfg4"; do
  n=$(echo $w | tr -d ' -')
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k_$n -- python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$n.json 2>/dev/null
  cp $(find /tmp/k_$n -name "*kernel_stats.csv" | head -1) $O/${n}_kernel_stats.csv
done
for i in 1 2 3; do python $R/bench.py --envs 512 --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('512', round(d['ms_per_step'],3))"; done
