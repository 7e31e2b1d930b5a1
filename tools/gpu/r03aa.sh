R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_fused_step.py tests/test_dist_gpu.py -m gpu -x -q 2>&1 | tail -2
cd /tmp
for rep in 1 2; do
for w in "cfg3 --envs 512" "cfg3" "cfg2" "cfg5"; do
  python $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; b=json.loads(sys.stdin.read()); print('$w', round(b['ms_per_step'],4), {k:round(v,3) for k,v in b['phase_ms'].items()})"
done; done
