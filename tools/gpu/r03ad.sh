R=$GRAFT_REPO_ROOT
cd /tmp
for w in "cfg3 --envs 256" "cfg3 --envs 128" "cfg3 --envs 384" "cfg4 --envs 128" "cfg2 --envs 512"; do
for ov in 1 2; do
  CM_CRITIC_OVERLAP=$ov python $R/bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; b=json.loads(sys.stdin.read()); print('$w overlap=$ov', round(b['ms_per_step'],4), {k:round(v,3) for k,v in b['phase_ms'].items()})"
done; done
