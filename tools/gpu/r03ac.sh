R=$GRAFT_REPO_ROOT
cd /tmp
for w in "cfg3 --envs 1024" "cfg3 --envs 2048" "cfg3 --envs 768" "cfg4 --envs 512" "cfg4 --envs 256" "cfg2"; do
for ov in 0 1 2; do
  CM_CRITIC_OVERLAP=$ov python $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; b=json.loads(sys.stdin.read()); print('$w overlap=$ov', round(b['ms_per_step'],4), {k:round(v,3) for k,v in b['phase_ms'].items()})"
done; done
