R=$GRAFT_REPO_ROOT
cd /tmp
for rep in 1 2 3; do for ov in auto 1 2; do
  if [ $ov = auto ]; then unset CM_CRITIC_OVERLAP; else export CM_CRITIC_OVERLAP=$ov; fi
  python $R/bench.py --workload cfg2 --steps 100 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; b=json.loads(sys.stdin.read()); print('cfg2 overlap=$ov', round(b['ms_per_step'],4), {k:round(v,3) for k,v in b['phase_ms'].items()})"
done; done
