R=$GRAFT_REPO_ROOT
cd /tmp
run() { CM_CRITIC_OVERLAP=$1 python $R/bench.py --workload $2 --envs $3 --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; b=json.loads(sys.stdin.read()); print('$2 envs $3 overlap=$1', round(b['ms_per_step'],4), {k:round(v,3) for k,v in b['phase_ms'].items()})"; }
for rep in 1 2; do
for ov in 0 2; do run $ov cfg3 1024; done
for ov in 0 2; do run $ov cfg3 768; done
for ov in 0 1 2; do run $ov cfg4 256; done
for ov in 0 2; do run $ov cfg4 512; done
done
