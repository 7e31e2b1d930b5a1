R=$GRAFT_REPO_ROOT
cd /tmp
run() { CM_CRITIC_OVERLAP=$1 CM_CRITIC_SCHEDULE=$4 CM_DW0_BATCH=$5 python $R/bench.py --workload $2 --envs $3 --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; b=json.loads(sys.stdin.read()); print('$2 envs $3 overlap=$1 critic=$4 dw0=$5', round(b['ms_per_step'],4), {k:round(v,3) for k,v in b['phase_ms'].items()})"; }
for rep in 1 2; do
for e in 1024 768 1536; do
run 2 cfg3 $e auto auto; run 2 cfg3 $e split auto; run 2 cfg3 $e split 4
done
done
