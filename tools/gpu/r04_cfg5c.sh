# config 5: where the critic's epochs run (CM_GRU_CRITIC), after the state rows were padded (one-pass critic, 111 us alone)
R=$GRAFT_REPO_ROOT
cd $R
for k in 1 2 3; do for sched in defer1 defer2; do CM_GRU_CRITIC=$sched python bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 gru_critic=$sched', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d.get('phase_ms').items()})"; done; done
