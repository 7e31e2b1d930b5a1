# COMA with 10 agents (the 16-agent form of the GEMM epilogue) against round 3's separate launches (wide_schedule=fused_r3), both critic widths
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for h in 64 128; do for sched in auto fused_r3; do
  CM_WIDE_SCHEDULE=$sched python $R/tools/bench_coma.py --agents 10 --envs 2048 --critic-hidden $h --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('agents 10, envs 2048, critic $h, $sched:', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms'].items()})"
done; done
