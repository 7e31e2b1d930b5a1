# round 2, step w: fused critic at 2 input chunks (config 4, per-agent critic on 125-wide observations)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02w
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -k "update or padded or critic" 2>&1 | tail -3
for s in fused split; do
  CM_CRITIC_SCHEDULE=$s python bench.py --workload cfg4 --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 $s', round(d['ms_per_step'],3), {k: round(v['ms'],3) for k,v in d['phase_roofline'].items()})" | tee -a $O/ab.txt
done
