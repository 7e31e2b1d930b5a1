# round 2, step s: fused vs split critic per share size (threshold check) + parity of both schedules
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02s
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "critic or golden or update or additiv or padded" 2>&1 | tail -5
for e in 4096 2048 1024 512; do for s in fused split; do
  CM_CRITIC_SCHEDULE=$s python bench.py --envs $e --steps 30 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e $s', round(d['ms_per_step'],3), round(d['phase_roofline']['critic_fwd_bwd']['ms'],3))" | tee -a $O/ab.txt
done; done
