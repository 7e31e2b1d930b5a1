# round 2, step v: 16-row rollout with the policy in registers (LDS 49 -> 9.5 KB): parity, rollout time, share times
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02v
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "rollout" 2>&1 | tail -4
python tools/probes/rollout_tile_ab.py 2>&1 | grep -v amdgpu | tee $O/rollout_tile_ab.txt
for e in 1024 512 256; do
  python bench.py --envs $e --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', round(d['ms_per_step'],3), {k: round(v['ms'],3) for k,v in d['phase_roofline'].items()})" | tee -a $O/shares.txt
done
for o in 1 2; do CM_CRITIC_OVERLAP=$o python bench.py --envs 512 --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('512 overlap=$o', round(d['ms_per_step'],3))" | tee -a $O/shares.txt; done
python bench.py --workload cfg2 --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', round(d['ms_per_step'],3))" | tee -a $O/shares.txt
