R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "gru_rollout" 2>&1 | tail -5
cd /tmp
for t in 16 auto; do
  CM_ROLLOUT_TILE=$t timeout 300 python $R/bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; b=json.loads(sys.stdin.read()); print('cfg5 rollout_tile=$t', round(b['ms_per_step'],4), {k:round(v,3) for k,v in b['phase_ms'].items()})"
done
timeout 300 python $R/tools/phase_prof.py grurollout 2>&1 | grep -v amdgpu.ids
