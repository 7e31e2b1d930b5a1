R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_hip_parity.py -q -x -k "forward_sweeps" 2>&1 | tail -2
for t in auto; do
CM_GRU_TILE=$t timeout 300 python bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg5 $t', 'ms_per_step %.4f' % d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items()})"
done
timeout 300 python tools/phase_prof.py grurollout 2>&1 | grep -v amdgpu.ids | tail -12
