R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_fused_step.py tests/test_hip_parity.py -q -x -k "fused or golden or overlapped or deterministic" 2>&1 | tail -3
for w in "cfg3 --envs 512" "cfg2" "cfg3 --envs 1024" "cfg3"; do
python bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w', 'ms_per_step %.4f' % d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items()})"
done
