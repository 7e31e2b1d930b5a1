R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03o
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -q -x -k "act or shape or wide_actor" 2>&1 | tail -4
CM_PROF_WARMUP=20 python tools/phase_prof.py act 2>&1 | grep -v amdgpu | tee $O/phase_act.txt
python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg4', 'ms_per_step %.4f' % d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items()}, d['roofline']['frac'])" | tee $O/bench_cfg4.txt
