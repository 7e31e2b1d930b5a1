# round 2, step x: GRU step products with hand-ordered LDS reads: parity, config 5, phase profile
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02x
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "gru" 2>&1 | tail -3
python bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5', round(d['ms_per_step'],3), {k: round(v['ms'],3) for k,v in d['phase_roofline'].items()})" | tee $O/cfg5.txt
CM_PROF_WARMUP=50 python tools/phase_prof.py gru 2>&1 | grep -v amdgpu | tee $O/phase_gru.txt
