# round 2, step z: hand-ordered forms in the 64-row rollout too
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02z
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "rollout" 2>&1 | tail -3
python tools/probes/rollout_tile_ab.py 2>&1 | grep -v amdgpu | tail -8
run() { python bench.py "$@" --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['roofline']['frac'],3) if d.get('roofline') else '')"; }
  echo "cfg3 full        $(run --steps 20 --warmup 5)" | tee -a $O/ab3.txt
  echo "cfg3 2048 envs   $(run --envs 2048 --steps 20 --warmup 5)" | tee -a $O/ab3.txt
