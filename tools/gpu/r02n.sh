R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "gru or rollout" 2>&1 | tail -4 > $O/pytest.txt
cat $O/pytest.txt
python tools/probes/rollout_tile_ab.py 2>&1 | grep -v amdgpu | tee $O/rollout_tile_ab.txt
for a in "--workload cfg5" "--workload cfg2" "--envs 512"; do python bench.py $a --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['phase_ms'].items()})"; done | tee $O/bench.txt
