R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02p
mkdir -p $O
cd $R
for a in "--envs 512" "--envs 1024" "--envs 2048" "--workload cfg2" "--workload cfg4 --envs 256" "--workload cfg4 --envs 512"; do python bench.py $a --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:30], d['config']['envs_per_gpu'], round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['phase_ms'].items()})"; done | tee $O/bench.txt
timeout 1500 python -m pytest tests/test_dist_gpu.py tests/test_hip_parity.py -x -q -m gpu -k "dist or two_rank or golden" 2>&1 | tail -4
