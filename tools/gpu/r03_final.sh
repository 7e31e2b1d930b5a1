R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r03f; mkdir -p $O
(cd $R && timeout 900 python -m pytest tests/test_cli_gpu.py tests/test_dist_gpu.py tests/test_coma_gpu.py -m gpu -x -q 2>&1 | tail -4) > $O/tests.txt
cat $O/tests.txt
for e in 512 1024 2048 4096; do python $R/bench.py --workload cfg3 --envs $e --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; b=json.loads(sys.stdin.read()); print('cfg3 envs', $e, round(b['ms_per_step'],4), {k:round(v,3) for k,v in b['phase_ms'].items()}, b['kernel_ms'])"; done | tee $O/bench_shares.txt
python $R/tools/cli_steady_state.py 2>/dev/null | tee $O/cli_steady_state.txt
bash $R/tools/gpu/r03_cli_trace.sh | tail -30
