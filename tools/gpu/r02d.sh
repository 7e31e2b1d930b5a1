set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02d
mkdir -p $O
cd $R
python tools/probes/rollout_tile_ab.py > $O/rollout_tile_ab.txt 2>&1
cat $O/rollout_tile_ab.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
CM_CRITIC_OVERLAP=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --envs 512 > $O/bench_512_nooverlap.json 2>> $O/bench.err
CM_CRITIC_OVERLAP=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --envs 512 > $O/bench_512_overlap.json 2>> $O/bench.err
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_dist_gpu.py -x -q -m gpu -k "rollout or dist or bench or golden" 2>&1 | tail -8 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
