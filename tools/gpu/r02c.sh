set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
for e in 512 2048; do CM_PROF_WARMUP=100 python tools/phase_prof.py rollout $e 8 >> $O/phase_rollout16.txt 2>&1; done
CM_ROLLOUT_TILE=64 CM_PROF_WARMUP=100 python tools/phase_prof.py rollout 512 8 >> $O/phase_rollout16.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
