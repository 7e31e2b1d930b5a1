# round 4: full GPU suite + bench (new issued_flop fields) + rollout A/B of the compile-time K
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_noextras.json 2> $O/bench.err; cat $O/bench_noextras.json
timeout 600 python bench.py --steps 20 --warmup 5 --envs 512 --no-cpu-baseline --no-extras > $O/bench_512.json 2>> $O/bench.err; cat $O/bench_512.json
(time timeout 3000 python -m pytest tests -q -m gpu) > $O/gpu_tests.txt 2>&1; tail -12 $O/gpu_tests.txt
