# round 4, first GPU pass: new tests first, then the whole GPU suite, bench, CLI steady state with the reference's eval cadence
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r04a
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_eval.py tests/test_pins.py -x -q -m gpu > $O/new_tests.txt 2>&1; tail -15 $O/new_tests.txt
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "full_size or rollout_tilings or fused_rollout" > $O/fullsize_tests.txt 2>&1; tail -5 $O/fullsize_tests.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_noextras.json 2> $O/bench.err; cat $O/bench_noextras.json
timeout 600 python bench.py --steps 20 --warmup 5 --envs 512 --no-cpu-baseline --no-extras > $O/bench_512.json 2>> $O/bench.err; cat $O/bench_512.json
timeout 1500 python tools/cli_steady_state.py > $O/cli_steady_state.txt 2>&1; cat $O/cli_steady_state.txt
(time timeout 2400 python -m pytest tests -x -q -m gpu) > $O/gpu_tests.txt 2>&1; tail -8 $O/gpu_tests.txt
