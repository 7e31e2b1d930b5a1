# round 5, first call: the dist / eval tests that changed (world 2, 4, 8 on one GPU), then the whole GPU suite, then the bench line
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r05a
mkdir -p $O
cd $GRAFT_REPO_ROOT
(time timeout 1500 python -m pytest tests/test_dist_gpu.py tests/test_eval.py -x -q -m gpu) > $O/dist_tests.txt 2>&1; tail -25 $O/dist_tests.txt
(time timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_dist_gpu.py --deselect tests/test_eval.py) > $O/gpu_tests.txt 2>&1; tail -8 $O/gpu_tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
