R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tests
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
