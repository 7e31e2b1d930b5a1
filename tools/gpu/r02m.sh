R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02m
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -12 > $O/pytest_parity.txt
cat $O/pytest_parity.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2>> $O/err.txt
tail -c 1500 $O/bench.json
