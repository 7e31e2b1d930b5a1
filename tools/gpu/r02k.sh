R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02k
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_cli_gpu.py -x -q -m gpu -k "pinned or config1 or smaclite or script_runs" 2>&1 | tail -12 > $O/pytest.txt
cat $O/pytest.txt
python tools/bench_host_env.py 256 8 128 > $O/host_env.txt 2>&1; cat $O/host_env.txt
