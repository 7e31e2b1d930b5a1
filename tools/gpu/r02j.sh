R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02j
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_cli_gpu.py tests/test_dist_gpu.py -x -q -m gpu -k "gru or lstm" 2>&1 | tail -8 > $O/pytest_gru.txt
cat $O/pytest_gru.txt
python bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/cfg5.json 2>> $O/err.txt
CM_PROF_WARMUP=50 python tools/phase_prof.py gru > $O/phase_gru.txt 2>&1; cat $O/phase_gru.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k5 -- python $R/bench.py --workload cfg5 --no-cpu-baseline --no-extras > /dev/null 2>&1
cp $(find /tmp/k5 -name "*kernel_stats.csv" | head -1) $O/cfg5_kernel_stats.csv
head -6 $O/cfg5_kernel_stats.csv | cut -c1-130
