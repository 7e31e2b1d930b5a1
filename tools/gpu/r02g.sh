set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "gru" 2>&1 | tail -15 > $O/pytest_gru.txt
cat $O/pytest_gru.txt
python bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/cfg5_v2.json 2>> $O/err.txt
CM_GRU_TILE=32 python bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/cfg5_v1.json 2>> $O/err.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k5 -- python $R/bench.py --workload cfg5 --no-cpu-baseline --no-extras > /dev/null 2>&1
cp $(find /tmp/k5 -name "*kernel_stats.csv" | head -1) $O/cfg5_kernel_stats.csv
head -8 $O/cfg5_kernel_stats.csv | cut -c1-150
