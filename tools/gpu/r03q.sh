R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03q
mkdir -p $O
cd $R
bash tools/gpu/r03o.sh
timeout 900 python -m pytest tests/test_hip_parity.py -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k4 -- python $R/bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
cp $(find /tmp/k4 -name "*kernel_stats.csv" | head -1) $O/cfg4_kernel_stats.csv
head -8 $O/cfg4_kernel_stats.csv | cut -c1-150
