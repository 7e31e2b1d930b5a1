# round 2, step a: per-GPU shares of the strong-scaling split at HEAD (before any change) + kernel breakdown of the 512-env share
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for e in 4096 2048 1024 512; do python $R/bench.py --workload cfg3 --envs $e --steps 20 --warmup 5 --no-cpu-baseline >> $O/cfg3_shares.txt 2>/dev/null; done
for e in 2048 1024 512 256; do python $R/bench.py --workload cfg4 --envs $e --steps 10 --warmup 3 --no-cpu-baseline >> $O/cfg4_shares.txt 2>/dev/null; done
for e in 1024 512 256 128; do python $R/bench.py --workload cfg5 --envs $e --steps 10 --warmup 3 --no-cpu-baseline >> $O/cfg5_shares.txt 2>/dev/null; done
for e in 1024 128; do python $R/bench.py --workload cfg2 --envs $e --steps 20 --warmup 5 --no-cpu-baseline >> $O/cfg2_shares.txt 2>/dev/null; done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k512 -- python $R/bench.py --envs 512 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/k512 -name "*kernel_stats.csv" | head -1) $O/cfg3_512_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k256 -- python $R/bench.py --workload cfg4 --envs 256 --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/k256 -name "*kernel_stats.csv" | head -1) $O/cfg4_256_kernel_stats.csv
ls -la $O
