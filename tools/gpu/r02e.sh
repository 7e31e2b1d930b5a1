set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
for m in 0 1 2; do
  for e in 512 1024 2048 4096; do CM_CRITIC_OVERLAP=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --envs $e >> $O/cfg3_m$m.txt 2>> $O/err.txt; done
  for e in 256 512 2048; do CM_CRITIC_OVERLAP=$m python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --envs $e >> $O/cfg4_m$m.txt 2>> $O/err.txt; done
  CM_CRITIC_OVERLAP=$m python bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline --no-extras >> $O/cfg2_m$m.txt 2>> $O/err.txt
done
tail -5 $O/err.txt
