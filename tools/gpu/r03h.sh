R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03h
mkdir -p $O
cd $R
for lib in "" "cleanmarl_amd/libcleanmarl_hip_prio0.so"; do
 for ov in 1 2; do
  for cs in "" "fused"; do
    CM_LIB_PATH=$lib CM_CRITIC_OVERLAP=$ov CM_CRITIC_SCHEDULE=$cs python bench.py --workload cfg3 --envs 512 --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg3envs512 lib=$lib overlap=$ov critic=$cs', 'ms_per_step %.4f' % d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items()})" | tee -a $O/ab.txt
  done
 done
done
