R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03i
mkdir -p $O
cd $R
for w in "cfg3 --envs 256" "cfg3 --envs 512" "cfg3 --envs 1024" "cfg2" "cfg4 --envs 256" "cfg4 --envs 512"; do
 for ov in 1 2; do
  for cs in "" "fused"; do
    n=$(echo $w | tr -d ' -')
    CM_CRITIC_OVERLAP=$ov CM_CRITIC_SCHEDULE=$cs python bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$n overlap=$ov critic=$cs', 'ms_per_step %.4f' % d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items()})" | tee -a $O/ab.txt
  done
 done
done
