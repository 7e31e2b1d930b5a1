# rollout wave priority x critic-overlap schedule x fused step at the 512-env share (and config 2)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03d
mkdir -p $O
cd $R
for w in "cfg3 --envs 512" "cfg2" "cfg3 --envs 1024"; do
 for f in 1 0; do
  for ov in 1 2; do
    n=$(echo $w | tr -d ' -')
    CM_CRITIC_OVERLAP=$ov CM_FUSED_STEP=$f python bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$n', 'fused=$f overlap=$ov', 'ms_per_step %.4f' % d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items()})" | tee -a $O/ab.txt
  done
 done
done
