# fused optimiser step: bit-identity tests + A/B against the stand-alone launches
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03b
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_fused_step.py -q -x 2>&1 | tail -15 > $O/pytest_fused.txt
cat $O/pytest_fused.txt
for f in 1 0; do
  for w in "cfg3 --envs 512" "cfg2" "cfg5" "cfg3"; do
    n=$(echo $w | tr -d ' -')
    CM_FUSED_STEP=$f python bench.py --workload $w --steps 30 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fused=$f', '$n', 'ms_per_step %.4f' % d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items()})" | tee -a $O/ab_fused_step.txt
  done
done
