# the driver's command once more at the final commit: profiles/r04_bench.json with the stamped PMC record accepted
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r04final
mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json,os
o=json.load(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04final/bench.json'))
print(o['ms_per_step'], o['value'], o['roofline'])
print({k:round(v['ms_per_step'],3) for k,v in o['other_workloads'].items()}, {k:v['shares'] and {kk:round(vv['ms_per_step'],3) for kk,vv in v['shares'].items()} for k,v in o['strong_scaling_shares'].items()})
print(o.get('cpu_baseline',{}).get('value'), o.get('extras_error'), o.get('cpu_baseline_error'))
PY
