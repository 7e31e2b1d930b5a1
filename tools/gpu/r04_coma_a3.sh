# COMA at the reference's default 3-agent simple_spread shapes (rows of 21 / 54 floats): parity on padded buffers + bench at both critic widths
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_coma_gpu.py tests/test_eval.py -q -m gpu 2>&1 | tail -3
python -m pytest tests -q -m gpu -k "wide or 128 or layered or forward" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for h in 64 128; do
rm -rf /tmp/kc
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc -- python $R/tools/bench_coma.py --agents 3 --envs 4096 --critic-hidden $h --no-cpu-baseline > /tmp/o.json 2>/dev/null
python - <<PY
import json,csv,glob
d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); print('COMA 3 agents critic $h', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms'].items()})
f=glob.glob('/tmp/kc/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:7]: print("   %-90s calls %4s avg %8.1f us %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
done
python $R/tools/bench_coma.py --critic-hidden 64 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('COMA 8 agents critic 64', round(d['ms_per_step'],3))"
