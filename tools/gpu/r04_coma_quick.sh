# COMA 128: bench line + kernel stats only (no tests)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04coma3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/tools/bench_coma.py --critic-hidden 128 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(128, d['ms_per_step'], d['kernel_ms'])"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc -- python $R/tools/bench_coma.py --critic-hidden 128 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/kc -name "*kernel_stats.csv" | head -1) $O/coma128_kernel_stats.csv
python - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04coma3/coma128_kernel_stats.csv')))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:9]: print("%-90s calls %5s avg %9.1f us %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
