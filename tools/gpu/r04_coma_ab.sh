# COMA: parity tests + bench at both critic widths (64 = the MAPPO-sized critic, 128 = the reference's default)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04coma2
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q -k "coma or wide or 128 or fused" 2>&1 | tail -5 | tee $O/tests.txt
for w in 64 128; do
  python tools/bench_coma.py --critic-hidden $w --no-cpu-baseline > $O/coma$w.json 2> $O/coma$w.err || tail -5 $O/coma$w.err
  python - <<PY
import json
d=json.loads(open("$O/coma$w.json").read().strip().splitlines()[-1])
print($w, d["ms_per_step"], d["phase_ms"], d["kernel_ms"])
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc -- python $R/tools/bench_coma.py --critic-hidden 128 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/kc -name "*kernel_stats.csv" | head -1) $O/coma128_kernel_stats.csv
python - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04coma2/coma128_kernel_stats.csv')))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]: print("%-90s calls %5s avg %9.1f us %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
