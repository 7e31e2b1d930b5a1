# COMA with the reference's default 128-wide critic: kernel stats of one bench run
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r04coma
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/bench_coma.py --critic-hidden 128 --no-cpu-baseline > $O/coma128.json 2> $O/coma128.err || python $R/tools/bench_coma.py --critic-hidden 128 > $O/coma128.json 2>> $O/coma128.err
cat $O/coma128.json | head -c 3000
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc -- python $R/tools/bench_coma.py --critic-hidden 128 > /dev/null 2>&1
cp $(find /tmp/kc -name "*kernel_stats.csv" | head -1) $O/coma128_kernel_stats.csv
python - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04coma/coma128_kernel_stats.csv')))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]: print("%-90s calls %5s avg %9.1f us %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
