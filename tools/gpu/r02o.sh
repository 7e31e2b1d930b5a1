R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02o
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --envs 512 --steps 6 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# keep the last ~2 iterations
t0=int(rows[0]["Start_Timestamp"])
out=open("$O/trace_512.txt","w")
last=rows[-140:]
base=int(last[0]["Start_Timestamp"])
for r in last:
    s=(int(r["Start_Timestamp"])-base)/1e3; e=(int(r["End_Timestamp"])-base)/1e3
    out.write("%9.1f %9.1f %7.1f  q%s  %s\n"%(s,e,e-s,r.get("Queue_Id","?"),r["Kernel_Name"][:70]))
PY
